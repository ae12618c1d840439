"""Build libbpp_amd.so in-tree with hipcc for gfx950 (MI355X).

    python -m bpp_amd.build

Cross-compiles without a GPU.  -ffp-contract=off is part of the numerical
contract (CLVs bit-identical to the reference's AVX2 back-end): the only fused
multiply-adds are explicit __builtin_fma calls.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libbpp_amd.so")
HOST_OUT = os.path.join(HERE, "libbpp_amd_host.so")      # host-side MCMC control in C (gcc), links libbpp_amd.so
HOST_SRC = os.path.join(CSRC, "host", "a00_driver.c")
RCCL_OUT = os.path.join(HERE, "libbpp_amd_rccl.so")      # the several-GPU exchange as native code (links librccl), a library of its own
RCCL_SRC = os.path.join(CSRC, "rccl_exchange.c")
SOURCES = ["engine.hip", "host_math.cpp", "host_input.cpp"]
# every header the library's sources include: all of csrc/*.hpp and include/*.h (a header missing from a hand-kept list is a
# stale library that looks built)
import glob as _glob
DEPS = sorted(_glob.glob(os.path.join(CSRC, "*.hpp"))) + sorted(_glob.glob(os.path.join(CSRC, "experimental", "*.hpp"))) + sorted(_glob.glob(os.path.join(ROOT, "include", "*.h")))


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    files = [os.path.join(CSRC, s) for s in SOURCES] + \
            [d if os.path.isabs(d) else os.path.join(CSRC, d) for d in DEPS] + [__file__]
    return any(os.path.getmtime(f) > t for f in files)


def build_host(force=False, verbose=False):
    deps = [HOST_SRC, os.path.join(ROOT, "include", "bpp_amd_host.h"), os.path.join(ROOT, "include", "bpp_amd.h"), OUT]
    if not force and os.path.exists(HOST_OUT) and all(os.path.getmtime(f) <= os.path.getmtime(HOST_OUT) for f in deps):
        return HOST_OUT
    cmd = ["gcc", "-O2", "-std=c99", "-fopenmp", "-fPIC", "-shared", "-Wall", "-I", os.path.join(ROOT, "include")] + \
          (["-DBPA_EXPERIMENTAL"] if experimental() else []) + [
           HOST_SRC, "-o", HOST_OUT, "-L", HERE, "-lbpp_amd", "-Wl,-rpath,$ORIGIN", "-lm"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return HOST_OUT


def build_rccl(force=False, verbose=False):
    deps = [RCCL_SRC, os.path.join(ROOT, "include", "bpp_amd_rccl.h")]
    if not force and os.path.exists(RCCL_OUT) and all(os.path.getmtime(f) <= os.path.getmtime(RCCL_OUT) for f in deps):
        return RCCL_OUT
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-O2", "-std=gnu99", "-fPIC", "-shared", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
           RCCL_SRC, "-o", RCCL_OUT, "-L", os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return RCCL_OUT


def build_rccl_optional(force=False, verbose=False):
    """libbpp_amd_rccl.so is a library of its own so that libbpp_amd.so does not depend on RCCL: a box without
    rccl.h / librccl gets the core libraries and a warning (RcclExchange.lib() reports the missing .so)."""
    try:
        return build_rccl(force, verbose)
    except (subprocess.CalledProcessError, OSError) as ex:
        print(f"[bpp_amd.build] libbpp_amd_rccl.so not built ({ex}); the RCCL exchange is unavailable", file=sys.stderr)
        return None


def experimental():
    """BPA_EXPERIMENTAL=1 (or -DBPA_EXPERIMENTAL among BPA_HIPCC_FLAGS): the build that also compiles csrc/experimental/ and
    reads the A/B switches of superseded variants (csrc/device_types.hpp: BPA_EXP_SWITCH)"""
    return bool(os.environ.get("BPA_EXPERIMENTAL")) or "-DBPA_EXPERIMENTAL" in os.environ.get("BPA_HIPCC_FLAGS", "").split()


def build(force=False, verbose=False):
    if not force and not stale():
        build_host(False, verbose)
        build_rccl_optional(False, verbose)
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
           "-fPIC", "-shared", "-Wall", "-Wno-unused-result", "-Wno-pass-failed",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "-o", OUT] + os.environ.get("BPA_HIPCC_FLAGS", "").split() + (["-DBPA_EXPERIMENTAL"] if os.environ.get("BPA_EXPERIMENTAL") else []) + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    build_host(True, verbose)
    build_rccl_optional(force, verbose)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
