"""CPU side of the drop-in check (tests/test_gpu_bpp_hip.py runs it on the GPU): oracle/_ref/bpp_hip — the reference's
unmodified objects + integration/locus_hip.c + libbpp_amd.so — links, resolves BPP's locus API to the shim (also for
the callers inside locus.c), and fails loudly without a GPU instead of falling back to the CPU kernels it still
contains."""
import os
import subprocess
import pytest
import bpphip as B

pytestmark = pytest.mark.skipif(not B.have_binaries(), reason="oracle/_ref/bpp{,_hip} not built (needs /root/reference)")

API = ["locus_update_matrices", "locus_update_all_matrices", "locus_update_partials", "locus_update_all_partials",
       "locus_root_loglikelihood", "locus_create", "locus_destroy", "pll_set_tip_states", "pll_set_pattern_weights",
       "pll_core_update_pmatrix"]


def test_symbols_resolve_to_the_shim():
    nm = subprocess.run(["nm", B.HIP_BIN], check=True, stdout=subprocess.PIPE, text=True).stdout
    sym = {}
    for ln in nm.splitlines():
        p = ln.split()
        if len(p) == 3:
            sym[p[2]] = (int(p[0], 16), p[1])
    for s in API:
        assert s in sym and sym[s][1] == "T", s            # one strong definition: the shim's
    for s in ("locus_create", "locus_destroy", "pll_set_tip_states", "pll_set_pattern_weights"):
        assert "bppref_" + s in sym                           # the reference's host-side originals
    for s in ("bpa_locus_create", "bpa_locus_update_matrices", "bpa_locus_update_partials", "bpa_locus_root_loglikelihood",
              "bpa_core_update_pmatrix"):
        assert f" U {s}" in nm                                # bound to libbpp_amd.so
    # the reference's own locus.o code lies between its renamed functions; the shim's definitions come after it
    lo = min(sym["bppref_locus_create"][0], sym["bppref_pll_set_pattern_weights"][0])
    assert all(sym[s][0] > lo for s in API)
    # the simulator's call of the library form (simulate.c:694) goes to the shim as well
    assert any("call" in ln and "<pll_core_update_pmatrix>" in ln for ln in subprocess.run(
        ["objdump", "-d", "--no-show-raw-insn", B.HIP_BIN], check=True, stdout=subprocess.PIPE, text=True).stdout.splitlines())
    # calls inside locus.c (its substitution-parameter proposals) go to the shim too: the weakened originals are gone
    dis = subprocess.run(["objdump", "-d", "--no-show-raw-insn", B.HIP_BIN], check=True, stdout=subprocess.PIPE, text=True).stdout
    callers = [ln for ln in dis.splitlines() if "call" in ln and "<locus_update_partials>" in ln]
    assert len(callers) >= 10                                  # gtree.c, stree.c, method.c ... and locus.c's four


def test_fails_loudly_without_a_gpu():
    import bpp_amd
    if bpp_amd.lib().bpa_device_count() > 0:
        pytest.skip("a GPU is visible")
    G = B.GOLDEN
    files = {"frogs.txt": os.path.join(G, "frogs", "frogs.txt"), "frogs.Imap.txt": os.path.join(G, "frogs", "frogs.Imap.txt")}
    rc, out, _ = B.run_program(B.HIP_BIN, B.FROGS_CTL.format(burnin=0, sampfreq=1, nsample=2, extra=""), files)
    assert rc != 0
    assert "[bpp_hip]" in out and "no HIP device" in out
