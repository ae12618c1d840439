"""registers / LDS / scratch of the kernels of libbpp_amd.so's gfx950 code object whose mangled name holds every argument:
python tools/kernel_regs.py gstep2"""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = open(os.path.join(ROOT, "bpp_amd", "libbpp_amd.so"), "rb").read()
pos = so.find(b"__CLANG_OFFLOAD_BUNDLE__")
n, = struct.unpack_from("<Q", so, pos + 24)
p = pos + 32
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", so, p)
    triple = so[p + 24:p + 24 + tl].decode()
    p += 24 + tl
    if "gfx950" not in triple:
        continue
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(so[pos + off:pos + off + size])
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    os.unlink(f.name)
    for blk in out.split("- .agpr_count")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        if all(a in name for a in sys.argv[1:]):
            print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
