"""registers / scratch / LDS of the kernels in libbpp_amd.so (the gfx950 code object inside its fat binary)
usage: python tools/kres.py [substring of the kernel name ...]"""
import os, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = open(os.path.join(ROOT, "bpp_amd", "libbpp_amd.so"), "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
pos = so.find(MAGIC)
assert pos >= 0, "no uncompressed offload bundle (CCOB-compressed?)"
n, = struct.unpack_from("<Q", so, pos + 24)
p = pos + 32
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", so, p)
    triple = so[p + 24:p + 24 + tl].decode()
    p += 24 + tl
    if "gfx950" in triple:
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(so[pos + off:pos + off + size])
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        cur = {}
        rows = []
        for line in out.splitlines():
            line = line.strip()
            for key in (".name:", ".vgpr_count:", ".sgpr_count:", ".private_segment_fixed_size:", ".group_segment_fixed_size:", ".vgpr_spill_count:", ".agpr_count:"):
                if line.startswith(key) or line.startswith("- " + key):
                    cur[key] = line.split(":", 1)[1].strip()
            if line.startswith(".wavefront_size:") or line.startswith("- .wavefront_size:"):
                rows.append(cur); cur = {}
        for r in rows:
            name = r.get(".name:", "?")
            if sys.argv[1:] and not any(a in name for a in sys.argv[1:]):
                continue
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print(f"{dem[:90]:90s} vgpr {r.get('.vgpr_count:')} agpr {r.get('.agpr_count:')} sgpr {r.get('.sgpr_count:')} scratch {r.get('.private_segment_fixed_size:')} spill {r.get('.vgpr_spill_count:')} lds {r.get('.group_segment_fixed_size:')}")
