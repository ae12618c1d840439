"""disassemble one kernel of libbpp_amd.so's gfx950 code object: python tools/disasm.py <substring of the mangled/demangled name>"""
import os, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = open(os.path.join(ROOT, "bpp_amd", "libbpp_amd.so"), "rb").read()
pos = so.find(b"__CLANG_OFFLOAD_BUNDLE__")
n, = struct.unpack_from("<Q", so, pos + 24)
p = pos + 32
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", so, p)
    triple = so[p + 24:p + 24 + tl].decode()
    p += 24 + tl
    if "gfx950" in triple:
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(so[pos + off:pos + off + size])
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--demangle", f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        on = False
        for line in out.splitlines():
            if line.endswith(">:"):
                on = all(a in line for a in sys.argv[1:])
            if on:
                print(line)
