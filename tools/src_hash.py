"""sha256 over the library's sources (bpp_amd/csrc/**, include/*.h, bpp_amd/build.py): what a committed profile must have been
taken on.  tools/profile_cfg.sh records it as `kernels_sha` (the GPU box has no .git); tests/test_profiles_current.py and
bench.py compare it with the tree's."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src_hash(root=ROOT):
    files = []
    for base, _, names in os.walk(os.path.join(root, "bpp_amd", "csrc")):
        files += [os.path.join(base, n) for n in names if n.endswith((".hip", ".hpp", ".cpp", ".c", ".h"))]
    files += [os.path.join(root, "include", n) for n in os.listdir(os.path.join(root, "include")) if n.endswith(".h")]
    files.append(os.path.join(root, "bpp_amd", "build.py"))
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    sys.stdout.write(src_hash() + "\n")
