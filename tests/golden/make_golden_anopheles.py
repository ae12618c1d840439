"""BASELINE config 5 fixture (run in the build container, where /root/reference and oracle/_ref/libbppref.so exist):

    python tests/golden/make_golden_anopheles.py

  tests/golden/anopheles/loci_realign.txt, Imap.txt   the data files of examples/anopheles, copied verbatim
  tests/golden/anopheles_pipeline.json                what the REAL reference makes of them with the settings of
      anopheles-bpp-msci.ctl (JC69, cleandata = 1, 100 loci): per locus the kept sequences, the compressed patterns
      with their weights (method.c:3299-3459 through oracle/ref_shim_input.c) and the reference's own
      locus_root_loglikelihood (AVX2 back-end) on a seeded random gene tree (the tree is stored too).
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
SRC = "/root/reference/examples/anopheles"


def main():
    import make_golden_input as G
    import oraclelib as O
    from common import rand_tree
    L = G.shim()
    adir = os.path.join(HERE, "anopheles")
    os.makedirs(adir, exist_ok=True)
    for f in ("loci_realign.txt", "Imap.txt"):
        shutil.copyfile(os.path.join(SRC, f), os.path.join(adir, f))
        os.chmod(os.path.join(adir, f), 0o644)
    pipe = G.pipeline(L, os.path.join(adir, "loci_realign.txt"), 0, True, True, None, None)["loci"]
    rng = np.random.default_rng(55)
    loci = []
    for w in pipe:
        a1 = w["a1"]
        tips = len(a1["seqs"])
        left, right, times, root = rand_tree(tips, rng, 0.02)
        rl = O.RefLocus(4, 1, a1["seqs"], a1["weights"])
        rl.set_tree(left, right, times, root)
        lnl = rl.full_lnl()
        loci.append(dict(labels=a1["labels"], removed=w["removed"], clean_length=len(w["clean"][0]), seqs=a1["seqs"],
                         weights=a1["weights"], tree=dict(left=left, right=right, times=[float(t).hex() for t in times], root=root),
                         lnl=float(lnl).hex()))
    out = dict(species=["G", "C", "R", "L", "A", "Q"], loci=loci,
               total_lnl=float(sum(float.fromhex(l["lnl"]) for l in loci)).hex())
    with open(os.path.join(HERE, "anopheles_pipeline.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(len(loci), "loci; patterns", min(len(l["weights"]) for l in loci), "-", max(len(l["weights"]) for l in loci),
          "total lnL", float.fromhex(out["total_lnl"]))


if __name__ == "__main__":
    main()
