"""Posterior of the UNMODIFIED reference program (oracle/_ref/bpp, A00, JC69) on a synthetic 30-locus
4-species data set — the end-to-end known answer for the sampler (tests/test_a00_posterior.py).

    python tests/golden/make_golden_a00.py         ->  tests/golden/a00_posterior.json
    python tests/golden/make_golden_a00.py --10k   ->  tests/golden/a00_posterior_10k.json   (BASELINE config 2:
        10 000 loci x 1 000 sites, seed 12345; thetaprior gamma 2 1000, tauprior gamma 2 666, 1 000 + 3 000
        iterations, threads = 8 1 1; about three minutes)

The data come from bpp_amd.synth (seed 77; the tests regenerate them), are written as a sequential
PHYLIP file + Imap + control file, and bpp runs 4 000 burn-in + 20 000 x 2 iterations with
thetaprior = gamma 2 500, tauprior = gamma 2 400, seed 1.  Only the summary goes into the fixture.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from bpp_amd import synth          # noqa: E402

CFG = dict(nloci=30, sites=500, taxa=4, seed=77, theta=0.004, theta_prior=(2.0, 500.0), tau_prior=(2.0, 400.0),
           burnin=4000, sampfreq=2, nsample=20000)


BIG = dict(nloci=10000, sites=1000, taxa=4, seed=12345, theta=None, theta_prior=(2.0, 1000.0), tau_prior=(2.0, 666.0),
           burnin=1000, sampfreq=1, nsample=3000, threads="8 1 1")


def main():
    global CFG
    big = "--10k" in sys.argv
    if big:
        CFG = BIG
    data = synth.make_dataset(CFG["nloci"], CFG["sites"], CFG["taxa"], "jc69", 1, seed=CFG["seed"], theta=CFG["theta"])
    names = "ABCD"
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "seqs.txt"), "w") as f:
            for d in data:
                seqs = ["".join(ch * int(w) for ch, w in zip(s, d["weights"])) for s in d["seqs"]]
                f.write(f"4 {len(seqs[0])}\n")
                for nm, s in zip(names, seqs):
                    f.write(f"s^{nm.lower()}  {s}\n")
                f.write("\n")
        with open(os.path.join(td, "imap.txt"), "w") as f:
            f.write("a A\nb B\nc C\nd D\n")
        with open(os.path.join(td, "a00.ctl"), "w") as f:
            f.write("seed = 1\nseqfile = seqs.txt\nImapfile = imap.txt\njobname = out\nspeciesdelimitation = 0\n"
                    "speciestree = 0\nspecies&tree = 4  A B C D\n                  1 1 1 1\n"
                    "                 (((A, B), C), D);\nusedata = 1\n"
                    f"nloci = {CFG['nloci']}\ncleandata = 0\n"
                    f"thetaprior = gamma {CFG['theta_prior'][0]:g} {CFG['theta_prior'][1]:g}\n"
                    f"tauprior = gamma {CFG['tau_prior'][0]:g} {CFG['tau_prior'][1]:g}\nfinetune = 1\nprint = 1 0 0 0\n"
                    f"burnin = {CFG['burnin']}\nsampfreq = {CFG['sampfreq']}\nnsample = {CFG['nsample']}\n"
                    + (f"threads = {CFG['threads']}\n" if "threads" in CFG else ""))
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "bpp"), "--cfile", "a00.ctl"], cwd=td, check=True,
                       stdout=subprocess.DEVNULL)
        rows = [ln.split("\t") for ln in open(os.path.join(td, "out.mcmc.txt"))]
    head, body = rows[0], np.array([[float(x) for x in r] for r in rows[1:]])
    # bpp's columns: theta:5:A,B,C,D theta:6:A,B,C theta:7:A,B tau:5.. tau:6.. tau:7.. lnL -> our population order
    want = {"theta_root": 1, "theta_ABC": 2, "theta_AB": 3, "tau_root": 4, "tau_ABC": 5, "tau_AB": 6, "lnL": 7}
    out = dict(config=CFG, columns=[h.strip() for h in head], samples=len(body),
               posterior={k: dict(mean=float(body[:, c].mean()), sd=float(body[:, c].std())) for k, c in want.items()})
    with open(os.path.join(HERE, "a00_posterior_10k.json" if big else "a00_posterior.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["posterior"], indent=1))


if __name__ == "__main__":
    main()
