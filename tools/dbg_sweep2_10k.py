"""persistent kernel vs one-launch-per-step path on config 2's 10 000 loci (157 workgroups)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bpp_amd
from bpp_amd import synth
import tape
eng = bpp_amd.Engine(0)
data = synth.make_dataset(10000, 1000, 4, "jc69", 1, seed=12345)
NOMIX = os.environ.get("DBG_NOMIX") is not None
def make(v1):
    os.environ.pop("BPA_SMP_V1", None); os.environ.pop("BPA_SMP_NOMIX", None)
    if v1: os.environ["BPA_SMP_V1"] = "1"
    if NOMIX: os.environ["BPA_SMP_NOMIX"] = "1"
    loci = tape.make_engine_loci(eng, data)
    s = bpp_amd.Sampler(eng, loci, data, seed=3)
    par, tau, theta = synth.species_tree_arrays(4)
    s.set_species_tree(par, tau, theta)
    s.set_theta_prior(3.0, 1500.0, 8e-5)
    s.set_tau_prior(3.0, 1000.0)
    s.set_finetune(0.004, 0.004, 4e-5, 0.006)
    s.initialize()
    return s
new, old = make(False), make(True)
prev = None
for it in range(40):
    new.iterate(1); old.iterate(1)
    tn, to, hn, ho = new.taus(), old.taus(), new.thetas(), old.thetas()
    sn, so = new.summary(), old.summary()
    same = tn == to and hn == ho and (sn["proposals"], sn["accepted"]) == (so["proposals"], so["accepted"])
    print(it, "same" if same else "DIFF", sn["proposals"], sn["accepted"], so["accepted"], sn["total_lnl"] - so["total_lnl"], [a - b for a, b in zip(tn[4:], to[4:])], [a - b for a, b in zip(hn[4:], ho[4:])])
    if not same:
        nbad = 0
        for i in range(10000):
            x, y = new.tree(i), old.tree(i)
            keys = [k for k in ("left", "right", "parent", "clv", "pmat", "pop", "time") if list(x[k]) != list(y[k])]
            if keys or x["lnl"] != y["lnl"] or x["logpr"] != y["logpr"]:
                nbad += 1
                if nbad <= 3:
                    print("locus", i, keys, "np", len(data[i]["weights"]))
                    for k in keys + ["lnl", "logpr"]:
                        print("   new", k, x[k]); print("   old", k, y[k])
        print("loci differing:", nbad)
        break
