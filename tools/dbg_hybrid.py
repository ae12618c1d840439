"""hybrid path (persistent kernel's sweep + one launch per all-loci step, an all-reduce callback installed) against the
one-launch-per-step path with the same (identity) callback"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bpp_amd
from bpp_amd import synth
import tape
eng = bpp_amd.Engine(0)
data = synth.make_dataset(160, 300, 4, "jc69", 1, seed=3)
t = torch.zeros(16, dtype=torch.float64, device="cuda")
def make(v1, ar):
    os.environ.pop("BPA_SMP_V1", None)
    if v1: os.environ["BPA_SMP_V1"] = "1"
    s = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=7)
    if ar: s.set_allreduce((lambda p, n, st: (eng.synchronize(), torch.cuda.synchronize(), True)[2]) if os.environ.get("DBG_SYNC") else (lambda p, n, st: True), t.data_ptr(), 0)
    par, tau, theta = synth.species_tree_arrays(4)
    s.set_species_tree(par, tau, theta)
    s.set_tau_prior(3.0, 1000.0); s.set_theta_prior(2.0, 1000.0, 0.001); s.set_finetune(0.003, 0.005, 0.0008, 0.2)
    s.initialize()
    return s
hyb, old, per = make(False, True), make(True, True), make(False, False)
print(hyb.kind(), old.kind(), per.kind())
for it in range(12):
    for s in (hyb, old, per): s.iterate(int(os.environ.get("DBG_N", "1")))
    a, b, c = hyb.summary(), old.summary(), per.summary()
    print(it, "hyb/old/per accepted", a["accepted"], b["accepted"], c["accepted"], "taus equal:", hyb.taus() == old.taus(), per.taus() == old.taus(),
          "thetas equal:", hyb.thetas() == old.thetas(), per.thetas() == old.thetas())
    if a["accepted"] != b["accepted"]:
        for i in range(160):
            x, y = hyb.tree(i), old.tree(i)
            d = [k for k in ("left", "right", "parent", "clv", "pmat", "pop", "time", "lnl", "logpr") if (list(x[k]) if hasattr(x[k], "__len__") else x[k]) != (list(y[k]) if hasattr(y[k], "__len__") else y[k])]
            if d:
                print(" locus", i, d)
                for k in d: print("   hyb", k, x[k]); print("   old", k, y[k])
                break
        break
