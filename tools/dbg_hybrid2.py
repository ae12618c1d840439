import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bpp_amd
from bpp_amd import synth
import tape
eng = bpp_amd.Engine(0)
data = synth.make_dataset(160, 300, 4, "jc69", 1, seed=3)
t = torch.zeros(16, dtype=torch.float64, device="cuda")
def make(settle):
    os.environ.pop("BPA_SMP_SETTLE", None)
    if settle: os.environ["BPA_SMP_SETTLE"] = "1"
    s = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=7)
    s.set_allreduce(lambda p, n, st: True, t.data_ptr(), 0)
    par, tau, theta = synth.species_tree_arrays(4)
    s.set_species_tree(par, tau, theta)
    s.set_tau_prior(3.0, 1000.0); s.set_theta_prior(2.0, 1000.0, 0.001); s.set_finetune(0.003, 0.005, 0.0008, 0.2)
    s.initialize()
    return s
a, b = make(False), make(True)
for call in range(8):
    a.iterate(2); b.iterate(2)
    sa, sb = a.summary(), b.summary()
    print(call, sa["accepted"], sb["accepted"], a.taus() == b.taus())
    if sa["accepted"] != sb["accepted"]:
        nd = 0
        for i in range(160):
            x, y = a.tree(i), b.tree(i)
            d = [k for k in ("left", "right", "parent", "clv", "pmat", "pop", "time", "lnl", "logpr") if (list(x[k]) if hasattr(x[k], "__len__") else x[k]) != (list(y[k]) if hasattr(y[k], "__len__") else y[k])]
            if d:
                nd += 1
                if nd <= 2:
                    print(" locus", i, d)
                    for k in d: print("   v2 ", k, [float(v) for v in x[k]] if hasattr(x[k], "__len__") else x[k]); print("   old", k, [float(v) for v in y[k]] if hasattr(y[k], "__len__") else y[k])
        print("loci differing", nd)
        break
