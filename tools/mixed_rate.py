"""config 2's 10 000 JC69 loci with a few loci of another kind among them (GTR+Gamma4): iterations/s of the composite sampler
next to the pure set on the persistent kernel (library's own moves both)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bpp_amd
from bpp_amd import synth
import tape
eng = bpp_amd.Engine(0)
full = synth.make_dataset(10000, 1000, 4, "jc69", 1, seed=12345)
odd = synth.make_dataset(64, 1000, 4, "gtr", 4, seed=99)
for nodd in (0, 10, 64):
    data = full[:10000 - nodd] + odd[:nodd]
    s = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=3)
    par, tau, theta = synth.species_tree_arrays(4)
    s.set_species_tree(par, tau, theta)
    s.set_theta_prior(2.0, 1000.0, 8e-5); s.set_tau_prior(2.0, 500.0)
    s.set_finetune(0.004, 0.004, 4e-5, 0.006)
    s.initialize(); s.iterate(20); eng.synchronize()
    n = 2000 if nodd == 0 else 200
    t0 = time.perf_counter(); s.iterate(n); eng.synchronize(); dt = time.perf_counter() - t0
    sm = s.summary()
    print(f"{10000 - nodd} JC69 + {nodd} GTR+G4 loci: {s.kind():10s} {n/dt:9.1f} it/s  {dt/n*1e3:.4f} ms/iteration  acceptance {sm['accepted']/sm['proposals']:.3f}", flush=True)
    s.close()
eng.close()
