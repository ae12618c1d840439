"""all-loci decisions of the persistent kernel (BPA_SMP_DBG=256) next to the host driver's (A00_DECLOG=1), config 2"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BPA_SMP_DBG"] = "256"; os.environ["A00_DECLOG"] = "1"
import numpy as np
import bpp_amd
from bpp_amd import synth
import tape, hostdrv
N = int(os.environ.get("DBG_LOCI", "10000"))
eng = bpp_amd.Engine(0)
data = synth.make_dataset(N, 1000, 4, "jc69", 1, seed=12345)
la, lb = tape.make_engine_loci(eng, data), tape.make_engine_loci(eng, data)
host = hostdrv.hip_driver(eng, la, data, seed=3)
dev = bpp_amd.Sampler(eng, lb, data, seed=3)
par, tau, theta = synth.species_tree_arrays(4)
for s in (host, dev):
    s.set_species_tree(par, tau, theta)
    s.set_theta_prior(3.0, 1500.0, 8e-5)
    s.set_tau_prior(3.0, 1000.0)
    s.set_finetune(0.004, 0.004, 4e-5, 0.006)
host.initialize(); dev.initialize()
for it in range(4):
    print("=== iteration", it, flush=True); sys.stderr.flush()
    host.iterate(); dev.iterate(1)
    s = dev.summary()
    print("dev", s["proposals"], s["accepted"], "host", host.counters()[:2], flush=True)
