"""the device-resident sampler on BASELINE config 2 (10 000 loci x 1 000 sites, 4 species, JC69) with the priors of
the reference run recorded in tests/golden/a00_posterior_10k.json; prints the posterior summary"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bpp_amd
from bpp_amd import synth
import tape
t0 = time.time()
data = synth.make_dataset(10000, 1000, 4, "jc69", 1, seed=12345)
eng = bpp_amd.Engine(0)
loci = tape.make_engine_loci(eng, data)
smp = bpp_amd.Sampler(eng, loci, data, seed=3)
parent, tau, theta = synth.species_tree_arrays(4)
smp.set_species_tree(parent, tau, theta)
smp.set_theta_prior(2.0, 1000.0, 8e-5)
smp.set_tau_prior(2.0, 666.0)
smp.set_finetune(0.004, 0.004, 4e-5, 0.006)
smp.initialize()
print("setup", time.time() - t0, file=sys.stderr)
t0 = time.time()
smp.iterate(1000)
S = []
for _ in range(6000):
    smp.iterate(1)
    S.append(smp.thetas()[4:] + smp.taus()[4:])
eng.synchronize()
print("sampling", time.time() - t0, file=sys.stderr)
S = np.array(S)
names = ["theta_AB", "theta_ABC", "theta_root", "tau_AB", "tau_ABC", "tau_root"]
out = {}
for k, nm in enumerate(names):
    x = S[:, k]
    r1 = float(np.corrcoef(x[:-1], x[1:])[0, 1])
    out[nm] = dict(mean=float(x.mean()), sd=float(x.std()), rho1=r1)
out["lnL"] = smp.summary()["total_lnl"]
out["acceptance"] = smp.summary()["accepted"] / smp.summary()["proposals"]
print(json.dumps(out, indent=1))
