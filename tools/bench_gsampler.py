"""per-iteration time of the generic device-resident sampler (gsampler.hpp) on a BASELINE config's shape
usage: bench_gsampler.py c2|c3|c5like [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import bpp_amd
from bpp_amd import synth
import tape
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
shape = {"c2": (10000, 1000, 4, "jc69", 1), "c3": (10000, 1000, 8, "gtr", 4), "c3s": (2000, 1000, 8, "gtr", 4)}[cfg]
t0 = time.time()
data = synth.make_dataset(*shape, seed=12345)
eng = bpp_amd.Engine(0)
loci = tape.make_engine_loci(eng, data)
if cfg == "c2":
    os.environ["BPA_SMP_GENERIC"] = "1"
smp = bpp_amd.Sampler(eng, loci, data, seed=1)
par, tau, theta = synth.species_tree_arrays(shape[2])
smp.set_species_tree(par, tau, theta)
smp.set_tau_prior(3.0, 3.0 / tau[-1])
smp.set_theta_prior(2.0, 2.0 / theta[0], 0.5 * theta[0])
smp.initialize(); smp.iterate(2); eng.synchronize()
print("setup %.1fs" % (time.time() - t0), file=sys.stderr)
l0 = smp.summary()["launches"]
smp.enable_timing(1)
t0 = time.perf_counter(); smp.iterate(iters); eng.synchronize(); dt = time.perf_counter() - t0
tm = smp.timing(); sm = smp.summary()
print(cfg, "ms/iter %.3f  it/s %.1f  launches/iter %.1f  acceptance %.3f" % (1e3*dt/iters, iters/dt, (sm["launches"] - l0 - 1)/iters, sm["accepted"]/max(sm["proposals"], 1)),
      "step-kernel us: per-locus %.1f all-loci %.1f" % (1e3*tm["sweep_ms"]/max(tm["sweep_launches"], 1), 1e3*tm["allloci_ms"]/max(tm["allloci_launches"], 1)))
