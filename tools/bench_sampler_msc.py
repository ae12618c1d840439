"""per-launch times of the device-resident MSC sampler on config 2 (BPA_SMP_TRACE=1 prints them)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bpp_amd
from bpp_amd import synth
import tape
eng = bpp_amd.Engine(0)
data = synth.make_dataset(10000, 1000, 4, "jc69", 1, seed=12345)
loci = tape.make_engine_loci(eng, data)
smp = bpp_amd.Sampler(eng, loci, data, seed=1)
par, tau, theta = synth.species_tree_arrays(4)
smp.set_species_tree(par, tau, theta)
smp.set_tau_prior(3.0, 3.0 / tau[-1])
smp.initialize(); smp.iterate(3); eng.synchronize()
if not os.environ.get("BPA_SMP_TRACE"):
    t0 = time.perf_counter(); smp.iterate(50); eng.synchronize(); dt = time.perf_counter() - t0
    print("ms/iter", 1e3 * dt / 50, smp.summary())
else:
    smp.iterate(2)
