"""the C host driver on GTR+G4 loci of 8 taxa (bpa_batch_evaluate's general path): ms per A00 iteration"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bpp_amd
from bpp_amd import synth
import hostdrv, tape
n = int(os.environ.get("NLOCI", "2000"))
eng = bpp_amd.Engine(0)
data = synth.make_dataset(n, 1000, 8, "gtr", 4, seed=12345)
loci = tape.make_engine_loci(eng, data)
g = hostdrv.hip_driver(eng, loci, data, seed=1)
parent, tau, theta = synth.species_tree_arrays(8)
g.set_species_tree(parent, tau, theta)
g.set_tau_prior(3.0, 3.0 / tau[-1])
g.initialize()
for _ in range(2): g.iterate()
t0 = time.perf_counter(); k = 8
for _ in range(k): g.iterate()
dt = time.perf_counter() - t0
p, a, s = g.counters()
print(f"host driver, {n} loci x 8 taxa GTR+G4: {1e3*dt/k:.2f} ms/iteration, {s//(k+2)} steps/iteration, acceptance {a/p:.3f}")
