"""the C host driver (a00_driver.c: host MCMC control, one batched bpa_batch_evaluate per proposal step) on config 2:
whole A00 iterations per second of the host-driven drop-in path"""
import os, sys, time
os.environ.setdefault("OMP_PROC_BIND", "close")     # worker threads of the driver stay put (A00_THREADS=n sets their number)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bpp_amd
from bpp_amd import synth
import hostdrv, tape
n = int(os.environ.get("NLOCI", "10000"))
eng = bpp_amd.Engine(0)
data = synth.make_dataset(n, 1000, 4, "jc69", 1, seed=12345)
loci = tape.make_engine_loci(eng, data)
g = hostdrv.hip_driver(eng, loci, data, seed=1)
parent, tau, theta = synth.species_tree_arrays(4)
g.set_species_tree(parent, tau, theta)
g.set_tau_prior(3.0, 3.0 / tau[-1])
g.set_theta_prior(2.0, 2.0 / theta[0], 0.5 * theta[0])
g.initialize()
for _ in range(3): g.iterate()
t0 = time.perf_counter()
k = 20
for _ in range(k): g.iterate()
dt = time.perf_counter() - t0
p, a, s = g.counters()
print(f"host driver on libbpp_amd.so: {1e3*dt/k:.2f} ms/iteration = {k/dt*n/10000:.1f} iterations/s (10k-locus), {s//(k+3)} steps/iteration, acceptance {a/p:.3f}")
