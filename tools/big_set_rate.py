"""more config-2 loci than the persistent kernel holds (16 384 four-taxon loci) fall to the one-launch-per-step path: its rate
next to the resident 10 000 loci (python tools/big_set_rate.py 10000 | 40000); the library's moves"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bpp_amd
from bpp_amd import synth
import tape
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
eng = bpp_amd.Engine(0)
data = synth.make_dataset(N, 1000, 4, "jc69", 1, seed=12345)
s = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=3)
par, tau, theta = synth.species_tree_arrays(4)
s.set_species_tree(par, tau, theta)
s.set_theta_prior(2.0, 1000.0, 8e-5*(10000.0/N)**0.5); s.set_tau_prior(2.0, 500.0)
s.set_finetune(0.004, 0.004, 4e-5*(10000.0/N)**0.5, 0.006*(10000.0/N)**0.5)
s.initialize(); s.iterate(20); eng.synchronize()
n = 300
t0 = time.perf_counter(); s.iterate(n); eng.synchronize(); dt = time.perf_counter() - t0
sm = s.summary()
print(f"{N} loci: {s.kind():10s} {n/dt:9.1f} it/s  {dt/n*1e3:.4f} ms/iteration  acceptance {sm['accepted']/sm['proposals']:.3f}  lnL {sm['total_lnl']:.6f}", flush=True)
s.close(); eng.close()
