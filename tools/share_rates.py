"""iterations/s of the persistent kernel (the program's moves) on 10 000 / 5 000 / 2 500 / 1 250 loci of config 2 — what one rank
of an N-GPU strong-scaling run works on"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bpp_amd
from bpp_amd import synth
import tape
eng = bpp_amd.Engine(0)
full = synth.make_dataset(10000, 1000, 4, "jc69", 1, seed=12345)
for n in (10000, 8192, 5000, 2500, 1250):
    data = full[:n]
    s = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=3)
    s.set_proposal_kernel(1); s.set_program_moves(True, 0.1)
    par, tau, theta = synth.species_tree_arrays(4)
    s.set_species_tree(par, tau, theta)
    sc = (10000.0/n)**0.5
    s.set_theta_prior(2.0, 1000.0, 3e-5*sc); s.set_tau_prior(2.0, 500.0)
    s.set_finetune(18.5, 0.0019, 1.9e-5*sc, 0.0059*sc)
    s.initialize(); s.iterate(300); eng.synchronize()
    t0 = time.perf_counter(); s.iterate(3000); eng.synchronize(); dt = time.perf_counter() - t0
    print(f"{n:6d} loci  {3000/dt:9.1f} it/s  {dt/3000*1e3:.4f} ms/iteration", flush=True)
    s.close()
eng.close()
