"""config 2's 10 000 loci: iterations/s of the persistent kernel with (a) our uniform kernel, (b) BPP's kernel, (c) BPP's kernel +
the program's moves; BPA_SMP_DBG=16 prints the phase cycles of workgroup 0"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bpp_amd
from bpp_amd import synth
import tape
eng = bpp_amd.Engine(0)
data = synth.make_dataset(10000, 1000, 4, "jc69", 1, seed=12345)
which = sys.argv[1:] or ["uniform", "bpp", "program"]
for mode in which:
    loci = tape.make_engine_loci(eng, data)
    s = bpp_amd.Sampler(eng, loci, data, seed=3)
    if mode != "uniform":
        s.set_proposal_kernel(1)
    if mode == "program":
        s.set_program_moves(True, 0.1)
    par, tau, theta = synth.species_tree_arrays(4)
    s.set_species_tree(par, tau, theta)
    s.set_theta_prior(2.0, 1000.0, 3e-5 if mode != "uniform" else 8e-5)
    s.set_tau_prior(2.0, 500.0)
    if mode == "uniform":
        s.set_finetune(0.004, 0.004, 4e-5, 0.006)
    else:
        s.set_finetune(18.5, 0.0019, 1.9e-5, 0.0059)
    s.initialize()
    s.iterate(300); eng.synchronize()
    t0 = time.perf_counter(); s.iterate(3000); eng.synchronize(); dt = time.perf_counter() - t0
    sm = s.summary()
    print(f"{mode:8s} {3000/dt:9.1f} it/s  {dt/3000*1e3:.4f} ms/iteration  acceptance {sm['accepted']/sm['proposals']:.3f} gibbs {s.gibbs_counters()}", flush=True)
    s.close()
eng.close()
