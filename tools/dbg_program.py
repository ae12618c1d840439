"""decision logs of the host driver and the persistent kernel side by side on the data of
tests/test_gpu_sampler.py::test_several_sequences_per_species[True]:  BPA_SMP_DBG=256 A00_DECLOG=1 python tools/dbg_program.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bpp_amd, tape, hostdrv
eng = bpp_amd.Engine(0)
rng = np.random.default_rng(8)
nloci = 120
species = [0, 0, 0, 1, 1, 1]
parent, tau0, thetas = [2, 2, -1], [0.0, 0.0, 0.003], [0.002, 0.003, 0.004]
data = []
for _ in range(nloci):
    t = sorted(rng.uniform(0.0002, 0.0028, 4))
    left = [-1] * 6 + [0, 6, 3, 8, 7]
    right = [-1] * 6 + [1, 2, 4, 5, 9]
    times = [0.0] * 6 + [t[0], t[2], t[1], t[3], 0.003 + rng.uniform(0.0005, 0.004)]
    seqs = ["".join(rng.choice(list("ACGT"), 60)) for _ in range(2)]
    seqs = [seqs[0]] * 3 + [seqs[1]] * 3
    seqs = ["".join(c if rng.random() > 0.05 else rng.choice(list("ACGT")) for c in s) for s in seqs]
    pats, w = bpp_amd.compress_site_patterns(seqs, True, True)
    data.append(dict(seqs=pats, weights=w, left=left, right=right, times=times, root=10, states=4, rate_cats=1, model="jc69", rates=np.ones(1)))
host = hostdrv.hip_driver(eng, tape.make_engine_loci(eng, data), data, seed=4)
dev = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=4)
for drv in (host, dev):
    drv.set_proposal_kernel(1)
    drv.set_program_moves(True, 0.5)
    drv.set_species_tree(parent, tau0, thetas)
    for i in range(nloci):
        drv.set_tip_species(i, species)
    drv.set_tau_prior(3.0, 1000.0)
    drv.set_theta_prior(2.0, 700.0, 0.002)
    drv.set_finetune(0.003, 0.004, 0.0008, 0.2)
host.initialize(); dev.initialize()
for it in range(3):
    print(f"--- iteration {it}", file=sys.stderr, flush=True)
    host.iterate(); dev.iterate(1)
    s = dev.summary()
    print(it, (s["proposals"], s["accepted"]), host.counters()[:2], list(dev.thetas()), list(host.thetas()), list(dev.taus()), list(host.taus()), file=sys.stderr, flush=True)
