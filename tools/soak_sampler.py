"""one-off soak: host driver (libbpp_amd back-end) vs device-resident sampler, 2 000 loci x 150 full A00 iterations,
4 and 8 taxa — the trajectories must still coincide (same accept/reject history, taus, thetas, trees)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bpp_amd
from bpp_amd import synth
import hostdrv, tape
SCALE = int(os.environ.get("SOAK_SCALE", "1"))
for taxa, nloci, iters in ((4, 2000, 150*SCALE), (8, 600, 60*SCALE)):
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(nloci, 500, taxa, "jc69", 1, seed=41)
    la, lb = tape.make_engine_loci(eng, data), tape.make_engine_loci(eng, data)
    host = hostdrv.hip_driver(eng, la, data, seed=77)
    dev = bpp_amd.Sampler(eng, lb, data, seed=77)
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    for d in (host, dev):
        d.set_species_tree(parent, tau0, thetas)
        d.set_tau_prior(3.0, 3.0 / tau0[-1])
        d.set_theta_prior(2.0, 1000.0, 0.0004)
        d.set_finetune(0.003, 0.004, 0.0002, 0.02)
    host.initialize(); dev.initialize()
    for it in range(iters):
        host.iterate(); dev.iterate(1)
    s = dev.summary(); hp, ha, _ = host.counters()
    ok = (s["proposals"], s["accepted"]) == (hp, ha)
    ok &= bool(np.allclose(dev.taus(), host.taus(), rtol=1e-10, atol=0)) and bool(np.allclose(dev.thetas(), host.thetas(), rtol=1e-10, atol=0))
    bad = 0
    for i in range(nloci):
        a, b = dev.tree(i), host.tree(i)
        if [int(x) for x in a["parent"]] != [int(x) for x in b["parent"]] or not np.allclose(a["time"], b["time"], rtol=1e-10, atol=0):
            bad += 1
    print(f"taxa {taxa}: {iters} iterations, proposals {s['proposals']} accepted {s['accepted']} (host {hp} {ha}), "
          f"taus/thetas equal: {ok}, loci with different trees: {bad}, lnL {s['total_lnl']:.4f} vs {host.total_lnl():.4f}")
    host.close(); dev.close(); eng.close()
