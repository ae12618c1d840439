"""edge sizes of the device samplers against the host driver: 1, 2, 13, 64, 65 loci (one workgroup, a partial one, the
boundary), 4 / 6 / 8 taxa (BPA_SMP_GENERIC=1: the generic sampler) — same decisions, trees, taus and thetas (run by hand through gpurun)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bpp_amd
from bpp_amd import synth
import hostdrv, tape
bad_total = 0
for taxa in (4, 6, 8):
    for nloci in (1, 2, 13, 64, 65):
        eng = bpp_amd.Engine(0)
        data = synth.make_dataset(nloci, 300, taxa, "jc69", 1, seed=100 + nloci)
        la, lb = tape.make_engine_loci(eng, data), tape.make_engine_loci(eng, data)
        host = hostdrv.hip_driver(eng, la, data, seed=5)
        dev = bpp_amd.Sampler(eng, lb, data, seed=5)
        parent, tau0, thetas = synth.species_tree_arrays(taxa)
        for d in (host, dev):
            d.set_species_tree(parent, tau0, thetas)
            d.set_tau_prior(3.0, 3.0 / tau0[-1])
            d.set_theta_prior(2.0, 1000.0, 0.0004)
        host.initialize(); dev.initialize()
        for it in range(25):
            host.iterate(); dev.iterate(1)
        s = dev.summary(); hp, ha, _ = host.counters()
        ok = (s["proposals"], s["accepted"]) == (hp, ha)
        ok &= bool(np.allclose(dev.taus(), host.taus(), rtol=1e-10, atol=0)) and bool(np.allclose(dev.thetas(), host.thetas(), rtol=1e-10, atol=0))
        bad = 0
        for i in range(nloci):
            a, b = dev.tree(i), host.tree(i)
            if [int(x) for x in a["parent"]] != [int(x) for x in b["parent"]] or not np.allclose(a["time"], b["time"], rtol=1e-10, atol=0):
                bad += 1
        print(f"taxa {taxa:2d} loci {nloci:3d}: counters/taus/thetas equal {ok}, trees differing {bad}, lnL {s['total_lnl']:.6f} vs {host.total_lnl():.6f}")
        bad_total += (0 if ok else 1) + bad
        host.close(); dev.close(); eng.close()
print("EDGE", "OK" if bad_total == 0 else f"FAILED ({bad_total})")
