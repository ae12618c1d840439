import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import bpp_amd
from bpp_amd import synth
import hostdrv, tape
eng = bpp_amd.Engine(0)
taxa, model, R, nloci = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
data = synth.make_dataset(nloci, 300, taxa, model, R, seed=19)
la = tape.make_engine_loci(eng, data); lb = tape.make_engine_loci(eng, data)
host = hostdrv.hip_driver(eng, la, data, seed=29)
os.environ["BPA_SMP_GENERIC"] = "1"
dev = bpp_amd.Sampler(eng, lb, data, seed=29)
parent, tau0, thetas = synth.species_tree_arrays(taxa)
for drv in (host, dev):
    drv.set_species_tree(parent, tau0, thetas); drv.set_tau_prior(3.0, 3.0/tau0[-1]); drv.set_theta_prior(2.0, 1000.0, 0.001); drv.set_finetune(0.003, 0.005, 0.0008, 0.2)
host.initialize(); dev.initialize()
for it in range(int(sys.argv[5]) if len(sys.argv) <= 6 else 0):
    bad = 0
    for i in range(nloci):
        a, b = dev.tree(i), host.tree(i)
        same = all([int(x) for x in a[k]] == [int(x) for x in b[k]] for k in ("left","right","parent","clv","pmat","pop")) and a["root"] == b["root"]
        ok = same and abs(a["lnl"]-b["lnl"]) <= 1e-10*abs(b["lnl"]) and abs(a["logpr"]-b["logpr"]) <= 1e-10*abs(b["logpr"]) and np.allclose(a["time"], b["time"], rtol=1e-12, atol=0)
        if not ok:
            bad += 1
            if bad <= 3:
                print("iter", it, "locus", i, "same-ints", same, "lnl", a["lnl"], b["lnl"], "logpr", a["logpr"], b["logpr"])
                if not same:
                    for k in ("left","right","parent","clv","pmat","pop"): print("   ", k, [int(x) for x in a[k]], [int(x) for x in b[k]])
                print("    time", [float(x) for x in a["time"]], [float(x) for x in b["time"]])
    print("iter", it, "bad loci", bad, "taus", dev.taus()[taxa:], host.taus()[taxa:], "summary", dev.summary(), host.counters())
    host.iterate(); dev.iterate(1)
if len(sys.argv) > 6:
    for li in (0, 1, 2):
        print("locus", li)
        for k in range(2*(2*taxa-2)):
            pa, pb = lb[li].get_pmatrix(k), la[li].get_pmatrix(k)
            if not np.array_equal(pa, pb): print("  pmat", k, pa[0,0,:2], pb[0,0,:2])
        for c in range(taxa, taxa + 2*(taxa-1)):
            ca, cb = lb[li].get_clv(c), la[li].get_clv(c)
            if not np.array_equal(ca, cb): print("  clv", c, ca.ravel()[:4], cb.ravel()[:4])
