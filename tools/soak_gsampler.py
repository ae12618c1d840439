"""one-off soak of the generic device sampler's default path (per-locus steps as two half-batches on two streams) and of the
host driver's two-cohort pipeline: 3 000 GTR+G4 loci of 8 taxa with all parameter moves —
 (1) device sampler vs the plain host driver: same accept/reject history, taus, thetas, trees after every block;
 (2) host driver with two cohorts on two engines vs the plain one: the same to the bit;
 (3) every locus's lnL in the device state against a root evaluation of the buffers the state names."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bpp_amd
from bpp_amd import synth
import hostdrv, tape
nloci, taxa, R = int(os.environ.get("SOAK_LOCI", "3000")), 8, 4
blocks, per = int(os.environ.get("SOAK_BLOCKS", "6")), int(os.environ.get("SOAK_PER", "5"))
data = synth.make_dataset(nloci, 400, taxa, "gtr", R, seed=61)
e0, e1, e2, e3 = (bpp_amd.Engine(0) for _ in range(4))
plain = hostdrv.hip_driver(e0, tape.make_engine_loci(e0, data), data, seed=5)
split = nloci // 2
co = hostdrv.hip_driver_cohorts([e1, e2], tape.make_engine_loci(e1, data[:split]) + tape.make_engine_loci(e2, data[split:]), data, split, seed=5)
ldev = tape.make_engine_loci(e3, data)
dev = bpp_amd.Sampler(e3, ldev, data, seed=5)
parent, tau0, thetas = synth.species_tree_arrays(taxa)
for d in (plain, co, dev):
    d.set_species_tree(parent, tau0, thetas)
    d.set_tau_prior(3.0, 3.0 / tau0[-1])
    d.set_theta_prior(2.0, 1000.0, 0.0004)
    d.set_finetune(0.003, 0.004, 0.0002, 0.02)
    d.set_subst_moves(0.3, 0.4, 0.8, 1.0, 1.0)
for i, d in enumerate(data):
    plain.set_subst_model(i, list(d["freqs"]), list(d["exch"]), 0.5, R)
    co.set_subst_model(i, list(d["freqs"]), list(d["exch"]), 0.5, R)
    dev.set_subst_model(i, d["freqs"], d["exch"], 0.5)
co.set_threads(8); plain.set_threads(8)
plain.initialize(); co.initialize(); dev.initialize()
print("device sampler kind", dev.kind(), "streams", dev.streams())
for b in range(blocks):
    for _ in range(per):
        plain.iterate(); co.iterate()
    dev.iterate(per)
    s = dev.summary(); hp, ha, _ = plain.counters(); cp, ca, _ = co.counters()
    same_co = (hp, ha) == (cp, ca) and plain.taus() == co.taus() and plain.thetas() == co.thetas() and plain.total_lnl() == co.total_lnl()
    same_dev = (s["proposals"], s["accepted"]) == (hp, ha) and bool(np.allclose(dev.taus(), plain.taus(), rtol=1e-10, atol=0)) \
        and bool(np.allclose(dev.thetas(), plain.thetas(), rtol=1e-10, atol=0))
    bad = worst = 0
    for i in range(nloci):
        a, h, c = dev.tree(i), plain.tree(i), co.tree(i)
        if h != c:
            bad += 1
        if [int(x) for x in a["parent"]] != list(h["parent"]) or not np.allclose(a["time"], h["time"], rtol=1e-10, atol=0):
            bad += 1
        if i % 10 == 0:
            have = ldev[i].root_loglikelihood(int(a["clv"][a["root"]]), -1)
            worst = max(worst, abs(have - a["lnl"]) / abs(have))
    print(f"block {b}: {per*(b+1)} iterations, proposals {s['proposals']} accepted {s['accepted']}; cohorts == plain: {same_co}; device == plain: {same_dev}; "
          f"loci that differ: {bad}; device lnL vs its buffers, worst rel: {worst:.1e}; lnL {s['total_lnl']:.4f} / {plain.total_lnl():.4f}", flush=True)
    assert same_co and same_dev and bad == 0 and worst < 1e-11
print("soak ok")
