"""Drop-in check: the same C host driver (same seeds) on libbpp_amd.so and on the real
reference's locus API walks the same trajectory — same accept/reject decisions, same trees,
log-likelihoods equal to 1e-12."""
import numpy as np
import pytest

import bpp_amd
from bpp_amd import synth
import oraclelib as O
import hostdrv
import tape
from common import rel

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref did not travel")
@pytest.mark.parametrize("taxa,model,R,scaling,nloci", [(4, "jc69", 1, False, 200), (8, "gtr", 4, False, 40),
                                                      (6, "jc69", 2, True, 30)])
def test_same_trajectory_on_gpu_and_reference(engine, taxa, model, R, scaling, nloci):
    data = synth.make_dataset(nloci, 400, taxa, model, R, seed=5, theta=0.004 if taxa == 6 else None)
    loci = tape.make_engine_loci(engine, data, scaling)
    g = hostdrv.hip_driver(engine, loci, data, seed=11, scaling=scaling)
    r = hostdrv.reference_driver(data, seed=11, scaling=scaling)
    parent, tau0, thetas = synth.species_tree_arrays(taxa, 0.004 if taxa == 6 else 0.002)
    for drv in (g, r):
        drv.set_species_tree(parent, tau0, thetas)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        if taxa == 6:
            drv.set_finetune(0.02, 0.02, 0.005, 0.3)
    g.initialize(); r.initialize()
    assert rel(g.total_lnl(), r.total_lnl()) < 1e-13
    for it in range(5):
        g.iterate(); r.iterate()
        assert rel(g.total_lnl(), r.total_lnl()) < 1e-12, it
        assert g.counters() == r.counters(), it            # identical accept/reject history
    assert g.taus() == r.taus() and g.taus() != list(tau0)
    for i in range(nloci):
        a, b = g.tree(i), r.tree(i)
        for key in ("root", "left", "right", "parent", "clv", "pmat", "scaler", "pop"):
            assert a[key] == b[key]
        assert a["time"] == b["time"] and a["logpr"] == b["logpr"]
        assert rel(a["lnl"], b["lnl"]) < 1e-12
    g.close(); r.close()


@pytest.mark.parametrize("taxa,model,R,nloci,split,threads,params", [(4, "jc69", 1, 300, 150, 4, False), (8, "gtr", 4, 50, 17, 1, True),
                                                                 (6, "jc69", 2, 40, 39, 3, False)])
def test_cohorts_on_two_engines_walk_the_plain_drivers_trajectory(taxa, model, R, nloci, split, threads, params):
    """a00_set_cohorts: the loci in two cohorts on two engines, a per-locus step of one proposed and marshalled while the
    other's launch runs (bpa_batch_end_async / bpa_batch_wait), all-loci steps sent to both engines at once — per-locus
    random streams and sums in locus order: the same decisions, trees and parameters as the plain driver on one engine,
    to the bit"""
    data = synth.make_dataset(nloci, 400, taxa, model, R, seed=5, theta=0.004 if taxa == 6 else None)
    e0, e1, e2 = bpp_amd.Engine(0), bpp_amd.Engine(0), bpp_amd.Engine(0)
    plain = hostdrv.hip_driver(e0, tape.make_engine_loci(e0, data), data, seed=11)
    loci = tape.make_engine_loci(e1, data[:split]) + tape.make_engine_loci(e2, data[split:])
    co = hostdrv.hip_driver_cohorts([e1, e2], loci, data, split, seed=11)
    parent, tau0, thetas = synth.species_tree_arrays(taxa, 0.004 if taxa == 6 else 0.002)
    for drv in (plain, co):
        drv.set_species_tree(parent, tau0, thetas)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        drv.set_theta_prior(2.0, 1000.0, 0.001)
        if taxa == 6:
            drv.set_finetune(0.02, 0.02, 0.005, 0.3)
        if params:
            drv.set_subst_moves(0.3, 0.4, 0.8, 1.0, 1.0)
            for i, d in enumerate(data):
                drv.set_subst_model(i, list(d["freqs"]), list(d["exch"]), 0.5, R)
    co.set_threads(threads)
    plain.initialize(); co.initialize()
    assert plain.total_lnl() == co.total_lnl()
    for it in range(4):
        plain.iterate(); co.iterate()
        assert plain.total_lnl() == co.total_lnl(), it
        assert plain.counters()[:2] == co.counters()[:2], it
    assert plain.taus() == co.taus() and plain.thetas() == co.thetas() and co.taus() != list(tau0)
    for i in range(nloci):
        a, b = plain.tree(i), co.tree(i)
        assert a == b, i
        if params:
            fa, fb = plain.get_subst_model(i), co.get_subst_model(i)
            assert all(np.array_equal(x, y) for x, y in zip(fa, fb))
    plain.close(); co.close(); e0.close(); e1.close(); e2.close()


def test_driver_incremental_equals_scratch_on_gpu(engine):
    data = synth.make_dataset(500, 500, 4, "jc69", 1, seed=9)
    loci = tape.make_engine_loci(engine, data)
    g = hostdrv.hip_driver(engine, loci, data, seed=3)
    g.set_species_tree(*synth.species_tree_arrays(4))
    g.initialize()
    for _ in range(3):
        g.iterate()
    props, acc, steps = g.counters()
    assert 0.05 < acc / props < 0.98
    for i in range(0, 500, 25):
        t = g.tree(i)
        d = data[i]
        full = O.OracleLocus(4, 1, d["seqs"], d["weights"]).full_lnl(t["left"], t["right"], t["time"], t["root"])
        assert rel(t["lnl"], full) < 1e-12
        assert rel(t["logpr"], g.logpr(i)) < 1e-12
    g.close()


_msc_start_tree = synth.msc_start_tree


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref did not travel")
def test_anopheles_a00_on_gpu_and_reference(engine):
    """BASELINE config 5's data (examples/anopheles: 100 loci x 12 sequences, 6 species, two sequences each, JC69,
    cleandata = 1; priors and step lengths of anopheles-bpp-msci.ctl) under the MSC on the tree (R,((C,G),((A,Q),L)))
    of the control file: the C host driver over libbpp_amd.so and over the real reference's locus API — same seeds,
    same decisions, same gene trees, same thetas and taus."""
    import json
    import os
    from bpp_amd import seqio
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    species = ["G", "C", "R", "L", "A", "Q"]
    recs = seqio.load_dataset(os.path.join(G, "anopheles", "loci_realign.txt"), os.path.join(G, "anopheles", "Imap.txt"),
                              species, None, model="jc69", cleandata=True)
    parent = [6, 6, 10, 8, 7, 7, 9, 8, 9, 10, -1]
    tau0 = [0.0] * 6 + [0.004, 0.004, 0.008, 0.012, 0.016]
    thetas = [0.02] * 11
    rng = np.random.default_rng(77)
    data = []
    for r in recs:
        left, right, times, root = _msc_start_tree(r["species"], parent, tau0, thetas, rng)
        data.append(dict(seqs=r["seqs"], weights=r["weights"], left=left, right=right, times=times, root=root, states=4,
                         rate_cats=1, model="jc69", rates=np.ones(1)))
    loci = tape.make_engine_loci(engine, data)
    g = hostdrv.hip_driver(engine, loci, data, seed=21)
    r_ = hostdrv.reference_driver(data, seed=21)
    for drv in (g, r_):
        drv.set_species_tree(parent, tau0, thetas)
        for i, r in enumerate(recs):
            drv.set_tip_species(i, r["species"])
        drv.set_tau_prior(2.0, 10.0)                           # tauprior = gamma 2 10
        drv.set_theta_prior(2.0, 100.0, 0.002)                 # thetaprior = gamma 2 100; finetune theta 0.002
        drv.set_finetune(0.003, 0.003, 0.00002, 0.9)           # GBtj, GBspr, tau, mix of the control file
    g.initialize(); r_.initialize()
    assert rel(g.total_lnl(), r_.total_lnl()) < 1e-13
    lnl0 = g.total_lnl()
    for it in range(6):
        g.iterate(); r_.iterate()
        assert rel(g.total_lnl(), r_.total_lnl()) < 1e-12, it
        assert g.counters() == r_.counters(), it
    assert g.taus() == r_.taus() and g.thetas() == r_.thetas()
    assert g.total_lnl() != lnl0 and g.thetas() != thetas
    props, acc, _ = g.counters()
    assert 0.1 < acc / props < 0.95
    for i in range(len(data)):
        a, b = g.tree(i), r_.tree(i)
        for key in ("root", "left", "right", "parent", "clv", "pmat", "pop"):
            assert a[key] == b[key]
        assert a["time"] == b["time"] and a["logpr"] == b["logpr"]
        assert rel(a["lnl"], b["lnl"]) < 1e-12
    g.close(); r_.close()
