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
