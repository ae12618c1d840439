"""Parity at BASELINE.json's full sizes: (1) EVERY locus of the data set equals the oracle (the C restatement takes
0.3-3 ms per locus: seconds for 10 000); (2) the batched plan equals the single-locus calls bit for bit; (3) after a
whole A00 iteration of incremental proposal steps every locus still equals the oracle's full recompute on its current
tree (the reference's check_logl invariant, method.c:4699-4717); (4) linearity: doubling every pattern weight doubles
every lnL exactly; (5) the device sum equals the sum of the per-locus values."""
import numpy as np
import pytest

import bpp_amd
from bpp_amd import synth
import oraclelib as O
import tape
from common import rel

pytestmark = pytest.mark.gpu


def oracle_full(d, tr=None, scaling=False):
    ol = O.OracleLocus(d["states"], d["rate_cats"], d["seqs"], d["weights"], model=d["model"],
                       freqs=None if d["model"] == "jc69" else d["freqs"],
                       qrates=None if d["model"] == "jc69" else d["exch"], rates=d["rates"], scaling=scaling)
    if tr is None:
        return ol.full_lnl(d["left"], d["right"], d["times"], d["root"])
    return ol.full_lnl(tr.left, tr.right, tr.time, tr.root)


@pytest.mark.parametrize("name,nloci,sites,taxa,model,R,taus", [
    ("C2", 10000, 1000, 4, "jc69", 1, (0.001, 0.002, 0.003)),
    ("C3", 10000, 1000, 8, "gtr", 4, (0.0011, 0.0025, 0.005)),
    ("C4", 2000, 500, 6, "lg", 4, (0.01, 0.015, 0.02, 0.035, 0.05)),
])
def test_full_size_properties(name, nloci, sites, taxa, model, R, taus):
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(nloci, sites, taxa, model, R, seed=777, divergence=3.0 if name == "C4" else 1.0)     # (C4: the bench's data, ~190 patterns per locus)
    loci = tape.make_engine_loci(eng, data)
    sch = tape.make_schedule(data, seed=11, taus=tuple(t*(3.0 if name == "C4" else 1.0) for t in taus))
    sch.keep_records = False                                                  # (launches only: no replay on the reference's API here)
    init = sch.initial_step()
    p0 = tape.plan_for_step(eng, loci, init)
    p0.enable_sum()
    p0.launch()
    lnl0 = p0.lnl()
    assert np.isfinite(lnl0).all()
    assert rel(p0.lnl_sum(), float(np.sum(lnl0))) < 1e-12                      # (5)
    rng = np.random.default_rng(1)
    sample = rng.choice(nloci, 24, replace=False)
    want0 = np.array([oracle_full(d) for d in data])                          # (1): all loci
    err0 = np.abs(lnl0 - want0)/np.abs(want0)
    assert err0.max() < 1e-13, (int(err0.argmax()), float(err0.max()))
    for li in sample[:8]:                                                     # (2)
        tr = sch.trees[li]
        assert loci[li].root_loglikelihood(tr.clv[tr.root], tr.scaler[tr.root]) == lnl0[li]
    # (3) one whole A00 iteration of incremental proposal steps, then compare with scratch
    steps = sch.iteration()
    last = {}
    for st in steps:
        p = tape.plan_for_step(eng, loci, st)
        p.launch()
        v = p.lnl()
        p.close()
        assert np.isfinite(v).all()
    # every locus on its tree after the iteration: one batched root evaluation (no update: the CLVs are what the
    # incremental steps left) against the oracle's recompute from scratch
    have = tape.root_lnl_all(eng, loci, sch)
    want = np.array([oracle_full(d, sch.trees[li]) for li, d in enumerate(data)])
    err = np.abs(have - want)/np.abs(want)
    assert err.max() < 1e-12, (int(err.argmax()), float(err.max()))
    for li in sample[:8]:
        tr = sch.trees[li]
        assert loci[li].root_loglikelihood(tr.clv[tr.root], tr.scaler[tr.root]) == have[li]
    # (4) linearity in the pattern weights (integers, exact doubling)
    for li in sample[:8]:
        loci[li].set_pattern_weights(np.asarray(data[li]["weights"]) * 2)
    for li in sample[:8]:
        tr = sch.trees[li]
        have = loci[li].root_loglikelihood(tr.clv[tr.root], tr.scaler[tr.root])
        base = oracle_full(data[li], tr)
        assert rel(have, 2 * base) < 1e-12
    p0.close()
    eng.close()
