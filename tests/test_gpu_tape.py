"""GPU replay of whole A00 proposal tapes (batched plans with explicit double-buffer
indices, accept/reject toggling) vs the oracle, step by step, and the reference's own
full-recompute invariant (check_logl, method.c:4699-4717) at the end."""
import numpy as np
import pytest

import bpp_amd
from bpp_amd import synth
import oraclelib as O
import tape
from common import rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("taxa,model,R,scaling,nloci", [(4, "jc69", 1, False, 300), (8, "gtr", 4, False, 60),
                                                      (8, "jc69", 2, True, 40)])
def test_tape_gpu_vs_oracle(engine, taxa, model, R, scaling, nloci):
    data = synth.make_dataset(nloci, 400, taxa, model, R, seed=21)
    loci = tape.make_engine_loci(engine, data, scaling)
    sch = tape.make_schedule(data, seed=4, scaling=scaling,
                             taus=(0.001, 0.002, 0.003) if taxa == 4 else (0.0011, 0.0025, 0.005))
    steps = [sch.initial_step()]
    for _ in range(3):
        steps += sch.iteration()
    got = []
    for st in steps:
        p = tape.plan_for_step(engine, loci, st)
        if st.global_decision is not None:
            # the all-loci reduction: as one total (own launch) and as the step kernel's per-workgroup partial sums
            if len(got) % 2:
                p.enable_sum()
            else:
                assert p.enable_partial_sums() >= 1
        p.launch()
        lnl = p.lnl()
        if st.global_decision is not None:
            assert rel(p.lnl_sum(), float(np.sum(lnl))) < 1e-13
        got.append(lnl)
        p.close()
    check = range(nloci) if nloci <= 60 else range(0, nloci, 7)
    for li in check:
        sub = tape.locus_subtape(steps, li)
        want = tape.oracle_replay(data[li], sub, scaling)
        mine = np.array([got[s["step"]][s["task"]] for s in sub])
        assert np.all(np.abs(mine - want) <= 1e-13 * np.abs(want)), (li, np.max(np.abs(mine - want)))
    # final state: incremental result == from-scratch evaluation of the final trees
    for li in list(check)[:10]:
        tr = sch.trees[li]
        d = data[li]
        ol = O.OracleLocus(d["states"], d["rate_cats"], d["seqs"], d["weights"], model=d["model"],
                           freqs=None if model == "jc69" else d["freqs"],
                           qrates=None if model == "jc69" else d["exch"], rates=d["rates"], scaling=scaling)
        want = ol.full_lnl(tr.left, tr.right, tr.time, tr.root)
        have = loci[li].root_loglikelihood(tr.clv[tr.root], tr.scaler[tr.root])
        assert rel(have, want) < 1e-12


@pytest.mark.parametrize("taxa,R,nloci", [(8, 4, 50), (4, 1, 80)])
def test_gtr_tape_with_substitution_parameter_proposals(engine, taxa, R, nloci):
    """the GTR(+Gamma) tape of BASELINE config 3's kind: tree moves plus per-locus frequency / exchangeability / alpha
    proposals (bpa_plan_set_params: batched install + eigensystem refresh on the device), GPU == oracle step by step"""
    data = synth.make_dataset(nloci, 300, taxa, "gtr", R, seed=8)
    loci = tape.make_engine_loci(engine, data)
    sch = tape.make_schedule(data, seed=12, subst=True)
    steps = [sch.initial_step()]
    for _ in range(2):
        steps += sch.iteration()
    allp = tape.plan_for_step(engine, loci, steps[0])          # holds every locus: carrier of the parameter installs
    got = []
    for st in steps:
        tape.apply_params(allp, st)
        p = tape.plan_for_step(engine, loci, st)
        p.launch()
        got.append(p.lnl())
        p.close()
    assert sum(1 for s in steps if s.params) >= 2 * (3 + 5)
    for li in range(0, nloci, 5):
        sub = tape.locus_subtape(steps, li)
        want = tape.oracle_replay(data[li], sub)
        mine = np.array([got[s["step"]][s["task"]] for s in sub])
        assert np.all(np.abs(mine - want) <= 1e-12 * np.abs(want)), (li, np.max(np.abs(mine - want) / np.abs(want)))
    allp.close()


@pytest.mark.parametrize("taxa,scaling,nloci", [(4, False, 700), (8, True, 90), (4, False, 10000)])
def test_chain_launch_equals_step_by_step(taxa, scaling, nloci):
    """bpa_plans_launch sends consecutive per-locus steps out as ONE chain launch (step_jc69_v2_chain_kernel): every
    step's per-locus lnL, and the CLVs / scalers / P-matrices left behind, are the bits of the launches one by one"""
    data = synth.make_dataset(nloci, 500, taxa, "jc69", 1, seed=33)
    runs = []
    sch = tape.make_schedule(data, seed=9, scaling=scaling)           # (the same steps for both runs: indices only)
    steps = [sch.initial_step()]
    for _ in range(2):
        steps += sch.iteration()
    for chained in (False, True):
        eng = bpp_amd.Engine(0)
        loci = tape.make_engine_loci(eng, data, scaling)
        plans = []
        for st in steps:
            p = tape.plan_for_step(eng, loci, st)
            if st.global_decision is not None:
                p.enable_partial_sums()                  # an all-loci step leaves its total as per-workgroup sums: still a link
            plans.append(p)
        if chained:
            eng.enable_timing(True)
            bpp_amd.PlanSequence(plans).launch()
            tm = eng.timing()
            eng.enable_timing(False)
            per_locus = sum(1 for st in steps if st.global_decision is None)
            # chains cover the per-locus steps: fewer launches than steps, every step accounted for
            assert tm["steps"] == len(steps) and tm["launches"] <= len(steps) - per_locus + 2*3 and tm["bytes"] > 0
        else:
            for p in plans:
                p.launch()
        lnl = [p.lnl() for p in plans]
        sums = [p.lnl_sum() for st, p in zip(steps, plans) if st.global_decision is not None]
        tr = sch.trees[0]
        clv = [loci[0].get_clv(len(data[0]["seqs"]) + c) for c in range(2*(taxa - 1))]
        pm = [loci[0].get_pmatrix(c) for c in range(2*(2*taxa - 2))]
        sc = [loci[0].get_scaler(c) for c in range(2*(taxa - 1))] if scaling else []
        runs.append((lnl, clv, pm, sc, loci[0].root_loglikelihood(tr.clv[tr.root], tr.scaler[tr.root]), sums))
        for p in plans:
            p.close()
        eng.close()
    a, b = runs
    assert all(np.array_equal(x, y) for x, y in zip(a[0], b[0]))
    assert all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
    assert all(np.array_equal(x, y) for x, y in zip(a[2], b[2]))
    assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
    assert a[4] == b[4]
    assert len(a[5]) > 0 and a[5] == b[5]            # the all-loci steps' totals: same bits from inside a chain
