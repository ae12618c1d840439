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
