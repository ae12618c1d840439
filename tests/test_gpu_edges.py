"""Edge cases of the batched path: the empty batch, loci of one pattern, loci at and beyond the packing's lane limit
(255 / 256 / 300 patterns), ragged steps (loci with nothing to update next to loci with everything), all against full
recomputation by the oracle."""
import ctypes as C

import numpy as np
import pytest

import bpp_amd
from bpp_amd import api
import oraclelib as O
from common import rand_tree, rel
from test_gpu_packing import full_step, make_plan

pytestmark = pytest.mark.gpu


def jc_locus(rng, tips, np_want):
    """a JC69 locus of exactly np_want distinct patterns (random columns, compressed, cut)"""
    while True:
        cols = rng.integers(0, 4, size=(tips, 6 * np_want + 40))
        seqs = ["".join("ACGT"[c] for c in row) for row in cols]
        pats, w = bpp_amd.compress_site_patterns(seqs, True, True)
        if len(w) >= np_want:
            break
    pats = [p[:np_want] for p in pats]
    w = np.asarray(w[:np_want], dtype=np.uint32) + 1
    left, right, times, root = rand_tree(tips, rng, 0.05)
    return dict(seqs=pats, weights=w, left=left, right=right, times=times, root=root, states=4, rate_cats=1, model="jc69",
                rates=np.ones(1))


def test_empty_batch_fails_loudly(engine):
    b = api.Batch(0, None, None, None, None, None, None, None, None)
    assert not api.lib().bpa_plan_create(engine.h, C.byref(b))
    assert "empty batch" in api._err()
    out = np.zeros(1)
    assert not api.lib().bpa_batch_evaluate(engine.h, C.byref(b), api._dp(out))


def test_pattern_count_limits_and_ragged_steps():
    import tape
    eng = bpp_amd.Engine(0)
    rng = np.random.default_rng(3)
    sizes = [1, 2, 255, 256, 300, 64, 1, 255, 17]
    data = [jc_locus(rng, 8 if n > 30 else 5, n) for n in sizes]
    loci = tape.make_engine_loci(eng, data)
    want = np.array([O.OracleLocus(4, 1, d["seqs"], d["weights"]).full_lnl(d["left"], d["right"], d["times"], d["root"]) for d in data])
    every = list(range(len(data)))
    p = make_plan(eng, loci, data, every)                 # 256 and 300 patterns are beyond the packing: general path
    p.launch()
    assert np.all(np.abs(p.lnl() - want) <= 1e-13 * np.abs(want))
    p.close()
    packed = [i for i in every if sizes[i] < 256]
    p = make_plan(eng, loci, data, packed)                # all on the engine's packing (incl. the 255-lane loci)
    n = p.enable_partial_sums()
    p.launch()
    assert np.all(np.abs(p.lnl() - want[packed]) <= 1e-13 * np.abs(want[packed]))
    assert n >= 2 and rel(p.lnl_sum(), float(np.sum(want[packed]))) < 1e-13
    p.close()
    # ragged: every other locus has nothing to update (its lnL is re-read from the CLVs left by the step above)
    mo, mp, ml, oo, ops, root = [0], [], [], [0], [], []
    for k, i in enumerate(packed):
        a, b_, c, d_, e, f = full_step(data, [i])
        if k % 2 == 0:
            mp += list(b_); ml += list(c); ops += [tuple(x) for x in e]
        mo.append(len(mp)); oo.append(len(ops)); root.append(f[0])
    ops = np.array(ops, dtype=api.OP_DTYPE)
    p = bpp_amd.Plan(eng, [loci[i] for i in packed], mo, mp, ml, oo, ops, root)
    p.launch()
    assert np.all(np.abs(p.lnl() - want[packed]) <= 1e-13 * np.abs(want[packed]))
    p.close()
    eng.close()


def test_all_forms_and_the_lazy_update_api(engine):
    """bpa_locus_update_all_matrices / _all_partials (locus.c:1922, 2523) against the explicit lists, and the lazy
    contract of the single-locus calls: queued work runs with the root term, survives interleaving with another
    locus's calls and with a plan launch, and a later update of the same P-matrix buffer wins"""
    rng = np.random.default_rng(11)
    data = [jc_locus(rng, 6, 9), jc_locus(rng, 5, 30)]
    import tape
    loci = tape.make_engine_loci(engine, data)
    want = [O.OracleLocus(4, 1, d["seqs"], d["weights"]).full_lnl(d["left"], d["right"], d["times"], d["root"]) for d in data]
    trees = [bpp_amd.GTree(d["left"], d["right"], d["times"], d["root"]) for d in data]
    # all-forms (C ABI) == explicit lists (same bits), lengths stored like node->length
    got = []
    for loc, gt in zip(loci, trees):
        bpp_amd.locus_update_all_matrices(loc, gt)
        bpp_amd.locus_update_all_partials(loc, gt)
        a = bpp_amd.locus_root_loglikelihood(loc, gt.root)
        assert all(nd.length == (nd.parent.time - nd.time)*gt.rate_mui for nd in gt.branches())
        bpp_amd.locus_update_matrices(loc, gt, gt.branches())
        bpp_amd.locus_update_partials(loc, gt.postorder())
        assert bpp_amd.locus_root_loglikelihood(loc, gt.root) == a
        got.append(a)
    assert all(rel(a, b) < 1e-13 for a, b in zip(got, want))
    # interleaved: queue on locus 0, then a whole proposal on locus 1, then a plan over both, then locus 0's root term
    l0, g0 = loci[0], trees[0]
    br = g0.branches()
    bad = [10.0*bpp_amd.api.branch_length(g0, nd) + 0.5 for nd in br]
    l0.update_matrices([nd.pmatrix_index for nd in br], bad)                  # overwritten by the next call
    bpp_amd.locus_update_matrices(l0, g0, br)
    bpp_amd.locus_update_partials(l0, g0.postorder())
    bpp_amd.locus_update_all_matrices(loci[1], trees[1])
    bpp_amd.locus_update_all_partials(loci[1], trees[1])
    assert bpp_amd.locus_root_loglikelihood(loci[1], trees[1].root) == got[1]
    p = make_plan(engine, loci, data, [0, 1])
    p.launch()
    assert list(p.lnl()) == got
    p.close()
    assert bpp_amd.locus_root_loglikelihood(l0, g0.root) == got[0]
    # a buffer read flushes the queue: P-matrix of the first branch after a queued update
    l0.update_matrices([br[0].pmatrix_index], [0.125])
    pm = l0.get_pmatrix(br[0].pmatrix_index)
    a = 0.25 + 0.75*np.exp(-4*0.125/3)
    assert abs(pm[0, 0, 0] - a) < 2e-15
    bpp_amd.locus_update_matrices(l0, g0, br)
    # malformed views fail loudly
    v = api._view(g0)
    v.root = 0                                                                # a tip
    assert not api.lib().bpa_locus_update_all_partials(l0.h, C.byref(v))
    assert "bad root" in api._err()
    v = api._view(g0)
    v.nodes = len(g0.nodes) - 2
    assert not api.lib().bpa_locus_update_all_matrices(l0.h, C.byref(v), None)
    assert "2*tips - 1" in api._err()
