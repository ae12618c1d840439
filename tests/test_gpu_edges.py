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
