"""The engine-level packing behind the compact JC69 step records: plans over a subset of the loci,
plans whose loci do not come in slot order (first record format), a plan that outlives a change of the slot numbering,
tip states / weights edited after a plan was built, bpa_batch_evaluate's one-image path, partial sums — all against
full recomputation by the oracle."""
import ctypes as C

import numpy as np
import pytest

import bpp_amd
from bpp_amd import api, synth
import oraclelib as O
import tape
from common import rel

pytestmark = pytest.mark.gpu


def full_step(data, idx):
    """an all-matrices, all-partials step for the loci idx (in that order)"""
    mat_off, mat_pm, mat_len, op_off, ops, root = [0], [], [], [0], [], []
    for i in idx:
        d = data[i]
        tips = len(d["seqs"])
        n = 2 * tips - 1
        parent = [-1] * n
        for k in range(tips, n):
            parent[d["left"][k]] = parent[d["right"][k]] = k
        for k in range(n):
            if parent[k] >= 0:
                mat_pm.append(k if k < d["root"] else k - 1)
                mat_len.append(d["times"][parent[k]] - d["times"][k])
        mat_off.append(len(mat_pm))
        pm = lambda k: k if k < d["root"] else k - 1
        for k in sorted(range(tips, n), key=lambda k: d["times"][k]):
            l, r = d["left"][k], d["right"][k]
            ops.append((k, -1, l, pm(l), -1, r, pm(r), -1))
        op_off.append(len(ops))
        root.append(d["root"])
    return mat_off, mat_pm, mat_len, op_off, np.array(ops, dtype=api.OP_DTYPE), root


def oracle_lnl(d, weights=None, seqs=None):
    ol = O.OracleLocus(4, 1, seqs or d["seqs"], d["weights"] if weights is None else weights)
    return ol.full_lnl(d["left"], d["right"], d["times"], d["root"])


def make_plan(eng, loci, data, idx):
    mo, mp, ml, oo, ops, root = full_step(data, idx)
    return bpp_amd.Plan(eng, [loci[i] for i in idx], mo, mp, ml, oo, ops, root)


def test_subsets_orders_and_repacking():
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(700, 300, 4, "jc69", 1, seed=21) + synth.make_dataset(40, 300, 8, "jc69", 1, seed=22)
    loci = tape.make_engine_loci(eng, data)
    want = np.array([oracle_lnl(d) for d in data])
    every = list(range(len(data)))
    for idx in (every, every[::3], every[5:400:7] + every[700:], [3], every[::-1][:50], [10, 4, 300]):
        p = make_plan(eng, loci, data, idx)
        n = p.enable_partial_sums()
        p.launch()
        got = p.lnl()
        assert np.all(np.abs(got - want[idx]) <= 1e-13 * np.abs(want[idx])), idx[:5]
        assert n >= 1 and rel(p.lnl_sum(), float(np.sum(got))) < 1e-13
        p.close()
    # partial sums into caller memory: enough room -> one value per workgroup; too little -> the plain total in slot 0
    dev = eng.stage(np.zeros(64))
    p = make_plan(eng, loci, data, every)
    n = p.enable_partial_sums(dev, 64)
    assert 1 < n <= 64
    p.launch()
    assert rel(p.lnl_sum(), float(np.sum(p.lnl()))) < 1e-13
    assert p.enable_partial_sums(dev, 1) == 1
    p.launch()
    assert rel(p.lnl_sum(), float(np.sum(want))) < 1e-12
    p.close()
    # a plan outlives a change of the slot numbering: new loci at the end, then one destroyed
    p_all = make_plan(eng, loci, data, every)
    p_all.launch()
    assert np.all(np.abs(p_all.lnl() - want) <= 1e-13 * np.abs(want))
    extra = synth.make_dataset(3, 300, 4, "jc69", 1, seed=23)
    loci2 = tape.make_engine_loci(eng, extra)
    p_new = make_plan(eng, loci + loci2, data + extra, every + [len(data), len(data) + 2])
    p_new.launch()
    w2 = np.array([oracle_lnl(d) for d in extra])
    assert np.all(np.abs(p_new.lnl()[-2:] - w2[[0, 2]]) <= 1e-13 * np.abs(w2[[0, 2]]))
    p_all.launch()                                            # built on the old numbering
    assert np.all(np.abs(p_all.lnl() - want) <= 1e-13 * np.abs(want))
    api.lib().bpa_locus_destroy(loci2[1].h); loci2[1].h = None                                          # a slot goes away: p_new is on an old numbering too
    p_new.launch()
    assert np.all(np.abs(p_new.lnl()[:len(data)] - want) <= 1e-13 * np.abs(want))
    assert np.all(np.abs(p_new.lnl()[-2:] - w2[[0, 2]]) <= 1e-13 * np.abs(w2[[0, 2]]))
    # tip states and weights edited after the plans were built: every path picks the new values up
    d0 = data[0]
    new_w = np.array(d0["weights"]) * 2 + 1
    loci[0].set_pattern_weights(new_w)
    seqs = list(d0["seqs"])
    seqs[1] = seqs[1][::-1]
    loci[0].set_tip_states(1, seqs[1])
    w0 = oracle_lnl(d0, new_w, seqs)
    for p in (p_all, p_new):
        p.launch()
        assert rel(p.lnl()[0], w0) < 1e-13
        assert np.all(np.abs(p.lnl()[1:len(data)] - want[1:]) <= 1e-13 * np.abs(want[1:]))
    # the one-image path of bpa_batch_evaluate == the plans
    mo, mp, ml, oo, ops, root = full_step(data, every[::2])
    keep = dict(loci=(C.c_void_p * len(every[::2]))(*[loci[i].h for i in every[::2]]), mo=api._u32(mo), mp=api._u32(mp),
                ml=api._f64(ml), oo=api._u32(oo), ops=np.ascontiguousarray(ops), root=api._u32(root),
                rs=np.full(len(root), api.SCALE_BUFFER_NONE, dtype=np.int32))
    b = api.Batch(len(root), keep["loci"], api._up(keep["mo"]), api._up(keep["mp"]), api._dp(keep["ml"]), api._up(keep["oo"]),
                  keep["ops"].ctypes.data_as(C.POINTER(api.Op)), api._up(keep["root"]), keep["rs"].ctypes.data_as(C.POINTER(C.c_int)))
    out = np.zeros(len(root))
    assert api.lib().bpa_batch_evaluate(eng.h, C.byref(b), api._dp(out)), api._err()
    ref = want[every[::2]].copy()
    ref[0] = w0
    assert np.all(np.abs(out - ref) <= 1e-13 * np.abs(ref))
    for p in (p_all, p_new):
        p.close()
    eng.close()


def test_mixed_models_share_the_packing():
    """JC69 and GTR+G4 loci interleaved in one engine: each kind's plans (whole, subsets, out of order) and the
    one-image bpa_batch_evaluate agree with the oracle; a plan that mixes the kinds takes the general path"""
    eng = bpp_amd.Engine(0)
    jc = synth.make_dataset(60, 300, 4, "jc69", 1, seed=31)
    gt = synth.make_dataset(60, 300, 8, "gtr", 4, seed=32)
    data = [d for pair in zip(jc, gt) for d in pair]                   # jc, gtr, jc, gtr, ...
    loci = tape.make_engine_loci(eng, data)

    def oracle(d):
        if d["model"] == "jc69":
            return oracle_lnl(d)
        ol = O.OracleLocus(4, 4, d["seqs"], d["weights"], model="gtr", freqs=d["freqs"], qrates=d["exch"], rates=d["rates"])
        return ol.full_lnl(d["left"], d["right"], d["times"], d["root"])
    want = np.array([oracle(d) for d in data])
    jcs, gts = list(range(0, len(data), 2)), list(range(1, len(data), 2))
    for idx in (jcs, gts, gts[::3], jcs[5:40:2], gts[::-1][:7], jcs + gts, list(range(len(data)))):
        p = make_plan(eng, loci, data, idx)
        n = p.enable_partial_sums()
        p.launch()
        got = p.lnl()
        assert np.all(np.abs(got - want[idx]) <= 1e-12 * np.abs(want[idx])), idx[:4]
        assert n >= 1 and rel(p.lnl_sum(), float(np.sum(got))) < 1e-13
        p.close()
    for idx in (gts, jcs, gts[4:50:5], list(range(len(data)))):
        mo, mp, ml, oo, ops, root = full_step(data, idx)
        keep = dict(loci=(C.c_void_p * len(idx))(*[loci[i].h for i in idx]), mo=api._u32(mo), mp=api._u32(mp), ml=api._f64(ml),
                    oo=api._u32(oo), ops=np.ascontiguousarray(ops), root=api._u32(root),
                    rs=np.full(len(root), api.SCALE_BUFFER_NONE, dtype=np.int32))
        b = api.Batch(len(root), keep["loci"], api._up(keep["mo"]), api._up(keep["mp"]), api._dp(keep["ml"]), api._up(keep["oo"]),
                      keep["ops"].ctypes.data_as(C.POINTER(api.Op)), api._up(keep["root"]), keep["rs"].ctypes.data_as(C.POINTER(C.c_int)))
        out = np.zeros(len(root))
        assert api.lib().bpa_batch_evaluate(eng.h, C.byref(b), api._dp(out)), api._err()
        assert np.all(np.abs(out - want[idx]) <= 1e-12 * np.abs(want[idx]))
    eng.close()


@pytest.mark.parametrize("taxa,R,model", [(8, 4, "gtr"), (8, 2, "gtr"), (4, 4, "gtr"), (6, 3, "gtr")])
def test_update_lists_in_any_order_on_the_multi_category_kernel(taxa, R, model):
    """step_s4_klane_v3_kernel keeps the parents of a step's last three updates with the lane (registers + two LDS words, round
    6) and reads an older one back from HBM: whatever the order of a step's updates — here random linear extensions of the tree's
    partial order, so a child lies 1, 2, 3 ... 6 updates behind its parent, in chunks of every phase — the CLVs are the ones a
    step in age order leaves (bit for bit, read back through the single-locus API) and lnL the oracle's"""
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(90, 400, taxa, model, R, seed=1000 + taxa + R)
    loci = tape.make_engine_loci(eng, data)
    rng = np.random.default_rng(5)
    want = []
    for d in data:
        ol = O.OracleLocus(4, R, d["seqs"], d["weights"], model=d["model"], freqs=d["freqs"], qrates=d["exch"], rates=d["rates"])
        want.append(ol.full_lnl(d["left"], d["right"], d["times"], d["root"]))
    want = np.array(want)

    def step(order_of):
        mat_off, mat_pm, mat_len, op_off, ops, root = [0], [], [], [0], [], []
        for d in data:
            tips = len(d["seqs"]); n = 2 * tips - 1
            parent = [-1] * n
            for k in range(tips, n):
                parent[d["left"][k]] = parent[d["right"][k]] = k
            pm = lambda k: k if k < d["root"] else k - 1
            for k in range(n):
                if parent[k] >= 0:
                    mat_pm.append(pm(k)); mat_len.append(d["times"][parent[k]] - d["times"][k])
            mat_off.append(len(mat_pm))
            for k in order_of(d, tips, n):
                l, r = d["left"][k], d["right"][k]
                ops.append((k, -1, l, pm(l), -1, r, pm(r), -1))
            op_off.append(len(ops))
            root.append(d["root"])
        return bpp_amd.Plan(eng, loci, mat_off, mat_pm, mat_len, op_off, np.array(ops, dtype=api.OP_DTYPE), root)

    def by_age(d, tips, n):
        return sorted(range(tips, n), key=lambda k: d["times"][k])

    def random_extension(d, tips, n):
        done, out = set(range(tips)), []
        while len(out) < n - tips:
            ready = [k for k in range(tips, n) if k not in done and d["left"][k] in done and d["right"][k] in done]
            k = ready[int(rng.integers(len(ready)))]
            out.append(k); done.add(k)
        return out

    p = step(by_age)
    p.launch()
    got = p.lnl()
    assert np.max(np.abs(got - want) / np.abs(want)) < 1e-12
    ref_clv = [[loci[i].get_clv(k).copy() for k in range(taxa, 2 * taxa - 1)] for i in range(0, len(data), 9)]
    p.close()
    for _ in range(6):
        p = step(random_extension)
        p.launch()
        assert (p.lnl() == got).all()                      # same CLVs -> same site terms -> same sums, to the bit
        for j, i in enumerate(range(0, len(data), 9)):
            for a, k in zip(ref_clv[j], range(taxa, 2 * taxa - 1)):
                assert (loci[i].get_clv(k) == a).all()
        p.close()
    eng.close()
