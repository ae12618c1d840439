"""Parity of the HIP path (through the C ABI) with the oracle, the golden vectors
from the reference and — when it travelled — the reference itself.

Bar (BASELINE.json north_star): bit-exact for integer work (scalers, pattern
counts); CLVs bit-identical to the reference's AVX2 back-end when fed the same
P-matrices; P-matrices within a few ulp (device exp/expm1 vs glibc); log-
likelihoods within 1e-10 relative (tolerance LNL_RTOL below)."""
import numpy as np
import pytest

import bpp_amd
from bpp_amd import (GTree, Locus, Plan, OP_DTYPE, locus_update_matrices, locus_update_partials,
                     locus_root_loglikelihood, DATA_DNA, DATA_AA, MODEL_JC69, MODEL_GTR, MODEL_LG)
import oraclelib as O
from common import load_golden, lg_model, rand_tree, rand_seqs, NT, AA, rel

pytestmark = pytest.mark.gpu

LNL_RTOL = 1e-10          # the contract
LNL_RTOL_TIGHT = 1e-13    # what we actually expect
PMAT_ULPS = 8
PMAT_ATOL = 5e-16      # eigen P-matrices: 1-ulp expm1 differences times O(1) eigenvector terms cancel to O(eps) absolute
fh = float.fromhex


def unhex(a, shape=None):
    v = np.array([fh(x) for x in a])
    return v.reshape(shape) if shape else v


def ulps(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.spacing(np.maximum(np.abs(a), np.abs(b))), 1e-320)


def make_locus(engine, S, R, model, seqs, weights, freqs=None, qrates=None, rates=None, scaling=False):
    tips, sites = len(seqs), len(seqs[0])
    inner, edges = tips - 1, 2 * tips - 2
    dtype = DATA_DNA if S == 4 else DATA_AA
    mdl = O.DNA_MODELS[model] if model in O.DNA_MODELS else MODEL_LG
    # buffer counts of method.c:4110-4146
    loc = Locus(engine, dtype, mdl, tips, 2 * inner, S, sites, 1, 2 * edges, R, 2 * inner if scaling else 0)
    for i, s in enumerate(seqs):
        loc.set_tip_states(i, s)
    loc.set_pattern_weights(weights)
    if freqs is not None:
        loc.set_frequencies(0, freqs)
    if qrates is not None:
        loc.set_subst_params(0, qrates)
    if rates is not None:
        loc.set_category_rates(rates)
    return loc


def full_eval(loc, gt):
    """the start-up sequence of method.c:4285-4297"""
    bpp_amd.locus_update_all_matrices(loc, gt)
    bpp_amd.locus_update_all_partials(loc, gt)
    return locus_root_loglikelihood(loc, gt.root)


def golden_case(engine, c):
    S, R = c["states"], c["rate_cats"]
    freqs = qr = None
    if c["model"] == "gtr":
        freqs, qr = unhex(c["freqs"]), unhex(c["qrates"])
    if c["model"] == "lg":
        qr, freqs = lg_model()
    loc = make_locus(engine, S, R, c["model"], c["seqs"], c["weights"], freqs, qr, unhex(c["rates"]),
                     c["scaling"])
    gt = GTree(c["left"], c["right"], unhex(c["times"]), c["root"], scaling=c["scaling"])
    return loc, gt


# ------------------------------------------------------------------ golden vectors
@pytest.mark.parametrize("idx", range(12))
def test_golden_locus(engine, idx):
    c = load_golden("loci.json")[idx]
    S, R, tips = c["states"], c["rate_cats"], c["tips"]
    loc, gt = golden_case(engine, c)
    lnl = full_eval(loc, gt)
    want = fh(c["lnl"])
    assert rel(lnl, want) < LNL_RTOL_TIGHT, (lnl, want)
    # P-matrices: device exp/expm1 vs glibc
    for key, idx_p in (("pmatrix0", 0), ("pmatrix_last", 2 * tips - 3)):
        got, ref = loc.get_pmatrix(idx_p), unhex(c[key], (R, S, S))
        assert ulps(got, ref).max() <= PMAT_ULPS or np.abs(got - ref).max() < PMAT_ATOL
    if c["scaling"]:
        assert list(loc.get_scaler(gt.root.scaler_index)) == c["root_scaler"]     # integers: bit-exact
    if c["model"] != "jc69":
        ev = loc.get_eigen(0)[2]
        assert (ev == unhex(c["eigenvals"])).all()                                 # K6 bit-exact


@pytest.mark.parametrize("idx", range(12))
def test_golden_locus_clv_bit_exact_given_reference_pmatrices(engine, idx):
    """K1 is bit-identical to the reference once both sides use the same P-matrices."""
    c = load_golden("loci.json")[idx]
    S, R = c["states"], c["rate_cats"]
    loc, gt = golden_case(engine, c)
    ol, lnl_o = __import__("test_oracle_pin").oracle_locus(c)   # oracle == reference (pinned bit-exact)
    for nd in gt.branches():
        loc.set_pmatrix(nd.pmatrix_index, ol.pmat[nd.node_index])
    locus_update_partials(loc, gt.postorder())
    for nd in gt.postorder():
        assert (loc.get_clv(nd.clv_index) == ol.clv[nd.node_index]).all()
    assert (loc.get_clv(gt.root.clv_index) == unhex(c["root_clv"], (c["sites"], R, S))).all()
    if c["scaling"]:
        for nd in gt.postorder():
            assert (loc.get_scaler(nd.scaler_index) == ol.scaler[nd.node_index]).all()
    lnl = locus_root_loglikelihood(loc, gt.root)
    assert rel(lnl, fh(c["lnl"])) < 1e-14


def test_golden_k1_vectors(engine):
    """raw pll_core_update_partial_ii vectors incl. scaling transitions"""
    for v in load_golden("k1_vectors.json"):
        S, R, n = v["states"], v["rate_cats"], v["sites"]
        dtype, mdl = (DATA_DNA, MODEL_GTR) if S == 4 else (DATA_AA, MODEL_LG)
        loc = Locus(engine, dtype, mdl, 2, 4, S, n, 1, 4, R, 4)
        l, r = unhex(v["left"], (n, R, S)), unhex(v["right"], (n, R, S))
        loc.set_clv(2, l)
        loc.set_clv(3, r)
        loc.set_pmatrix(0, unhex(v["lmat"], (R, S, S)))
        loc.set_pmatrix(1, unhex(v["rmat"], (R, S, S)))
        # the reference vector as it is: the left child carries the vector's own scale counters
        loc.set_scaler(0, np.array(v["lscaler"], dtype=np.uint32))
        ops = np.array([(4, 2, 2, 0, 0, 3, 1, -1)], dtype=OP_DTYPE)
        loc.update_partials(ops)
        p, ps = O.orc_partial(l, r, unhex(v["lmat"], (R, S, S)), unhex(v["rmat"], (R, S, S)),
                              lscaler=np.array(v["lscaler"], dtype=np.uint32), scaling=True,
                              order=O.ORDER_PAIR if S == 4 else O.ORDER_FMA4)
        assert (loc.get_clv(4) == p).all()
        assert (p == unhex(v["parent"], (n, R, S))).all()
        assert list(loc.get_scaler(2)) == v["pscaler"] and list(ps) == v["pscaler"]
        # and without child scalers (the counters only add)
        ops = np.array([(4, 2, 2, 0, -1, 3, 1, -1)], dtype=OP_DTYPE)
        loc.update_partials(ops)
        p, ps = O.orc_partial(l, r, unhex(v["lmat"], (R, S, S)), unhex(v["rmat"], (R, S, S)),
                              scaling=True, order=O.ORDER_PAIR if S == 4 else O.ORDER_FMA4)
        assert (loc.get_clv(4) == p).all()
        assert (loc.get_scaler(2) == ps).all()
        assert list(ps + np.array(v["lscaler"], dtype=np.uint32)) == v["pscaler"]
        # chained scalers: parent of (4,4) adds both children's counters
        ops = np.array([(5, 3, 4, 0, 2, 4, 1, 2)], dtype=OP_DTYPE)
        loc.update_partials(ops)
        p2, ps2 = O.orc_partial(p, p, unhex(v["lmat"], (R, S, S)), unhex(v["rmat"], (R, S, S)),
                                lscaler=ps, rscaler=ps, scaling=True,
                                order=O.ORDER_PAIR if S == 4 else O.ORDER_FMA4)
        assert (loc.get_clv(5) == p2).all() and (loc.get_scaler(3) == ps2).all()


# ------------------------------------------------------------------ vs oracle, seeded
@pytest.mark.parametrize("spec", [(4, 1, "jc69", 4, 5), (4, 1, "jc69", 4, 6), (4, 4, "gtr", 8, 29),
                                  (4, 2, "jc69", 3, 1), (4, 1, "jc69", 2, 1), (4, 4, "gtr", 16, 257),
                                  (4, 8, "gtr", 7, 300), (20, 4, "lg", 6, 200), (20, 1, "lg", 3, 2),
                                  (20, 4, "lg", 5, 513), (20, 2, "lg", 5, 70), (20, 3, "lg", 4, 129), (20, 3, "lg", 7, 64)])
def test_seeded_vs_oracle(engine, spec):
    S, R, model, tips, sites = spec
    rng = np.random.default_rng(sum(x if isinstance(x, int) else len(x) for x in spec) * 7919)
    seqs = rand_seqs(tips, sites, NT if S == 4 else AA, rng, extra="-NRYK" if S == 4 else "-XBZ")
    w = rng.integers(1, 1000, sites)
    left, right, times, root = rand_tree(tips, rng, 0.02 if S == 4 else 0.3)
    freqs = q = None
    if model == "gtr":
        freqs, q = rng.dirichlet([5] * 4), rng.random(6) + 0.5
    if model == "lg":
        q, freqs = lg_model()
    rates = bpp_amd.compute_gamma_cats(0.5, 0.5, R)
    loc = make_locus(engine, S, R, model, seqs, w, freqs, q, rates)
    gt = GTree(left, right, times, root)
    lnl = full_eval(loc, gt)
    ol = O.OracleLocus(S, R, seqs, w, model=model, freqs=freqs, qrates=q, rates=rates)
    lo = ol.full_lnl(left, right, times, root)
    assert rel(lnl, lo) < LNL_RTOL_TIGHT, (lnl, lo)
    v, ps = locus_root_loglikelihood(loc, gt.root, persite=True)
    _, pso = O.orc_lnl(ol.clv[root], ol.freqs, ol.rw, ol.weights, order=ol.order, persite=True)
    assert np.allclose(ps, pso, rtol=1e-12, atol=0)
    for nd in gt.branches():
        g_, w_ = loc.get_pmatrix(nd.pmatrix_index), ol.pmat[nd.node_index]
        assert ulps(g_, w_).max() <= PMAT_ULPS or np.abs(g_ - w_).max() < PMAT_ATOL


def test_tip_clv_readback(engine):
    loc = make_locus(engine, 4, 2, "jc69", ["ACGT-RN", "TTTTTTT"], [1] * 7)
    assert (loc.get_clv(0) == O.orc_tipclv(4, 2, "ACGT-RN")).all()
    q, f = lg_model()
    loc = make_locus(engine, 20, 1, "lg", ["ARNDX-BZV", "VVVVVVVVV"], [1] * 9, f, q)
    assert (loc.get_clv(0) == O.orc_tipclv(20, 1, "ARNDX-BZV", dna=False)).all()


def test_illegal_state_and_bad_indices(engine):
    loc = Locus(engine, DATA_DNA, MODEL_JC69, 2, 2, 4, 3, 1, 4, 1, 0)
    with pytest.raises(bpp_amd.BpaError, match="Illegal state code"):
        loc.set_tip_states(0, "AC!")
    loc.set_tip_states(0, "ACG")
    loc.set_tip_states(1, "ACT")
    with pytest.raises(bpp_amd.BpaError):
        loc.update_partials(np.array([(9, -1, 0, 0, -1, 1, 1, -1)], dtype=OP_DTYPE))   # clv out of range
    with pytest.raises(bpp_amd.BpaError):
        loc.update_partials(np.array([(2, 0, 0, 0, -1, 1, 1, -1)], dtype=OP_DTYPE))    # no scale buffers
    with pytest.raises(bpp_amd.BpaError):
        loc.update_matrices([7], [0.1])
    with pytest.raises(bpp_amd.BpaError):
        loc.update_matrices([0], [-0.1])
    with pytest.raises(bpp_amd.BpaError):
        Locus(engine, DATA_DNA, 8, 2, 2, 4, 3, 1, 4, 1, 0)      # unknown DNA model


def test_zero_branch_is_identity(engine):
    loc = Locus(engine, DATA_DNA, MODEL_JC69, 2, 2, 4, 2, 1, 4, 2, 0)
    loc.update_matrices([0, 1], [0.0, 1e-101])
    eye = np.broadcast_to(np.eye(4), (2, 4, 4))
    assert (loc.get_pmatrix(0) == eye).all() and (loc.get_pmatrix(1) == eye).all()


def test_eigen_and_library_pmatrix(engine):
    """pll_update_eigen / pll_core_update_pmatrix (library form) on the device"""
    rng = np.random.default_rng(11)
    for S in (4, 20):
        if S == 4:
            f, q = rng.dirichlet([5] * 4), rng.random(6) + 0.5
        else:
            q, f = lg_model()
        ev, iev, evals = engine.update_eigen(f, q, S)
        oev, oiev, oevals = O.orc_eigen(f, q)
        assert (ev == oev).all() and (iev == oiev).all() and (evals == oevals).all()
        rates = bpp_amd.compute_gamma_cats(0.7, 0.7, 4)
        bl = np.array([0.0, 1e-9, 0.01, 0.3, 2.0])
        got = engine.core_update_pmatrix(S, rates, bl, evals, ev, iev)
        for i, t in enumerate(bl):
            want = O.orc_pmatrix_eigen(rates, t, oevals, oev, oiev, library_form=True)
            assert ulps(got[i], want).max() <= PMAT_ULPS or np.abs(got[i] - want).max() < PMAT_ATOL
        assert (got[0] == np.eye(S)).all()
        assert np.allclose(got.sum(axis=-1), 1.0, atol=1e-12)


def test_usedata_off_and_bfbeta(engine):
    loc = make_locus(engine, 4, 1, "jc69", ["ACGT", "ACGA", "ACTT"], [3, 1, 2, 9])
    gt = GTree([-1, -1, -1, 0, 3], [-1, -1, -1, 1, 2], [0, 0, 0, 0.01, 0.02], 4)
    base = full_eval(loc, gt)
    engine.set_options(usedata=1, bfbeta=0.25)
    assert locus_root_loglikelihood(loc, gt.root) == pytest.approx(0.25 * base, rel=1e-15)
    engine.set_options(usedata=0, bfbeta=1.0)
    assert locus_root_loglikelihood(loc, gt.root) == 0.0           # locus.c:2581
    engine.set_options(usedata=1, bfbeta=1.0)
    assert locus_root_loglikelihood(loc, gt.root) == base


def test_diploid_root_loglikelihood(engine):
    """K3 + the phase-resolution averaging of locus.c:2586-2615"""
    rng = np.random.default_rng(3)
    tips, sites = 6, 40
    seqs = rand_seqs(tips, sites, NT, rng, extra="-N")
    left, right, times, root = rand_tree(tips, rng, 0.05)
    # unphased patterns: each maps to 1, 2 or 4 phased patterns
    counts, mapping = [], []
    k = 0
    while k < sites:
        c = int(min(rng.choice([1, 2, 4]), sites - k))
        counts.append(c)
        mapping += list(range(k, k + c))
        k += c
    uw = rng.integers(1, 30, len(counts))
    loc = make_locus(engine, 4, 1, "jc69", seqs, np.ones(sites, dtype=np.uint32))
    loc.set_diploid(counts, mapping, uw)
    gt = GTree(left, right, times, root)
    lnl = full_eval(loc, gt)
    ol = O.OracleLocus(4, 1, seqs, np.ones(sites))
    ol.full_lnl(left, right, times, root)
    lh = O.orc_lhvec(ol.clv[root], ol.freqs, ol.rw)
    want = O.orc_diploid_lnl(lh, counts, mapping, uw)
    assert rel(lnl, want) < LNL_RTOL_TIGHT


# ------------------------------------------------------------------ batched plans
def build_batch(loci, trees, all_nodes=True):
    mat_off, mat_p, mat_l, op_off, ops, root_clv, root_sc = [0], [], [], [0], [], [], []
    for loc, gt in zip(loci, trees):
        for nd in gt.branches():
            mat_p.append(nd.pmatrix_index)
            mat_l.append(bpp_amd.api.branch_length(gt, nd))
        mat_off.append(len(mat_p))
        for nd in gt.postorder():
            ops.append(bpp_amd.api.node_op(nd))
        op_off.append(len(ops))
        root_clv.append(gt.root.clv_index)
        root_sc.append(gt.root.scaler_index)
    return mat_off, mat_p, mat_l, op_off, np.array(ops, dtype=OP_DTYPE), root_clv, root_sc


@pytest.mark.parametrize("S,R,model", [(4, 1, "jc69"), (4, 4, "gtr"), (20, 4, "lg")])
def test_batched_plan_equals_single_locus_and_oracle(engine, S, R, model):
    rng = np.random.default_rng(77 + S + R)
    loci, trees, want = [], [], []
    nloci = 37 if S == 4 else 5
    for i in range(nloci):
        tips = int(rng.integers(2, 9))
        sites = int(rng.integers(1, 70)) if S == 4 else int(rng.integers(1, 300))
        seqs = rand_seqs(tips, sites, NT if S == 4 else AA, rng, extra="-")
        w = rng.integers(1, 1000, sites)
        left, right, times, root = rand_tree(tips, rng, 0.02 if S == 4 else 0.3)
        freqs = q = None
        if model == "gtr":
            freqs, q = rng.dirichlet([5] * 4), rng.random(6) + 0.5
        if model == "lg":
            q, freqs = lg_model()
        rates = bpp_amd.compute_gamma_cats(0.4 + 0.1 * i, 0.4 + 0.1 * i, R)
        loci.append(make_locus(engine, S, R, model, seqs, w, freqs, q, rates, scaling=(i % 2 == 0)))
        trees.append(GTree(left, right, times, root, scaling=(i % 2 == 0)))
        ol = O.OracleLocus(S, R, seqs, w, model=model, freqs=freqs, qrates=q, rates=rates,
                           scaling=(i % 2 == 0))
        want.append(ol.full_lnl(left, right, times, root))
    plan = Plan(engine, loci, *build_batch(loci, trees))
    plan.launch()
    got = plan.lnl()
    for a, b in zip(got, want):
        assert rel(a, b) < LNL_RTOL_TIGHT
    # the single-locus path gives the very same bits (same kernels, same order)
    for loc, gt, a in zip(loci, trees, got):
        assert full_eval(loc, gt) == a
    # relaunch is idempotent; work accounting follows SURVEY §8(d)
    plan.launch()
    assert (plan.lnl() == got).all()
    wk = plan.work()
    nodes = sum(t.inner_count for t in trees)
    assert wk["node_updates"] == nodes
    assert wk["pattern_updates"] == sum(t.inner_count * l.sites for t, l in zip(trees, loci))
    plan.close()


def test_batched_20_state_plan_with_mixed_rate_categories(engine):
    """one plan, 20-state loci with 1, 2, 3 and 4 rate categories and ragged pattern counts (1 ... 260: partial tiles,
    partial staging chunks of the P-matrices, idle waves of a workgroup): oracle, and the single-locus bits"""
    rng = np.random.default_rng(2024)
    q, freqs = lg_model()
    loci, trees, want = [], [], []
    for i, (R, sites) in enumerate([(1, 1), (4, 64), (2, 65), (3, 129), (4, 260), (1, 128), (3, 7), (2, 200)]):
        tips = int(rng.integers(3, 8))
        seqs = rand_seqs(tips, sites, AA, rng, extra="-X")
        w = rng.integers(1, 1000, sites)
        left, right, times, root = rand_tree(tips, rng, 0.3)
        rates = bpp_amd.compute_gamma_cats(0.5 + 0.1 * i, 0.5 + 0.1 * i, R) if R > 1 else np.ones(1)
        scaling = i % 3 == 0
        loci.append(make_locus(engine, 20, R, "lg", seqs, w, freqs, q, rates, scaling=scaling))
        trees.append(GTree(left, right, times, root, scaling=scaling))
        want.append(O.OracleLocus(20, R, seqs, w, model="lg", freqs=freqs, qrates=q, rates=rates,
                                  scaling=scaling).full_lnl(left, right, times, root))
    plan = Plan(engine, loci, *build_batch(loci, trees))
    plan.launch()
    got = plan.lnl()
    for a, b in zip(got, want):
        assert rel(a, b) < LNL_RTOL_TIGHT
    for loc, gt, a in zip(loci, trees, got):
        assert full_eval(loc, gt) == a
    plan.close()


def test_incremental_update_equals_full_recompute(engine):
    """the reference's own invariant (check_logl, method.c:4699-4717): after a proposal
    that re-computes only the root path into the spare buffers, lnL equals a from-scratch
    evaluation to 1e-9 — here to 1e-13."""
    rng = np.random.default_rng(5)
    tips, sites = 8, 64
    seqs = rand_seqs(tips, sites, NT, rng, extra="-")
    w = rng.integers(1, 50, sites)
    left, right, times, root = rand_tree(tips, rng, 0.05)
    loc = make_locus(engine, 4, 1, "jc69", seqs, w)
    gt = GTree(left, right, times, root)
    full_eval(loc, gt)
    inner, edges = tips - 1, 2 * tips - 2
    # age proposal on an inner, non-root node (gtree.c:5439-5467)
    node = next(nd for nd in gt.postorder() if nd.parent is not None and nd.left.left is None)
    lo = max(node.left.time, node.right.time)
    node.time = lo + 0.37 * (node.parent.time - lo)
    changed = [node.left, node.right, node]
    for nd in changed:                                   # SWAP_PMAT_INDEX (locus.c:25)
        nd.pmatrix_index = (nd.pmatrix_index + edges) % (2 * edges)
    locus_update_matrices(loc, gt, changed)
    path = []
    nd = node
    while nd is not None:                                # SWAP_CLV_INDEX (locus.c:24)
        nd.clv_index = tips + (nd.clv_index - tips + inner) % (2 * inner)
        path.append(nd)
        nd = nd.parent
    locus_update_partials(loc, path)
    lnl_inc = locus_root_loglikelihood(loc, gt.root)
    ol = O.OracleLocus(4, 1, seqs, w)
    want = ol.full_lnl(left, right, [n_.time for n_ in gt.nodes], root)
    assert rel(lnl_inc, want) < LNL_RTOL_TIGHT
    # rejection: swap the indices back — the old buffers still hold the old state
    for nd in changed:
        nd.pmatrix_index = (nd.pmatrix_index + edges) % (2 * edges)
    for nd in path:
        nd.clv_index = tips + (nd.clv_index - tips + inner) % (2 * inner)
    node.time = times[node.node_index]
    lnl_old = locus_root_loglikelihood(loc, gt.root)
    assert rel(lnl_old, O.OracleLocus(4, 1, seqs, w).full_lnl(left, right, times, root)) < LNL_RTOL_TIGHT


needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref did not travel")


@needs_ref
@pytest.mark.parametrize("spec", [(4, 1, "jc69", 4, 6), (4, 4, "gtr", 8, 30), (20, 4, "lg", 6, 120)])
def test_against_the_reference_itself(engine, spec):
    """same inputs through the reference's own locus API (AVX2) on this box's CPU"""
    S, R, model, tips, sites = spec
    rng = np.random.default_rng(sum(x if isinstance(x, int) else len(x) for x in spec) * 7919)
    seqs = rand_seqs(tips, sites, NT if S == 4 else AA, rng, extra="-")
    w = rng.integers(1, 1000, sites)
    left, right, times, root = rand_tree(tips, rng, 0.02 if S == 4 else 0.3)
    freqs = q = None
    if model == "gtr":
        freqs, q = rng.dirichlet([5] * 4), rng.random(6) + 0.5
    if model == "lg":
        q, freqs = O.lg_model()
    rl = O.RefLocus(S, R, seqs, w, model=model, freqs=freqs, qrates=q, alpha=0.5 if R > 1 else None)
    rl.set_tree(left, right, times, root)
    want = rl.full_lnl()
    loc = make_locus(engine, S, R, model, seqs, w, freqs, q, rl.rates())
    gt = GTree(left, right, times, root)
    got = full_eval(loc, gt)
    assert rel(got, want) < LNL_RTOL_TIGHT
    rl.free()


@pytest.mark.parametrize("kernel", ["waverl", "wave2", "pipe", "pipemfma", "tiled", "generic"])
def test_20_state_kernel_variants_are_bit_exact(engine, monkeypatch, kernel):
    """every 20-state node-update kernel kept next to the default (BPA_S20_KERNEL: the FP64-MFMA one, the one-wave
    fall-back of loci with more than 4 categories, the unstaged one) reproduces the reference's AVX2 summation order
    exactly: same CLVs, scalers and lnL bits as the default pipelined kernel"""
    c = load_golden("loci.json")
    for idx in (6, 7, 11):                      # the three 20-state golden loci (11: with scaling)
        case = c[idx]
        S, R = case["states"], case["rate_cats"]
        ol, _ = __import__("test_oracle_pin").oracle_locus(case)
        __import__("common").skip_unless_experimental(kernel)
        monkeypatch.setenv("BPA_S20_KERNEL", kernel)
        loc, gt = golden_case(engine, case)
        for nd in gt.branches():
            loc.set_pmatrix(nd.pmatrix_index, ol.pmat[nd.node_index])
        locus_update_partials(loc, gt.postorder())
        lnl_mfma = locus_root_loglikelihood(loc, gt.root)
        for nd in gt.postorder():
            assert (loc.get_clv(nd.clv_index) == ol.clv[nd.node_index]).all()
            if case["scaling"]:
                assert (loc.get_scaler(nd.scaler_index) == ol.scaler[nd.node_index]).all()
        monkeypatch.delenv("BPA_S20_KERNEL")
        loc2, gt2 = golden_case(engine, case)
        for nd in gt2.branches():
            loc2.set_pmatrix(nd.pmatrix_index, ol.pmat[nd.node_index])
        locus_update_partials(loc2, gt2.postorder())
        assert locus_root_loglikelihood(loc2, gt2.root) == lnl_mfma


@pytest.mark.parametrize("model", ["k80", "f81", "hky", "t92", "tn93", "f84"])
def test_closed_form_dna_models(engine, model):
    """K7 (locus.c:1981-2323) on the device vs the oracle (which equals the reference bit for bit)"""
    rng = np.random.default_rng(len(model) * 31 + ord(model[0]))
    tips, sites, R = 7, 90, 4
    seqs = rand_seqs(tips, sites, NT, rng, extra="-NRY")
    w = rng.integers(1, 500, sites)
    left, right, times, root = rand_tree(tips, rng, 0.2)
    freqs = rng.dirichlet([5] * 4)
    q = np.concatenate([rng.random(3) + 0.5, np.ones(3)])
    rates = bpp_amd.compute_gamma_cats(0.6, 0.6, R)
    loc = make_locus(engine, 4, R, model, seqs, w, freqs, q, rates)
    gt = GTree(left, right, times, root)
    lnl = full_eval(loc, gt)
    ol = O.OracleLocus(4, R, seqs, w, model=model, freqs=freqs, qrates=q, rates=rates)
    assert rel(lnl, ol.full_lnl(left, right, times, root)) < LNL_RTOL_TIGHT
    for nd in gt.branches():
        g_, w_ = loc.get_pmatrix(nd.pmatrix_index), ol.pmat[nd.node_index]
        assert ulps(g_, w_).max() <= PMAT_ULPS or np.abs(g_ - w_).max() < PMAT_ATOL
        assert np.allclose(g_.sum(-1), 1.0, atol=1e-14)


def test_concurrent_calls_on_different_loci(engine):
    """the reference's threading contract (threads.c:87-200): calls for different
    loci may come concurrently from different host threads"""
    import threading
    rng = np.random.default_rng(8)
    cases = []
    for i in range(16):
        tips, sites = int(rng.integers(3, 9)), int(rng.integers(5, 80))
        seqs = rand_seqs(tips, sites, NT, rng, extra="-")
        w = rng.integers(1, 100, sites)
        left, right, times, root = rand_tree(tips, rng, 0.05)
        loc = make_locus(engine, 4, 1, "jc69", seqs, w)
        gt = GTree(left, right, times, root)
        want = O.OracleLocus(4, 1, seqs, w).full_lnl(left, right, times, root)
        cases.append((loc, gt, want))
    errs = []

    def work(chunk):
        try:
            for _ in range(20):
                for loc, gt, want in chunk:
                    got = full_eval(loc, gt)
                    if rel(got, want) > LNL_RTOL_TIGHT:
                        errs.append((got, want))
        except Exception as ex:          # noqa: BLE001
            errs.append(ex)
    ts = [threading.Thread(target=work, args=(cases[i::4],)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs[:3]


def test_concurrent_callers_on_different_loci(engine):
    """SURVEY section 8b, threading: BPP's worker threads call the locus API for DIFFERENT loci concurrently
    (threads.c:87-200).  Eight threads hammer their own loci through the single-locus calls; every result must
    equal the serial one (the engine serialises internally; ctypes releases the GIL around each call)."""
    import threading
    rng = np.random.default_rng(17)
    cases = []
    for i in range(32):
        tips, sites = int(rng.integers(3, 9)), int(rng.integers(5, 60))
        seqs = rand_seqs(tips, sites, NT, rng, extra="-N")
        w = rng.integers(1, 50, sites)
        model = "jc69" if i % 2 else "gtr"
        fr = None if model == "jc69" else rng.dirichlet(np.full(4, 10.0))
        qr = None if model == "jc69" else np.append(np.exp(rng.normal(0, 0.3, 5)), 1.0)
        R = 1 if model == "jc69" else 4
        rates = None if R == 1 else bpp_amd.compute_gamma_cats(0.7, 0.7, 4)
        loc = make_locus(engine, 4, R, model, seqs, w, fr, qr, rates)
        trees = [GTree(*rand_tree(tips, rng, 0.02)) for _ in range(6)]
        cases.append((loc, trees))
    serial = [[full_eval(loc, gt) for gt in trees] for loc, trees in cases]
    out = [[None] * 6 for _ in cases]
    errs = []

    def worker(k):
        try:
            for rep in range(3):
                for i in range(k, len(cases), 8):
                    loc, trees = cases[i]
                    for j, gt in enumerate(trees):
                        out[i][j] = full_eval(loc, gt)
        except Exception as e:          # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert out == serial
