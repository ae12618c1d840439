#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs oracle/_ref/libbppref.so, i.e. the
reference compiled in place from /root/reference/src by `make -C oracle`).
Every expected value below is produced by the reference's own code through
oracle/ref_shim.c (AVX2 back-end, PLL_ATTRIB_ARCH_AVX2) — none by this repo's
oracle or kernels.  float64 values are stored as C99 hex strings (bit-exact).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oraclelib as O  # noqa: E402
from common import rand_tree, rand_seqs, NT, AA  # noqa: E402


def hx(a):
    return [float(x).hex() for x in np.asarray(a, dtype=np.float64).ravel()]


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


def locus_case(rng, S, R, model, tips, sites, scaling=False, depth=None, alpha=0.5, arch=O.ARCH_AVX2):
    dna = S == 4
    seqs = rand_seqs(tips, sites, NT if dna else AA, rng, extra="-NRY" if dna else "-XBZ")
    w = rng.integers(1, 60, sites)
    left, right, times, root = rand_tree(tips, rng, depth if depth else (0.02 if dna else 0.3))
    freqs = q = None
    if model == "gtr":
        freqs, q = rng.dirichlet([5] * 4), rng.random(6) + 0.5
    if model == "lg":
        q, freqs = O.lg_model()
    rl = O.RefLocus(S, R, seqs, w, model=model, freqs=freqs, qrates=q,
                    alpha=alpha if R > 1 else None, scaling=scaling, arch=arch)
    rl.set_tree(left, right, times, root)
    lnl = rl.full_lnl()
    case = dict(states=S, rate_cats=R, model=model, tips=tips, sites=sites, scaling=scaling,
                seqs=seqs, weights=[int(x) for x in w], left=left, right=right, times=hx(times),
                root=root, alpha=alpha if R > 1 else None, rates=hx(rl.rates()),
                freqs=None if freqs is None or model == "lg" else hx(freqs),
                qrates=None if q is None or model == "lg" else hx(q),
                lnl=float(lnl).hex(),
                root_clv=hx(rl.clv(root)),
                pmatrix0=hx(rl.pmatrix(0)),
                pmatrix_last=hx(rl.pmatrix(2 * tips - 3)))
    if scaling:
        case["root_scaler"] = [int(x) for x in rl.scaler(root - tips)]
    if model != "jc69":
        ev, iev, evals = rl.eigen()
        case["eigenvals"] = hx(evals)
    rl.free()
    return case


def main():
    if not O.have_ref():
        sys.exit("oracle/_ref/libbppref.so missing: run `make -C oracle` where /root/reference exists")
    rng = np.random.default_rng(20260928)

    q, f = O.lg_model()
    dump("lg_model.json", dict(source="pll_aa_rates_lg / pll_aa_freqs_lg (maps.c:299,868), data table",
                               rates=[float(x) for x in q], freqs=[float(x) for x in f]))

    gam = []
    for a in [0.05, 0.1, 0.3, 0.5, 1.0, 2.3, 10.0, 37.0]:
        for R in [2, 4, 5, 8]:
            rl = O.RefLocus(4, R, ["ACGT"] * 2, [1] * 4, alpha=a)
            gam.append(dict(alpha=a, cats=R, rates=hx(rl.rates())))
            rl.free()
    dump("gamma_cats.json", gam)

    cases = []
    for spec in [(4, 1, "jc69", 4, 6), (4, 1, "jc69", 4, 33), (4, 4, "jc69", 8, 29),
                 (4, 4, "gtr", 8, 31), (4, 1, "gtr", 5, 17), (4, 4, "gtr", 12, 70),
                 (20, 4, "lg", 6, 90), (20, 1, "lg", 4, 11), (4, 1, "jc69", 12, 13)]:
        cases.append(locus_case(rng, *spec))
    # deep trees with per-pattern scaling active (PLL_SCALE_FACTOR = 2^256, bpp.h:376)
    cases.append(locus_case(rng, 4, 1, "jc69", 200, 9, scaling=True, depth=4000.0))
    cases.append(locus_case(rng, 4, 4, "gtr", 180, 7, scaling=True, depth=3000.0))
    cases.append(locus_case(rng, 20, 4, "lg", 90, 5, scaling=True, depth=9000.0))
    dump("loci.json", cases)

    # raw K1 / K2 vectors on random (non-tip) CLVs, incl. scaling transitions
    k1 = []
    for S, R in [(4, 1), (4, 4), (20, 4)]:
        sites = 23
        l, r = rng.random((sites, R, S)), rng.random((sites, R, S))
        l[::3] *= 1e-60
        r[::3] *= 1e-60
        l[1::3] *= 1e-30
        lm, rm = rng.random((R, S, S)), rng.random((R, S, S))
        ls = rng.integers(0, 3, sites).astype(np.uint32)
        p, ps = O.ref_partial(l, r, lm, rm, lscaler=ls, rscaler=None, scaling=True)
        k1.append(dict(states=S, rate_cats=R, sites=sites, left=hx(l), right=hx(r), lmat=hx(lm),
                       rmat=hx(rm), lscaler=[int(x) for x in ls], parent=hx(p),
                       pscaler=[int(x) for x in ps]))
    dump("k1_vectors.json", k1)

    comp = []
    for trial in range(12):
        tips, L = int(rng.integers(2, 7)), int(rng.integers(5, 300))
        seqs = rand_seqs(tips, L, NT, rng, extra="-NRYM", pmut=0.15)
        for jc in (0, 1):
            s2, w = O.ref_compress(seqs, True, jc)
            comp.append(dict(seqs=seqs, jc69=jc, dna=True, patterns=s2, weights=[int(x) for x in w]))
    for trial in range(4):
        tips, L = int(rng.integers(2, 6)), int(rng.integers(5, 200))
        seqs = rand_seqs(tips, L, AA[:6], rng, extra="-X", pmut=0.1)
        s2, w = O.ref_compress(seqs, False, 0)
        comp.append(dict(seqs=seqs, jc69=0, dna=False, patterns=s2, weights=[int(x) for x in w]))
    # the relabel collision quirk (compress.c:293-337): A,C,M merges with A,G,T under JC69
    s2, w = O.ref_compress(["AAAA", "GCCG", "TMMT"], True, 1)
    comp.append(dict(seqs=["AAAA", "GCCG", "TMMT"], jc69=1, dna=True, patterns=s2,
                     weights=[int(x) for x in w]))
    dump("compress.json", comp)


if __name__ == "__main__":
    main()
