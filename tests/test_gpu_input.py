"""End to end from the files the reference ships (frogs: 5 loci, 21-30 unphased diploid sequences,
42-60 tips after phasing) to the likelihood: read -> compress -> phase -> compress -> device loci ->
random gene trees -> lnL, against the oracle fed with the REFERENCE's own phased patterns and tables
(tests/golden/input_pipeline.json).  config 1 of BASELINE.json (examples/frogs A00, JC69, phase = 1 1 1 1)."""
import json
import os

import numpy as np
import pytest

import bpp_amd
from bpp_amd import seqio
from bpp_amd.api import GTree, locus_root_loglikelihood, locus_update_matrices, locus_update_partials

import oraclelib as O
from common import rand_tree, rel

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
FROGS, IMAP = os.path.join(G, "frogs", "frogs.txt"), os.path.join(G, "frogs", "frogs.Imap.txt")
LNL_RTOL = 1e-13


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(G, "input_pipeline.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("key,phase,nloci", [("frogs_jc69_phased", [1, 1, 1, 1], 0), ("frogs_jc69_halfphased", [1, 0, 1, 0], 3)])
def test_frogs_files_to_lnl(engine, gold, key, phase, nloci):
    recs = seqio.load_dataset(FROGS, IMAP, gold["species"], phase, model="jc69", nloci=nloci)
    rng = np.random.default_rng(11)
    for r, w in zip(recs, gold[key]["loci"]):
        loc = seqio.make_locus(engine, r)
        tips = len(r["seqs"])
        left, right, times, root = rand_tree(tips, rng, 0.01)
        gt = GTree(left, right, times, root)
        locus_update_matrices(loc, gt, gt.branches())
        locus_update_partials(loc, gt.postorder())
        got = locus_root_loglikelihood(loc, gt.root)
        # the oracle on what the reference itself made of the file
        a3 = w["a3"]
        ol = O.OracleLocus(4, 1, a3["seqs"], np.ones(len(a3["weights"])))
        ol.full_lnl(left, right, times, root)
        lh = O.orc_lhvec(ol.clv[root], ol.freqs, ol.rw)
        want = O.orc_diploid_lnl(lh, w["resolution_count"], w["mapping"], w["a1"]["weights"])
        assert np.isfinite(want) and want < 0
        assert rel(got, want) < LNL_RTOL, (got, want)


def test_frogs_unphased_gtr(engine, gold):
    """the same files without phasing, GTR: plain pattern weights"""
    recs = seqio.load_dataset(FROGS, model="gtr", nloci=2)
    rng = np.random.default_rng(12)
    freqs, qr = np.array([0.3, 0.2, 0.2, 0.3]), np.array([1.0, 2.0, 1.0, 0.5, 1.5, 1.0])
    for r, w in zip(recs, gold["frogs_gtr"]["loci"]):
        loc = seqio.make_locus(engine, r, model="gtr")
        loc.set_frequencies(0, freqs)
        loc.set_subst_params(0, qr)
        left, right, times, root = rand_tree(len(r["seqs"]), rng, 0.01)
        gt = GTree(left, right, times, root)
        locus_update_matrices(loc, gt, gt.branches())
        locus_update_partials(loc, gt.postorder())
        got = locus_root_loglikelihood(loc, gt.root)
        ol = O.OracleLocus(4, 1, w["a1"]["seqs"], w["a1"]["weights"], model="gtr", freqs=freqs, qrates=qr)
        want = ol.full_lnl(left, right, times, root)
        assert rel(got, want) < LNL_RTOL, (got, want)


def test_anopheles_config5_files_to_lnl(engine):
    """BASELINE config 5 (examples/anopheles: 100 loci x 12 sequences, JC69, cleandata = 1, the settings of
    anopheles-bpp-msci.ctl): files -> cleaned, compressed device loci -> lnL on the fixture's gene trees, against the
    REAL reference's locus_root_loglikelihood on its own patterns (tests/golden/anopheles_pipeline.json)."""
    with open(os.path.join(G, "anopheles_pipeline.json")) as f:
        gold = json.load(f)
    recs = seqio.load_dataset(os.path.join(G, "anopheles", "loci_realign.txt"), os.path.join(G, "anopheles", "Imap.txt"),
                              gold["species"], None, model="jc69", cleandata=True)
    assert len(recs) == len(gold["loci"]) == 100
    total = 0.0
    for r, w in zip(recs, gold["loci"]):
        assert r["labels"] == w["labels"] and list(r["weights"]) == w["weights"]      # bit-exact compressed pattern counts
        assert set(r["species"]) <= set(range(6))
        loc = seqio.make_locus(engine, r)
        t = w["tree"]
        gt = GTree(t["left"], t["right"], [float.fromhex(x) for x in t["times"]], t["root"])
        locus_update_matrices(loc, gt, gt.branches())
        locus_update_partials(loc, gt.postorder())
        got = locus_root_loglikelihood(loc, gt.root)
        want = float.fromhex(w["lnl"])
        assert rel(got, want) < LNL_RTOL, (got, want)
        total += got
    assert rel(total, float.fromhex(gold["total_lnl"])) < 1e-12
