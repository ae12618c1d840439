"""Live pin of the input side against the REAL reference (oracle/_ref/libbppref.so through
oracle/ref_shim_input.c) on random alignments rich in heterozygotes, gaps and ambiguity codes — the
tie-breaking of the phasing rounds (diploid.c:427-489) and the pattern order are what this stresses.
Skipped where the reference build is absent (it is test infrastructure, never shipped)."""
import importlib.util
import os

import numpy as np
import pytest

from bpp_amd import seqio

HERE = os.path.dirname(os.path.abspath(__file__))
REFLIB = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libbppref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REFLIB), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def G():
    spec = importlib.util.spec_from_file_location("make_golden_input", os.path.join(HERE, "golden", "make_golden_input.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.L = m.shim()
    return m


def random_phylip(path, rng, nloci, het_rate):
    inds = {"K": ["k1", "k2", "k3"], "C": ["c1", "c2"], "L": ["l1", "l2", "l3"], "H": ["h1"]}
    with open(path, "w") as f:
        for _ in range(nloci):
            names = [i for sp in inds for i in inds[sp] if rng.random() < 0.85] or ["k1"]
            n, ln = len(names), int(rng.integers(20, 120))
            base = rng.choice(list("ACGT"), ln)
            f.write(f"{n} {ln}\n\n")
            for nm in names:
                s = base.copy()
                mut = rng.random(ln) < 0.08
                s[mut] = rng.choice(list("ACGT"), mut.sum())
                het = rng.random(ln) < het_rate
                s[het] = rng.choice(list("RYKMSW"), het.sum())
                odd = rng.random(ln) < 0.02
                s[odd] = rng.choice(list("N-?acgtBDHV"), odd.sum())
                f.write(f"seq^{nm}  {''.join(s)}\n")
            f.write("\n")
    imap = path + ".imap"
    with open(imap, "w") as f:
        for sp, lst in inds.items():
            for i in lst:
                f.write(f"{i}\t{sp}\n")
    return imap


@pytest.mark.parametrize("seed,model,phase,het", [(1, "jc69", [1, 1, 1, 1], 0.02), (2, "gtr", [1, 1, 1, 1], 0.03),
                                                  (3, "jc69", [1, 0, 1, 0], 0.05), (4, "gtr", [0, 1, 0, 1], 0.01),
                                                  (5, "jc69", [1, 1, 1, 1], 0.004)])
def test_pipeline_matches_reference(G, tmp_path, seed, model, phase, het):
    from test_input import same_patterns
    rng = np.random.default_rng(seed)
    path = str(tmp_path / "rand.phy")
    imap = random_phylip(path, rng, 6, het)
    jc = model == "jc69"
    want = G.pipeline(G.L, path, 0, jc, False, phase, imap)["loci"]
    got = seqio.load_dataset(path, imap, G.SPECIES, phase, model=model)
    assert len(got) == len(want) == 6
    for r, w in zip(got, want):
        assert r["ambiguous_sites"] == w["ambiguous_sites"]
        d = r["diploid"]
        assert list(d["unphased_weights"]) == w["a1"]["weights"]
        assert list(d["resolution_count"]) == w["resolution_count"]
        assert r["labels"] == w["a2"]["labels"]
        assert list(r["weights"]) == w["a3"]["weights"]
        assert list(d["mapping"]) == w["mapping"]
        same_patterns(r["seqs"], w["a3"]["seqs"], jc)


def test_frogs_lnl_reference_vs_oracle_on_loader_output(G):
    """the reference's likelihood (locus_root_loglikelihood with its diploid averaging, locus.c:2586-2615) on
    the reference's own phased frogs patterns == the oracle on what OUR loader made of the same files:
    same pattern order, so the sums agree to the last bit"""
    import json
    import ctypes as C
    import oraclelib as O
    from common import rand_tree
    here = os.path.join(HERE, "golden")
    gold = json.load(open(os.path.join(here, "input_pipeline.json")))
    recs = seqio.load_dataset(os.path.join(here, "frogs", "frogs.txt"), os.path.join(here, "frogs", "frogs.Imap.txt"),
                              gold["species"], [1, 1, 1, 1], model="jc69")
    rng = np.random.default_rng(5)
    ulp = C.POINTER(C.c_ulong)
    for r, w in zip(recs, gold["frogs_jc69_phased"]["loci"]):
        left, right, times, root = rand_tree(len(r["seqs"]), rng, 0.01)
        a3 = w["a3"]
        rl = O.RefLocus(4, 1, a3["seqs"], np.ones(len(a3["weights"])))
        rc = np.array(w["resolution_count"], dtype=np.uint64)
        mp = np.array(w["mapping"], dtype=np.uint64)
        uw = np.array(w["a1"]["weights"], dtype=np.uint32)
        rl.L.ref_set_diploid(rl.h, len(rc), rc.ctypes.data_as(ulp), mp.ctypes.data_as(ulp), C.c_ulong(len(mp)),
                             uw.ctypes.data_as(C.POINTER(C.c_uint)))
        rl.set_tree(left, right, times, root)
        want = rl.full_lnl()
        d = r["diploid"]
        ol = O.OracleLocus(4, 1, r["seqs"], np.ones(len(r["weights"])))
        ol.full_lnl(left, right, times, root)
        got = O.orc_diploid_lnl(O.orc_lhvec(ol.clv[root], ol.freqs, ol.rw), d["resolution_count"], d["mapping"],
                                d["unphased_weights"])
        assert got == want, (got, want)
        rl.free()


ANOPH = "/root/reference/examples/anopheles"


@pytest.mark.skipif(not os.path.exists(os.path.join(ANOPH, "loci_realign.txt")), reason="reference examples not present")
def test_anopheles_config5_pipeline_matches_reference(G):
    """BASELINE config 5 (examples/anopheles: 100 loci x 12 sequences, cleandata = 1, JC69): reader -> missing
    sequences -> ambiguous sites cut in the reference's swap order -> compression; labels, kept sites, pattern
    order and weights element by element against the reference's own routines on the same file"""
    from test_input import same_patterns
    path = os.path.join(ANOPH, "loci_realign.txt")
    want = G.pipeline(G.L, path, 0, True, True, None, None)["loci"]
    msas = seqio.read_phylip(path)
    assert len(msas) == len(want) == 100
    npat = []
    for m, w in zip(msas, want):
        assert m.labels == w["labels"] and (m.count, m.length) == (w["count"], w["length"])
        assert m.remove_missing_sequences() == w["removed"]
        assert m.remove_ambiguous() == w["clean_ok"] == 1
        assert m.sequences == w["clean"]
        wt = m.compress(True)
        assert list(wt) == w["a1"]["weights"]
        same_patterns(m.sequences, w["a1"]["seqs"], True)
        npat.append(len(wt))
    assert 3 <= min(npat) and max(npat) <= 40 and 10 < np.mean(npat) < 16       # 3-30, mean 13.2
    species = ["G", "C", "R", "L", "A", "Q"]
    im = seqio.Imap(os.path.join(ANOPH, "Imap.txt"))
    assert sorted(set(species[im.species_of(lab, species)] for lab in msas[0].labels)) == sorted(species)
