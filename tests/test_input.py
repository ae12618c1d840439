"""Input side (include/bpp_amd_input.h) against what the REAL reference made of
the same files (tests/golden/input_pipeline.json, written by tests/golden/make_golden_input.py through
oracle/ref_shim_input.c, and the files the unmodified program wrote).  Host code only: runs without a GPU.

Bit-exact contract: sequences as read, pattern ORDER and weights after compression, resolution counts,
the A2->A3 mapping.  The one thing left open by the reference itself is which member of a JC69-merged
class is printed (its multikey quicksort picks pivots with rand(), compress.c:41), so JC69 columns are
compared after renaming their nucleotides in order of first appearance.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import bpp_amd
from bpp_amd import seqio

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
FROGS = os.path.join(G, "frogs", "frogs.txt")
IMAP = os.path.join(G, "frogs", "frogs.Imap.txt")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(G, "input_pipeline.json")) as f:
        return json.load(f)


def columns(seqs):
    return ["".join(s[i] for s in seqs) for i in range(len(seqs[0]))] if seqs else []


def canon(col, jc69):
    """state codes of a column; JC69: unambiguous columns renamed in order of first appearance"""
    nt = bpp_amd.api.map_nt()
    codes = [int(nt[ord(c)]) for c in col]
    if jc69 and all(c in (1, 2, 4, 8, 15) for c in codes):
        ren = {15: 15}
        for c in codes:
            ren.setdefault(c, len(ren))
        codes = [ren[c] for c in codes]
    return tuple(codes)


def same_patterns(got_seqs, want_seqs, jc69):
    g, w = columns(got_seqs), columns(want_seqs)
    assert len(g) == len(w)
    for i, (a, b) in enumerate(zip(g, w)):
        assert canon(a, jc69) == canon(b, jc69), f"pattern {i}: {a} vs {b}"


def digest(labels, seqs):
    h = hashlib.sha256()
    for lab, s in zip(labels, seqs):
        h.update(lab.encode() + b"\0" + s.encode("latin-1") + b"\n")
    return h.hexdigest()


def test_character_tables(gold):
    for name in ("fasta", "amb", "nt_missing", "aa_missing"):
        assert list(seqio.char_table(name)) == gold["tables"][name], name


def test_reader_frogs(gold):
    msas = seqio.read_phylip(FROGS)
    want = gold["frogs_jc69_phased"]["loci"]
    assert len(msas) == len(want) == 5
    for m, w in zip(msas, want):
        assert (m.count, m.length) == (w["count"], w["length"])
        assert m.labels == w["labels"]
        assert digest(m.labels, m.sequences) == w["raw_sha256"]
    assert len(seqio.read_phylip(FROGS, 3)) == 3                      # the 'nloci' cut


def test_reader_quirks(gold):
    """leading blank lines, a label ended by a tab, CRLF, sequences over several lines, digits kept,
    punctuation and lower-case j/o dropped, trailing blanks in the header, all-missing sequences"""
    msas = seqio.read_phylip(os.path.join(G, "phylip_quirks.phy"))
    want = gold["quirks"]["loci"]
    assert len(msas) == len(want) == 4
    for m, w in zip(msas, want):
        assert digest(m.labels, m.sequences) == w["raw_sha256"]
    assert msas[3].sequences == ["AC12GT"]                 # the reader lets digits through (maps.c:185) ...
    with pytest.raises(bpp_amd.api.BpaError):              # ... they are not states: compression refuses them
        msas[3].compress(False)
    for m, w in zip(msas[:3], want[:3]):
        assert m.remove_missing_sequences() == w["removed"]
        assert m.count_ambiguous_sites() == w["ambiguous_sites"]
        wt = m.compress(False)
        assert m.labels == w["a1"]["labels"]
        assert list(wt) == w["a1"]["weights"]
        same_patterns(m.sequences, w["a1"]["seqs"], False)
    assert want[2]["removed"] == 2


@pytest.mark.parametrize("text,msg", [
    ("", "No alignment"),
    ("x 5\n", "Invalid number of sequences in header"),
    ("2 x\n", "Invalid sequence length in header"),
    ("2 4 I\n", "Invalid PHYLIP header"),
    ("2 4\na ACGT\n", "Found 1 sequence(s) but expected 2"),
    ("1 4\na ACG\n", "Sequence 1 (a) has 3 characters but expected 4"),
    ("1 4\na ACGTA\n", "Sequence 1 (a) longer than expected"),
    ("1 4\na AC.T\n", "illegal character '.' on line 2"),
    ("1 4\na AC\x01T\n", "illegal unprintable character 0x01"),
])
def test_reader_errors(tmp_path, text, msg):
    """the reference's messages (phylip.c:52-78, 179-193, 530-604); it dies in fatal(), we raise"""
    p = tmp_path / "bad.phy"
    p.write_bytes(text.encode("latin-1"))
    with pytest.raises(bpp_amd.api.BpaError) as e:
        seqio.read_phylip(p)
    assert msg in str(e.value)


def test_missing_file():
    with pytest.raises(bpp_amd.api.BpaError):
        seqio.read_phylip("/nonexistent/file.phy")
    with pytest.raises(bpp_amd.api.BpaError):
        seqio.Imap("/nonexistent/file.txt")


def test_imap(gold, tmp_path):
    im = seqio.Imap(IMAP)
    assert [list(e) for e in im.entries()] == gold["frogs_jc69_phased"]["imap"]
    sp = gold["species"]
    assert sp[im.species_of("^kiz2305", sp)] == "C" and sp[im.species_of("anything^gs49", sp)] == "K"
    for bad in ("nolabel", "tag^", "x^unknown"):
        with pytest.raises(bpp_amd.api.BpaError):
            im.species_of(bad, sp)
    p = tmp_path / "m.txt"
    p.write_text("# comment\n\n a  K  * trailing comment\n\tb\tC\n* another\n")
    assert seqio.Imap(p).entries() == [("a", "K"), ("b", "C")]
    p.write_text("a K extra\n")
    with pytest.raises(bpp_amd.api.BpaError) as e:
        seqio.Imap(p)
    assert "Invalid entry" in str(e.value) and "line 1" in str(e.value)
    p.write_text("lonely\n")
    with pytest.raises(bpp_amd.api.BpaError):
        seqio.Imap(p)


@pytest.mark.parametrize("key,model,nloci", [("frogs_jc69_phased", "jc69", 0), ("frogs_gtr", "gtr", 2)])
def test_compress_order_and_weights(gold, key, model, nloci):
    recs = seqio.load_dataset(FROGS, model=model, nloci=nloci)
    want = gold[key]["loci"]
    assert len(recs) == len(want)
    for r, w in zip(recs, want):
        assert r["ambiguous_sites"] == w["ambiguous_sites"] and r["removed_sequences"] == w["removed"]
        assert r["original_length"] == w["length"]
        assert list(r["weights"]) == w["a1"]["weights"]
        same_patterns(r["seqs"], w["a1"]["seqs"], model == "jc69")


def test_cleandata(gold):
    recs = seqio.load_dataset(FROGS, model="jc69", cleandata=True)
    for r, w in zip(recs, gold["frogs_jc69_clean"]["loci"]):
        assert w["clean_ok"] == 1
        assert r["original_length"] == len(w["clean"][0])
        assert list(r["weights"]) == w["a1"]["weights"]
        same_patterns(r["seqs"], w["a1"]["seqs"], True)
    # the kept sites come out in the reference's (swapped) order
    m = seqio.read_phylip(FROGS, 1)[0]
    assert m.remove_ambiguous() == 1
    assert m.sequences == gold["frogs_jc69_clean"]["loci"][0]["clean"]
    allamb = seqio.Msa(labels=["a", "b"], seqs=["NNR", "ACG"])
    assert allamb.remove_ambiguous() == 0


@pytest.mark.parametrize("key,model,phase,nloci", [
    ("frogs_jc69_phased", "jc69", [1, 1, 1, 1], 0),
    ("frogs_gtr_phased", "gtr", [1, 1, 1, 1], 0),
    ("frogs_jc69_halfphased", "jc69", [1, 0, 1, 0], 3),
])
def test_diploid_phasing(gold, key, model, phase, nloci):
    """diploid.c:307-647 + compress.c:378-547: resolution counts, expanded labels, A3 patterns in order,
    their weights and the A2->A3 mapping — identical to the reference's"""
    jc = model == "jc69"
    recs = seqio.load_dataset(FROGS, IMAP, gold["species"], phase, model=model, nloci=nloci)
    want = gold[key]["loci"]
    assert len(recs) == len(want)
    for r, w in zip(recs, want):
        d = r["diploid"]
        assert list(d["resolution_count"]) == w["resolution_count"]
        assert list(d["unphased_weights"]) == w["a1"]["weights"]
        assert r["labels"] == w["a2"]["labels"]
        assert list(r["weights"]) == w["a3"]["weights"]
        assert list(d["mapping"]) == w["mapping"]
        same_patterns(r["seqs"], w["a3"]["seqs"], jc)
        assert int(sum(d["resolution_count"])) == len(d["mapping"]) == len(w["a2"]["seqs"][0])


def test_frogs_pattern_counts_of_the_program():
    """the table the unmodified program prints for examples/frogs A00: 26/25/26/18/19
    patterns before and 45/59/102/31/22 after phasing"""
    recs = seqio.load_dataset(FROGS, IMAP, ["K", "C", "L", "H"], [1, 1, 1, 1], model="jc69")
    assert [r["unphased_patterns"] for r in recs] == [26, 25, 26, 18, 19]
    assert [len(r["weights"]) for r in recs] == [45, 59, 102, 31, 22]
    assert [len(r["seqs"]) for r in recs] == [42, 56, 56, 48, 60]


def parse_written(path):
    """[(labels, seqs, weights)] of a file in the msa_print_phylip format"""
    out, lines = [], [ln.rstrip("\n") for ln in open(path)]
    i = 0
    while i < len(lines):
        if not lines[i].strip():
            i += 1
            continue
        n, ln, p = lines[i].split()
        assert p == "P"
        rows = [lines[i + 1 + k].split() for k in range(int(n))]
        w = [int(x) for x in lines[i + 1 + int(n)].split()]
        out.append(([r[0] for r in rows], ["".join(r[1:]) for r in rows], w))
        assert all(len(s) == int(ln) for s in out[-1][1]) and len(w) == int(ln)
        i += int(n) + 2
    return out


def test_writer_against_the_program(gold, tmp_path):
    """<jobname>.compressed-aln.phy as the unmodified reference program wrote it for these files"""
    msas = seqio.read_phylip(FROGS)
    ws = []
    for m in msas:
        m.remove_missing_sequences()
        ws.append(m.compress(True))
    p = tmp_path / "mine.compressed-aln.phy"
    seqio.write_phylip(p, msas, ws)
    mine = parse_written(p)
    ref = parse_written(os.path.join(G, "frogs", "ref_program.compressed-aln.phy"))
    assert len(mine) == len(ref) == 5
    for (la, sa, wa), (lb, sb, wb) in zip(mine, ref):
        assert la == lb and wa == wb
        same_patterns(sa, sb, True)
    # same layout: header, label column width, blocks of ten, blank line after each alignment
    a, b = open(p).read().splitlines(), open(os.path.join(G, "frogs", "ref_program.compressed-aln.phy")).read().splitlines()
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert len(x) == len(y) and [len(t) for t in x.split(" ")] == [len(t) for t in y.split(" ")]
    # GTR: no class has more than one spelling once printed, so the text itself is the reference's
    msas = seqio.read_phylip(FROGS, 2)
    ws = [m.compress(False) for m in msas]
    seqio.write_phylip(p, msas, ws)
    assert open(p).read() == gold["frogs_gtr"]["a1_phylip"]


def test_phased_alignments_against_the_program():
    """the 'COMPRESSED ALIGNMENTS AFTER PHASING' block of the program's output file"""
    recs = seqio.load_dataset(FROGS, IMAP, ["K", "C", "L", "H"], [1, 1, 1, 1], model="jc69")
    ref = parse_written(os.path.join(G, "frogs", "ref_program.phased-aln.phy"))
    assert len(ref) == 5
    for r, (lb, sb, wb) in zip(recs, ref):
        assert r["labels"] == lb and list(r["weights"]) == wb
        same_patterns(r["seqs"], sb, True)
