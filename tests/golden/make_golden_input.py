"""Generates the input-side golden fixtures (run in the build container, where /root/reference and
oracle/_ref/libbppref.so exist):

    python tests/golden/make_golden_input.py

  tests/golden/frogs/frogs.txt, frogs.Imap.txt   data files the reference's tests hold
                                                 (test/testbed/small/common-data), copied verbatim
  tests/golden/phylip_quirks.phy                 our own PHYLIP file exercising the reader's corners
  tests/golden/input_pipeline.json               what the REAL reference (through oracle/ref_shim_input.c)
                                                 makes of them at every stage of method.c:3299-3672
  tests/golden/frogs/ref_program.compressed-aln.phy
                                                 written by the unmodified reference program (oracle/_ref/bpp)
"""
import ctypes as C
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_DATA = "/root/reference/test/testbed/small/common-data"
SPECIES = ["K", "C", "L", "H"]

QUIRKS = (
    "\n \t \n"
    "  3 12\n"
    "\n"
    "one^a   ACGTAC GTACGT\n"
    "two^b\tAC-T?C\r\n"
    "  NNacgt \r\n"
    "\n"
    "three^c ACGTAC!!@@GT\n"                           # '!' '@' are dropped
    "RYKM\n"
    "\n\n"
    "2 8\n"
    "x^a\tjoACGTjoACGT\n"                              # lower-case j and o are dropped
    "y^b ACGTACGT\n"
    "   \n"
    " 4   5  \n"
    "s1^a AAAAA\n"
    "s2^a ?????\n"
    "s3^b NNN--\n"
    "s4^b ACGTN\n"
    "1 6\n"
    "d^a AC12GT"                                       # digits are legal to the reader; no newline at EOF
)


def shim():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libbppref.so"))
    L.ref_phylip_read.restype = C.c_void_p
    L.ref_phylip_read.argtypes = [C.c_char_p, C.c_long, C.POINTER(C.c_long)]
    for f in ("ref_msa_label", "ref_msa_sequence"):
        getattr(L, f).restype = C.c_char_p
        getattr(L, f).argtypes = [C.c_void_p, C.c_long, C.c_int]
    for f in ("ref_msa_count", "ref_msa_length", "ref_msa_remove_missing", "ref_msa_remove_ambiguous",
              "ref_msa_count_ambiguous"):
        getattr(L, f).restype = C.c_int
        getattr(L, f).argtypes = [C.c_void_p, C.c_long]
    L.ref_msa_set_type.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int]
    L.ref_msa_compress.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.POINTER(C.c_uint)]
    L.ref_imap_read.restype = C.c_void_p
    L.ref_imap_read.argtypes = [C.c_char_p]
    L.ref_imap_count.restype = C.c_long
    L.ref_imap_count.argtypes = [C.c_void_p]
    for f in ("ref_imap_individual", "ref_imap_species"):
        getattr(L, f).restype = C.c_char_p
        getattr(L, f).argtypes = [C.c_void_p, C.c_long]
    L.ref_diploid_resolve.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.POINTER(C.c_char_p),
                                      C.POINTER(C.c_uint)]
    L.ref_resolution_count.argtypes = [C.c_long, C.c_long, C.POINTER(C.c_ulong)]
    L.ref_msa_compress_diploid.argtypes = [C.c_void_p, C.c_long, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_ulong)]
    L.ref_msa_print_phylip.argtypes = [C.c_char_p, C.c_void_p, C.c_long]
    for f in ("ref_map_fasta", "ref_map_amb", "ref_map_nt_missing", "ref_map_aa_missing"):
        getattr(L, f).restype = C.POINTER(C.c_uint)
    return L


def snapshot(L, lst, k):
    n, ln = L.ref_msa_count(lst, k), L.ref_msa_length(lst, k)
    return dict(labels=[L.ref_msa_label(lst, k, i).decode() for i in range(n)],
                seqs=[L.ref_msa_sequence(lst, k, i)[:ln].decode("latin-1") for i in range(n)])


def digest(rec):
    h = hashlib.sha256()
    for lab, s in zip(rec["labels"], rec["seqs"]):
        h.update(lab.encode() + b"\0" + s.encode("latin-1") + b"\n")
    return h.hexdigest()


def pipeline(L, path, nloci, jc69, cleandata, phase, imap_path):
    """method.c:3299-3672 through the reference's own functions"""
    n = C.c_long(0)
    lst = L.ref_phylip_read(path.encode(), nloci, C.byref(n))
    loci = []
    for k in range(n.value):
        L.ref_msa_set_type(lst, k, 0, 0 if jc69 else 7)           # BPP_DATA_DNA, JC69 / GTR
        raw = snapshot(L, lst, k)
        rec = dict(count=len(raw["labels"]), length=len(raw["seqs"][0]), labels=raw["labels"], raw_sha256=digest(raw))
        rec["removed"] = L.ref_msa_remove_missing(lst, k)
        if rec["removed"] < 0:
            rec["a1"] = None
            loci.append(rec)
            continue
        if cleandata:
            rec["clean_ok"] = L.ref_msa_remove_ambiguous(lst, k)
            rec["clean"] = snapshot(L, lst, k)["seqs"]
        else:
            rec["ambiguous_sites"] = L.ref_msa_count_ambiguous(lst, k)
        w = (C.c_uint * L.ref_msa_length(lst, k))()
        np_ = L.ref_msa_compress(lst, n.value, k, int(jc69), w)
        a1 = snapshot(L, lst, k)
        rec["a1"] = dict(labels=a1["labels"], seqs=a1["seqs"], weights=list(w[:np_]))
        loci.append(rec)
    out = dict(loci=loci)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "a1.phy")
        if all(r["a1"] for r in loci):
            L.ref_msa_print_phylip(p.encode(), lst, n.value)
            out["a1_phylip"] = open(p).read()
    if phase is not None:
        im = L.ref_imap_read(imap_path.encode())
        out["imap"] = [[L.ref_imap_individual(im, i).decode(), L.ref_imap_species(im, i).decode()]
                       for i in range(L.ref_imap_count(im))]
        sp = (C.c_char_p * len(SPECIES))(*[s.encode() for s in SPECIES])
        ph = (C.c_uint * len(SPECIES))(*phase)
        assert L.ref_diploid_resolve(lst, n.value, im, len(SPECIES), sp, ph)
        for k, rec in enumerate(loci):
            n1 = len(rec["a1"]["weights"])
            rc = (C.c_ulong * n1)()
            L.ref_resolution_count(k, n1, rc)
            n2 = L.ref_msa_length(lst, k)
            a2 = snapshot(L, lst, k)
            w3 = (C.c_uint * n2)()
            mp = (C.c_ulong * n2)()
            n3 = L.ref_msa_compress_diploid(lst, k, int(jc69), w3, mp)
            a3 = snapshot(L, lst, k)
            rec["resolution_count"] = list(rc)
            rec["a2"] = dict(labels=a2["labels"], seqs=a2["seqs"])
            rec["a3"] = dict(seqs=a3["seqs"], weights=list(w3[:n3]))
            rec["mapping"] = list(mp)
    return out


def main():
    L = shim()
    fdir = os.path.join(HERE, "frogs")
    os.makedirs(fdir, exist_ok=True)
    for f in ("frogs.txt", "frogs.Imap.txt"):
        shutil.copyfile(os.path.join(REF_DATA, f), os.path.join(fdir, f))
        os.chmod(os.path.join(fdir, f), 0o644)
    qpath = os.path.join(HERE, "phylip_quirks.phy")
    with open(qpath, "w", newline="") as f:
        f.write(QUIRKS)
    frogs, imap = os.path.join(fdir, "frogs.txt"), os.path.join(fdir, "frogs.Imap.txt")
    out = dict(
        tables={k: list(getattr(L, "ref_map_" + k)()[:256]) for k in ("fasta", "amb", "nt_missing", "aa_missing")},
        species=SPECIES,
        frogs_jc69_phased=pipeline(L, frogs, 0, True, False, [1, 1, 1, 1], imap),
        frogs_gtr_phased=pipeline(L, frogs, 0, False, False, [1, 1, 1, 1], imap),
        frogs_jc69_halfphased=pipeline(L, frogs, 3, True, False, [1, 0, 1, 0], imap),
        frogs_jc69_clean=pipeline(L, frogs, 0, True, True, None, None),
        frogs_gtr=pipeline(L, frogs, 2, False, False, None, None),
        quirks=pipeline(L, qpath, 0, False, False, None, None),
    )
    with open(os.path.join(HERE, "input_pipeline.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    # the unmodified program on the same files
    bpp = os.path.join(ROOT, "oracle", "_ref", "bpp")
    with tempfile.TemporaryDirectory() as td:
        for f in ("frogs.txt", "frogs.Imap.txt"):
            shutil.copyfile(os.path.join(fdir, f), os.path.join(td, f))
        with open(os.path.join(td, "A00.ctl"), "w") as f:
            f.write("seed = 1\nseqfile = frogs.txt\nImapfile = frogs.Imap.txt\njobname = out\n"
                    "speciesdelimitation = 0\nspeciestree = 0\n"
                    "species&tree = 4  K  C  L  H\n                  9  7 14  2\n                 (((K, C), L), H);\n"
                    "phase = 1 1 1 1\nusedata = 1\nnloci = 5\ncleandata = 0\n"
                    "thetaprior = gamma 2 2000\ntauprior = gamma 2 1000\nfinetune = 1\nprint = 1 0 0 0\n"
                    "burnin = 4\nsampfreq = 1\nnsample = 4\n")
        subprocess.run([bpp, "--cfile", "A00.ctl"], cwd=td, check=True, stdout=subprocess.DEVNULL)
        shutil.copyfile(os.path.join(td, "out.compressed-aln.phy"), os.path.join(fdir, "ref_program.compressed-aln.phy"))
        txt = open(os.path.join(td, "out.txt")).read()
        a = txt.index("COMPRESSED ALIGNMENTS AFTER PHASING OF DIPLOID SEQUENCES")
        b = txt.index("Summary of alignments *after* phasing sequences")
        with open(os.path.join(fdir, "ref_program.phased-aln.phy"), "w") as f:
            f.write(txt[a:b].split("\n", 2)[2])
    print("wrote", os.path.getsize(os.path.join(HERE, "input_pipeline.json")), "bytes of JSON")


if __name__ == "__main__":
    sys.exit(main())
