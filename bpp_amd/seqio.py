"""ctypes binding of the input side (include/bpp_amd_input.h) and the loader that strings it
together the way BPP's init() does (method.c:3299-3672, 4137-4196):

    read PHYLIP -> drop all-missing sequences -> (cleandata) drop ambiguous sites ->
    compress -> [phase = 1: resolve diploids -> compress again with the A2->A3 mapping] ->
    loci with tip states, pattern weights and the diploid tables set.

Host-side plumbing over libbpp_amd.so; nothing here computes likelihoods.
"""
import ctypes as C
import numpy as np

from . import api

_bound = False

INPUT_EXPORTED = [
    "bpa_map_fasta", "bpa_map_amb", "bpa_map_nt_missing", "bpa_map_aa_missing",
    "bpa_phylip_read", "bpa_msa_list_free", "bpa_msa_create", "bpa_msa_destroy", "bpa_msa_count",
    "bpa_msa_length", "bpa_msa_label", "bpa_msa_sequence", "bpa_msa_remove_missing_sequences",
    "bpa_msa_count_ambiguous_sites", "bpa_msa_remove_ambiguous", "bpa_msa_compress",
    "bpa_imap_read", "bpa_imap_destroy", "bpa_imap_count", "bpa_imap_individual", "bpa_imap_species",
    "bpa_imap_lookup", "bpa_msa_diploid_resolve", "bpa_msa_compress_diploid", "bpa_msa_write_phylip"]


def _lib():
    global _bound
    L = api.lib()
    if _bound:
        return L
    vp, i, ln = C.c_void_p, C.c_int, C.c_long
    up, ulp, cpp = C.POINTER(C.c_uint), C.POINTER(C.c_ulong), C.POINTER(C.c_char_p)
    sig = {
        "bpa_map_fasta": (up, []), "bpa_map_amb": (up, []), "bpa_map_nt_missing": (up, []),
        "bpa_map_aa_missing": (up, []),
        "bpa_phylip_read": (i, [C.c_char_p, ln, C.POINTER(C.POINTER(vp)), C.POINTER(ln)]),
        "bpa_msa_list_free": (None, [C.POINTER(vp), ln]),
        "bpa_msa_create": (vp, [i, i, cpp, cpp]),
        "bpa_msa_destroy": (None, [vp]),
        "bpa_msa_count": (i, [vp]), "bpa_msa_length": (i, [vp]),
        "bpa_msa_label": (C.c_char_p, [vp, i]), "bpa_msa_sequence": (C.c_char_p, [vp, i]),
        "bpa_msa_remove_missing_sequences": (i, [vp, i]),
        "bpa_msa_count_ambiguous_sites": (i, [vp, i]),
        "bpa_msa_remove_ambiguous": (i, [vp]),
        "bpa_msa_compress": (i, [vp, i, i, up]),
        "bpa_imap_read": (vp, [C.c_char_p]), "bpa_imap_destroy": (None, [vp]),
        "bpa_imap_count": (ln, [vp]),
        "bpa_imap_individual": (C.c_char_p, [vp, ln]), "bpa_imap_species": (C.c_char_p, [vp, ln]),
        "bpa_imap_lookup": (i, [vp, C.c_char_p, cpp, i]),
        "bpa_msa_diploid_resolve": (ln, [vp, up, up, ulp]),
        "bpa_msa_compress_diploid": (i, [vp, i, up, ulp]),
        "bpa_msa_write_phylip": (i, [C.c_char_p, C.POINTER(vp), ln, C.POINTER(up), C.POINTER(i)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _bound = True
    return L


def char_table(name):
    """one of the 256-entry character tables: fasta | amb | nt_missing | aa_missing"""
    return np.ctypeslib.as_array(getattr(_lib(), "bpa_map_" + name)(), shape=(256,)).copy()


class Msa:
    """msa_t (bpp.h:821): labels + equal-length sequences; owns the native object."""

    def __init__(self, handle=None, labels=None, seqs=None, dtype=api.DATA_DNA):
        L = _lib()
        self.dtype = dtype
        if handle is None:
            n = len(seqs)
            la = (C.c_char_p * n)(*[s.encode() for s in labels])
            sa = (C.c_char_p * n)(*[s.encode() for s in seqs])
            handle = L.bpa_msa_create(n, len(seqs[0]), la, sa)
            if not handle:
                raise api.BpaError(api._err())
        self.h = handle

    def close(self):
        if self.h:
            _lib().bpa_msa_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def count(self):
        return _lib().bpa_msa_count(self.h)

    @property
    def length(self):
        return _lib().bpa_msa_length(self.h)

    @property
    def labels(self):
        L = _lib()
        return [L.bpa_msa_label(self.h, i).decode() for i in range(self.count)]

    @property
    def sequences(self):
        L = _lib()
        n = self.length
        return [L.bpa_msa_sequence(self.h, i)[:n].decode("latin-1") for i in range(self.count)]

    def remove_missing_sequences(self):
        return _lib().bpa_msa_remove_missing_sequences(self.h, self.dtype)

    def count_ambiguous_sites(self):
        return _lib().bpa_msa_count_ambiguous_sites(self.h, self.dtype)

    def remove_ambiguous(self):
        return _lib().bpa_msa_remove_ambiguous(self.h)

    def compress(self, jc69):
        w = np.zeros(self.length, dtype=np.uint32)
        n = _lib().bpa_msa_compress(self.h, self.dtype, int(jc69), api._up(w))
        api._chk(n)
        return w[:n].copy()

    def diploid_resolve(self, diploid, weights):
        d = api._u32(diploid)
        w = api._u32(weights)
        rc = np.zeros(self.length, dtype=np.uint64)
        n = _lib().bpa_msa_diploid_resolve(self.h, api._up(d), api._up(w), rc.ctypes.data_as(C.POINTER(C.c_ulong)))
        api._chk(n)
        return rc

    def compress_diploid(self, jc69):
        n2 = self.length
        w = np.zeros(n2, dtype=np.uint32)
        mp = np.zeros(n2, dtype=np.uint64)
        n = _lib().bpa_msa_compress_diploid(self.h, int(jc69), api._up(w), mp.ctypes.data_as(C.POINTER(C.c_ulong)))
        api._chk(n)
        return w[:n].copy(), mp


def read_phylip(path, max_loci=0, dtype=api.DATA_DNA):
    """phylip_parse_multisequential (phylip.c:622): list of Msa"""
    L = _lib()
    arr = C.POINTER(C.c_void_p)()
    n = C.c_long(0)
    api._chk(L.bpa_phylip_read(str(path).encode(), max_loci, C.byref(arr), C.byref(n)))
    out = [Msa(handle=arr[k], dtype=dtype) for k in range(n.value)]
    L.bpa_msa_list_free(arr, 0)              # the array only; the alignments now belong to the Msa objects
    return out


def write_phylip(path, msas, weights):
    """msa_print_phylip (msa.c:109): the <jobname>.compressed-aln.phy format"""
    L = _lib()
    n = len(msas)
    hs = (C.c_void_p * n)(*[m.h for m in msas])
    ws = [api._u32(w) for w in weights]
    wp = (C.POINTER(C.c_uint) * n)(*[api._up(w) for w in ws])
    dt = (C.c_int * n)(*[m.dtype for m in msas])
    api._chk(L.bpa_msa_write_phylip(str(path).encode(), hs, n, wp, dt))


class Imap:
    """the individual -> species list of an Imap file (parse_mapfile, parsemap.c:227)"""

    def __init__(self, path):
        self.h = _lib().bpa_imap_read(str(path).encode())
        if not self.h:
            raise api.BpaError(api._err())

    def close(self):
        if self.h:
            _lib().bpa_imap_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def entries(self):
        L = _lib()
        return [(L.bpa_imap_individual(self.h, k).decode(), L.bpa_imap_species(self.h, k).decode())
                for k in range(L.bpa_imap_count(self.h))]

    def species_of(self, label, species):
        arr = (C.c_char_p * len(species))(*[s.encode() for s in species])
        k = _lib().bpa_imap_lookup(self.h, label.encode(), arr, len(species))
        if k < 0:
            raise api.BpaError(api._err())
        return k


def load_dataset(seqfile, imapfile=None, species=None, phase=None, model="jc69", cleandata=False,
                 nloci=0):
    """The input pipeline of method.c:3299-3672 for DNA data.  Returns one dict per locus:
    labels, species (index per sequence, when an Imap is given), seqs + weights (the patterns the
    likelihood runs on), ambiguous_sites, original_length, and for phased loci `diploid` =
    dict(resolution_count, mapping, unphased_weights) ready for Locus.set_diploid."""
    jc69 = model == "jc69"
    msas = read_phylip(seqfile, nloci)
    imap = Imap(imapfile) if imapfile else None
    out = []
    for k, m in enumerate(msas):
        deleted = m.remove_missing_sequences()
        if deleted < 0:
            raise api.BpaError(f"Locus {k} contains missing sequences only")
        if cleandata:
            if not m.remove_ambiguous():
                raise api.BpaError(f"All sites in locus {k} contain ambiguous characters")
            amb = 0
        else:
            amb = m.count_ambiguous_sites()
        original_length = m.length
        w = m.compress(jc69)
        rec = dict(index=k, removed_sequences=deleted, ambiguous_sites=amb, original_length=original_length,
                   unphased_patterns=m.length)
        sp = None
        if imap is not None and species is not None:
            sp = [imap.species_of(lab, species) for lab in m.labels]
        if phase is not None and any(phase):
            if sp is None:
                raise api.BpaError("phase needs the Imap and the species list")
            dip = [1 if phase[s] else 0 for s in sp]
            rc = m.diploid_resolve(dip, w)
            w3, mapping = m.compress_diploid(jc69)
            rec["diploid"] = dict(resolution_count=rc, mapping=mapping, unphased_weights=w)
            sp = [s for s, d in zip(sp, dip) for _ in range(2 if d else 1)]
            w = w3
        rec.update(labels=m.labels, species=sp, seqs=m.sequences, weights=w, msa=m)
        out.append(rec)
    return out


def make_locus(engine, rec, model="jc69", rate_cats=1):
    """locus_create + tip states + weights (+ diploid tables) for one record of load_dataset
    (method.c:4137-4196); JC69 or GTR, DNA."""
    tips = len(rec["seqs"])
    inner = tips - 1
    code = api.MODEL_JC69 if model == "jc69" else api.MODEL_GTR
    loc = api.Locus(engine, api.DATA_DNA, code, tips, 2 * inner, 4, len(rec["seqs"][0]), 1,
                    2 * (2 * tips - 2), rate_cats, 0)
    for i, s in enumerate(rec["seqs"]):
        loc.set_tip_states(i, s)
    if "diploid" in rec:
        d = rec["diploid"]
        loc.set_diploid(d["resolution_count"], d["mapping"], d["unphased_weights"])
    else:
        loc.set_pattern_weights(rec["weights"])
    return loc
