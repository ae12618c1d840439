"""ctypes access to the CHECKERS (test infrastructure, never product):

* ``oracle/liboracle.so``      – this repo's plain-C restatement (oracle/oracle.c)
* ``oracle/_ref/libbppref.so`` – the real reference, compiled in place from
  /root/reference/src by oracle/Makefile, driven through oracle/ref_shim.c

Only tests/, ``__graft_entry__.smoke()`` and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libbppref.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "bpp")

ORDER_SEQ, ORDER_PAIR, ORDER_FMA4 = 0, 1, 2
ARCH_CPU, ARCH_SSE, ARCH_AVX, ARCH_AVX2 = 0, 1, 2, 4
DATA_DNA, DATA_AA = 0, 1
MODEL_JC69, MODEL_GTR, MODEL_LG = 0, 7, 10
DNA_MODELS = {"jc69": 0, "k80": 1, "f81": 2, "hky": 3, "t92": 4, "tn93": 5, "f84": 6, "gtr": 7}

dp = C.POINTER(C.c_double)
up = C.POINTER(C.c_uint)
ulp = C.POINTER(C.c_ulong)
ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(dp) if a is not None else None


def _u(a):
    return a.ctypes.data_as(up) if a is not None else None


def build_oracle():
    """(Re)build liboracle.so (and oracle/_ref when the reference sources exist)."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-j8"], check=True,
                   stdout=subprocess.DEVNULL)


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.orc_root_loglikelihood.restype = C.c_double
        L.orc_diploid_loglikelihood.restype = C.c_double
        L.orc_get_map_nt.restype = up
        L.orc_get_map_aa.restype = up
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


_ref = None


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_locus_new.restype = C.c_void_p
        L.ref_root_loglikelihood.restype = C.c_double
        L.ref_full_loglikelihood.restype = C.c_double
        L.ref_run_tape.restype = C.c_double
        L.ref_aa_rates_lg.restype = dp
        L.ref_aa_freqs_lg.restype = dp
        L.ref_map_nt.restype = up
        L.ref_map_aa.restype = up
        L.pll_core_root_loglikelihood.restype = C.c_double
        _ref = L
    return _ref


# --------------------------------------------------------------------------
# oracle wrappers (numpy in / numpy out).  CLV layout = the reference's
# [pattern][rate][state]; P-matrix layout [rate][row][col].
# --------------------------------------------------------------------------
def orc_map(dna=True):
    L = oracle()
    p = L.orc_get_map_nt() if dna else L.orc_get_map_aa()
    return np.ctypeslib.as_array(p, shape=(256,)).copy()


def orc_tipclv(states, rate_cats, seq, dna=True):
    L = oracle()
    seq = seq.encode() if isinstance(seq, str) else seq
    sites = len(seq)
    clv = np.zeros((sites, rate_cats, states))
    m = np.ascontiguousarray(orc_map(dna), dtype=np.uint32)
    L.orc_set_tipclv(states, sites, rate_cats, _u(m), seq, _d(clv))
    return clv


def orc_partial(left, right, lmat, rmat, lscaler=None, rscaler=None, scaling=False,
                order=ORDER_PAIR):
    L = oracle()
    sites, R, S = left.shape
    parent = np.zeros_like(left)
    ps = np.zeros(sites, dtype=np.uint32) if scaling else None
    L.orc_update_partial_ii(S, sites, R, _d(parent), _u(ps), _d(left), _d(right),
                            _d(np.ascontiguousarray(lmat)), _d(np.ascontiguousarray(rmat)),
                            _u(lscaler), _u(rscaler), order)
    return parent, ps


def orc_lnl(clv, freqs, rw, weights, scaler=None, order=ORDER_PAIR, persite=False):
    L = oracle()
    sites, R, S = clv.shape
    ps = np.zeros(sites) if persite else None
    v = L.orc_root_loglikelihood(S, sites, R, _d(clv), _u(scaler), _d(freqs), _d(rw),
                                 _u(np.ascontiguousarray(weights, dtype=np.uint32)), _d(ps), order)
    return (v, ps) if persite else v


def orc_lhvec(clv, freqs, rw, order=ORDER_PAIR):
    L = oracle()
    sites, R, S = clv.shape
    out = np.zeros(sites)
    L.orc_root_likelihood_vector(S, sites, R, _d(clv), _d(freqs), _d(rw), _d(out), order)
    return out


def orc_diploid_lnl(lh, res_count, mapping, uweights):
    L = oracle()
    rc = np.ascontiguousarray(res_count, dtype=np.uint64)
    mp = np.ascontiguousarray(mapping, dtype=np.uint64)
    uw = np.ascontiguousarray(uweights, dtype=np.uint32)
    return L.orc_diploid_loglikelihood(_d(lh), len(rc), rc.ctypes.data_as(ulp),
                                       mp.ctypes.data_as(ulp), _u(uw))


def orc_pmatrix_jc69(rates, t):
    L = oracle()
    rates = np.ascontiguousarray(rates, dtype=np.float64)
    out = np.zeros((len(rates), 4, 4))
    L.orc_pmatrix_jc69(len(rates), _d(rates), C.c_double(t), _d(out))
    return out


def orc_pmatrix_dna(model, freqs, qrates, rates, t):
    """closed-form K80/F81/HKY/T92/TN93/F84 P(t) (locus.c:1981-2323)"""
    L = oracle()
    rates = np.ascontiguousarray(rates, dtype=np.float64)
    out = np.zeros((len(rates), 4, 4))
    L.orc_pmatrix_dna(DNA_MODELS[model], _d(np.ascontiguousarray(freqs, dtype=np.float64)),
                      _d(np.ascontiguousarray(qrates, dtype=np.float64)), len(rates), _d(rates),
                      C.c_double(t), _d(out))
    return out


def orc_eigen(freqs, qrates):
    L = oracle()
    S = len(freqs)
    ev, iev, evals = np.zeros((S, S)), np.zeros((S, S)), np.zeros(S)
    L.orc_update_eigen(S, _d(np.ascontiguousarray(freqs, dtype=np.float64)),
                       _d(np.ascontiguousarray(qrates, dtype=np.float64)), _d(ev), _d(iev), _d(evals))
    return ev, iev, evals


def orc_pmatrix_eigen(rates, t, evals, ev, iev, library_form=False):
    L = oracle()
    S = len(evals)
    rates = np.ascontiguousarray(rates, dtype=np.float64)
    out = np.zeros((len(rates), S, S))
    L.orc_pmatrix_eigen(S, len(rates), _d(rates), C.c_double(t), _d(evals), _d(ev), _d(iev),
                        _d(out), int(library_form))
    return out


def orc_gamma_cats(alpha, cats, beta=None):
    L = oracle()
    out = np.zeros(cats)
    L.orc_gamma_cats(C.c_double(alpha), C.c_double(alpha if beta is None else beta), cats, _d(out))
    return out


def _compress(fn, seqs, dna, jc69, extra=()):
    count, length = len(seqs), len(seqs[0])
    bufs = [C.create_string_buffer(s.encode() if isinstance(s, str) else s, length + 1) for s in seqs]
    arr = (C.c_char_p * count)(*[C.cast(b, C.c_char_p) for b in bufs])
    ln = C.c_int(length)
    w = np.zeros(length, dtype=np.uint32)
    ok = fn(arr, count, C.byref(ln), *extra, int(jc69), _u(w))
    assert ok
    n = ln.value
    return [b.raw[:n].decode() for b in bufs], w[:n].copy()


def orc_compress(seqs, dna=True, jc69=False):
    L = oracle()
    m = np.ascontiguousarray(orc_map(dna), dtype=np.uint32)
    return _compress(L.orc_compress, seqs, dna, jc69, extra=(_u(m),))


def ref_compress(seqs, dna=True, jc69=False):
    L = ref()
    return _compress(L.ref_compress, seqs, dna, jc69, extra=(int(dna),))


# --------------------------------------------------------------------------
# A whole-locus evaluation with the oracle: the start-up sequence of
# method.c:4285-4297 (all matrices, all partials, root lnL) on plain arrays.
# --------------------------------------------------------------------------
def postorder(left, right, root):
    out, stack = [], [(root, 0)]
    while stack:
        n, st = stack.pop()
        if left[n] < 0:
            continue
        if st == 0:
            stack.append((n, 1))
            stack.append((right[n], 0))
            stack.append((left[n], 0))
        else:
            out.append(n)
    return out


class OracleLocus:
    """Oracle-side twin of a reference locus: tips first, inner nodes after."""

    def __init__(self, states, rate_cats, seqs, weights, model="jc69", freqs=None, qrates=None,
                 rates=None, scaling=False, order=None):
        self.S, self.R = states, rate_cats
        self.dna = states == 4
        self.tips = len(seqs)
        self.sites = len(seqs[0])
        self.weights = np.ascontiguousarray(weights, dtype=np.uint32)
        self.model = model
        self.freqs = np.full(states, 1.0 / states) if freqs is None else np.asarray(freqs, float)
        self.qrates = qrates
        self.rates = np.ones(rate_cats) if rates is None else np.asarray(rates, float)
        self.rw = np.full(rate_cats, 1.0 / rate_cats)
        self.scaling = scaling
        self.order = order if order is not None else (ORDER_PAIR if states == 4 else ORDER_FMA4)
        n = 2 * self.tips - 1
        self.clv = [None] * n
        self.scaler = [None] * n
        self.pmat = [None] * n
        for i, s in enumerate(seqs):
            self.clv[i] = orc_tipclv(states, rate_cats, s, self.dna)
        if model not in DNA_MODELS or model == "gtr":
            self.eig = orc_eigen(self.freqs, self.qrates)

    def pmatrix(self, t):
        if self.model == "jc69":
            return orc_pmatrix_jc69(self.rates, t)
        if self.model in DNA_MODELS and self.model != "gtr":
            return orc_pmatrix_dna(self.model, self.freqs, self.qrates, self.rates, t)
        ev, iev, evals = self.eig
        return orc_pmatrix_eigen(self.rates, t, evals, ev, iev)

    def full_lnl(self, left, right, times, root, rate_mui=1.0):
        parent = {}
        for i in range(len(left)):
            if left[i] >= 0:
                parent[left[i]] = i
                parent[right[i]] = i
        for i, p in parent.items():
            self.pmat[i] = self.pmatrix((times[p] - times[i]) * rate_mui)
        for nd in postorder(left, right, root):
            l, r = left[nd], right[nd]
            self.clv[nd], self.scaler[nd] = orc_partial(
                self.clv[l], self.clv[r], self.pmat[l], self.pmat[r],
                self.scaler[l], self.scaler[r], self.scaling, self.order)
        return orc_lnl(self.clv[root], self.freqs, self.rw, self.weights,
                       self.scaler[root], self.order)


# --------------------------------------------------------------------------
# reference wrappers (through oracle/ref_shim.c)
# --------------------------------------------------------------------------
class RefLocus:
    def __init__(self, states, rate_cats, seqs, weights, model="jc69", freqs=None, qrates=None,
                 alpha=None, rates=None, scaling=False, arch=ARCH_AVX2):
        L = ref()
        self.L = L
        dtype = DATA_DNA if states == 4 else DATA_AA
        mdl = DNA_MODELS[model] if model in DNA_MODELS else MODEL_LG
        self.S, self.R = states, rate_cats
        self.tips, self.sites = len(seqs), len(seqs[0])
        self.h = C.c_void_p(L.ref_locus_new(dtype, mdl, self.tips, states, self.sites, rate_cats,
                                            int(scaling), arch))
        for i, s in enumerate(seqs):
            L.ref_set_tip(self.h, i, s.encode() if isinstance(s, str) else s)
        L.ref_set_weights(self.h, _u(np.ascontiguousarray(weights, dtype=np.uint32)))
        if freqs is None and model == "jc69":
            freqs = [0.25] * 4      # locus_set_frequencies_and_rates, locus.c:901
        if freqs is not None:
            L.ref_set_freqs(self.h, _d(np.ascontiguousarray(freqs, dtype=np.float64)))
        if qrates is not None:
            L.ref_set_qrates(self.h, _d(np.ascontiguousarray(qrates, dtype=np.float64)))
        if alpha is not None:
            L.ref_set_alpha(self.h, C.c_double(alpha))
        if rates is not None:
            L.ref_set_rates(self.h, _d(np.ascontiguousarray(rates, dtype=np.float64)))

    def rates(self):
        out = np.zeros(self.R)
        self.L.ref_get_rates(self.h, _d(out))
        return out

    def set_tree(self, left, right, times, root):
        l = np.ascontiguousarray(left, dtype=np.int32)
        r = np.ascontiguousarray(right, dtype=np.int32)
        t = np.ascontiguousarray(times, dtype=np.float64)
        self.L.ref_set_tree(self.h, l.ctypes.data_as(ip), r.ctypes.data_as(ip), _d(t), int(root))

    def full_lnl(self):
        return self.L.ref_full_loglikelihood(self.h)

    def clv(self, idx):
        out = np.zeros((self.sites, self.R, self.S))
        self.L.ref_get_clv(self.h, idx, _d(out))
        return out

    def pmatrix(self, idx):
        out = np.zeros((self.R, self.S, self.S))
        self.L.ref_get_pmatrix(self.h, idx, _d(out))
        return out

    def scaler(self, idx):
        out = np.zeros(self.sites, dtype=np.uint32)
        self.L.ref_get_scaler(self.h, idx, _u(out))
        return out

    def eigen(self):
        ev, iev, evals = np.zeros((self.S, self.S)), np.zeros((self.S, self.S)), np.zeros(self.S)
        self.L.ref_get_eigen(self.h, _d(ev), _d(iev), _d(evals))
        return ev, iev, evals

    def free(self):
        self.L.ref_locus_free(self.h)


def ref_partial(left, right, lmat, rmat, lscaler=None, rscaler=None, scaling=False, arch=ARCH_AVX2):
    """pll_core_update_partial_ii (core_partials.c:585) straight from the reference."""
    L = ref()
    sites, R, S = left.shape

    def al(a):  # 32-byte aligned copies (AVX loads are aligned, util.c:143)
        buf = np.zeros(a.size + 4, dtype=np.float64)
        off = (-buf.ctypes.data % 32) // 8
        v = buf[off:off + a.size].reshape(a.shape)
        v[...] = a
        return v

    l, r, lm, rm = al(left), al(right), al(np.asarray(lmat)), al(np.asarray(rmat))
    parent = al(np.zeros_like(left))
    ps = np.zeros(sites, dtype=np.uint32) if scaling else None
    L.pll_core_update_partial_ii(S, sites, R, _d(parent), _u(ps), _d(l), _d(r), _d(lm), _d(rm),
                                 _u(lscaler), _u(rscaler), arch)
    return parent.copy(), ps


def lg_model():
    L = ref()
    rates = np.ctypeslib.as_array(L.ref_aa_rates_lg(), shape=(190,)).copy()
    freqs = np.ctypeslib.as_array(L.ref_aa_freqs_lg(), shape=(20,)).copy()
    return rates, freqs
