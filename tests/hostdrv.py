"""ctypes harness for the C host driver (include/bpp_amd_host.h, bpp_amd/libbpp_amd_host.so)."""
import ctypes as C
import os
import numpy as np

import bpp_amd
from bpp_amd import build as _build

HOST_SO = _build.HOST_OUT


class A00Tree(C.Structure):
    _fields_ = [("tips", C.c_int), ("n", C.c_int), ("root", C.c_int),
                ("left", C.POINTER(C.c_int)), ("right", C.POINTER(C.c_int)), ("parent", C.POINTER(C.c_int)),
                ("time", C.POINTER(C.c_double)),
                ("clv", C.POINTER(C.c_int)), ("pmat", C.POINTER(C.c_int)), ("scaler", C.POINTER(C.c_int)),
                ("rate_mui", C.c_double), ("lnl", C.c_double),
                ("pop", C.POINTER(C.c_int)), ("logpr", C.c_double)]


class HipCtx(C.Structure):
    _fields_ = [("engine", C.c_void_p), ("loci", C.POINTER(C.c_void_p))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        bpp_amd.lib()                      # libbpp_amd.so first (rpath $ORIGIN also finds it)
        L = C.CDLL(HOST_SO)
        L.a00_create.restype = C.c_void_p
        L.a00_create.argtypes = [C.c_uint, C.c_void_p, C.c_void_p, C.c_ulong]
        L.a00_destroy.argtypes = [C.c_void_p]
        L.a00_set_tree.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_double), C.c_int, C.c_int]
        L.a00_tree.restype = C.POINTER(A00Tree)
        L.a00_tree.argtypes = [C.c_void_p, C.c_uint]
        L.a00_set_species_tree.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double)]
        L.a00_set_tip_species.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_int)]
        L.a00_set_finetune.argtypes = [C.c_void_p] + [C.c_double] * 4
        L.a00_set_tau_prior.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.a00_set_theta_prior.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.a00_get_thetas.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.a00_get_thetas.restype = C.c_uint
        L.a00_locus_logpr.restype = C.c_double
        L.a00_locus_logpr.argtypes = [C.c_void_p, C.c_uint]
        L.a00_get_taus.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.a00_get_taus.restype = C.c_uint
        L.a00_initialize.argtypes = [C.c_void_p]
        L.a00_iterate.argtypes = [C.c_void_p]
        L.a00_total_lnl.restype = C.c_double
        L.a00_total_lnl.argtypes = [C.c_void_p]
        L.a00_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.POINTER(C.c_ulong), C.POINTER(C.c_ulong)]
        _lib = L
    return _lib


class Driver:
    def __init__(self, data, eval_fn_addr, ctx_addr, seed=1, scaling=False):
        L = lib()
        self.n = len(data)
        self.h = C.c_void_p(L.a00_create(self.n, eval_fn_addr, ctx_addr, seed))
        for i, d in enumerate(data):
            l = np.ascontiguousarray(d["left"], dtype=np.int32)
            r = np.ascontiguousarray(d["right"], dtype=np.int32)
            t = np.ascontiguousarray(d["times"], dtype=np.float64)
            ok = L.a00_set_tree(self.h, i, len(d["seqs"]), l.ctypes.data_as(C.POINTER(C.c_int)),
                                r.ctypes.data_as(C.POINTER(C.c_int)), t.ctypes.data_as(C.POINTER(C.c_double)),
                                int(d["root"]), int(scaling))
            assert ok

    def set_species_tree(self, parent, tau, theta):
        n = len(parent)
        self.species = (n + 1) // 2
        ok = lib().a00_set_species_tree(self.h, self.species, (C.c_int * n)(*parent), (C.c_double * n)(*tau),
                                        (C.c_double * n)(*theta))
        assert ok, "bad species tree"

    def set_tip_species(self, i, species):
        assert lib().a00_set_tip_species(self.h, i, (C.c_int * len(species))(*species))

    def set_finetune(self, gage, gspr, tau, mix):
        lib().a00_set_finetune(self.h, gage, gspr, tau, mix)

    def set_subst_model(self, i, freqs, qrates, alpha, ncat):
        L = lib()
        L.a00_set_subst_model.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_int]
        assert L.a00_set_subst_model(self.h, i, (C.c_double * 4)(*freqs), (C.c_double * 6)(*qrates), float(alpha), int(ncat))

    def get_subst_model(self, i):
        L = lib()
        L.a00_get_subst_model.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        f, q, a = (C.c_double * 4)(), (C.c_double * 6)(), C.c_double()
        assert L.a00_get_subst_model(self.h, i, f, q, C.byref(a))
        return list(f), list(q), a.value

    def set_subst_moves(self, ft_freqs, ft_qrates, ft_alpha, alpha_a=1.0, alpha_b=1.0):
        L = lib()
        L.a00_set_subst_moves.argtypes = [C.c_void_p] + [C.c_double] * 5
        L.a00_set_subst_moves(self.h, ft_freqs, ft_qrates, ft_alpha, alpha_a, alpha_b)

    def set_param_backend(self, fn_addr):
        L = lib()
        L.a00_set_param_backend.argtypes = [C.c_void_p, C.c_void_p]
        L.a00_set_param_backend(self.h, fn_addr)

    def set_threads(self, n):
        """worker threads of the per-locus loops (the trajectory does not depend on the count)"""
        lib().a00_set_threads.argtypes = [C.c_void_p, C.c_int]
        lib().a00_set_threads(self.h, int(n))

    def set_program_moves(self, on, slide_prob=0.1):
        """THETA / TAU / MIX as the program runs them (BPP kernel): Gibbs draws of the thetas, thetas re-drawn inside TAU and MIX"""
        lib().a00_set_program_moves.argtypes = [C.c_void_p, C.c_int, C.c_double]
        lib().a00_set_program_moves(self.h, int(bool(on)), float(slide_prob))

    def gibbs_counters(self):
        a, b = C.c_ulong(), C.c_ulong()
        lib().a00_gibbs_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.POINTER(C.c_ulong)]
        lib().a00_gibbs_counters(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def set_proposal_kernel(self, kind):
        """0 uniform windows on the a00 streams (default), 1 BPP's legacy_rndu + Bactrian-Laplace (A00_KERNEL_BPP)"""
        lib().a00_set_proposal_kernel.argtypes = [C.c_void_p, C.c_int]
        lib().a00_set_proposal_kernel(self.h, int(kind))

    def set_tau_prior(self, alpha, beta):
        lib().a00_set_tau_prior(self.h, alpha, beta)

    def set_theta_prior(self, alpha, beta, finetune):
        lib().a00_set_theta_prior(self.h, alpha, beta, finetune)

    def thetas(self):
        a = (C.c_double * 15)()
        n = lib().a00_get_thetas(self.h, a)
        return [a[i] for i in range(n)]

    def taus(self):
        a = (C.c_double * 15)()
        n = lib().a00_get_taus(self.h, a)
        return [a[i] for i in range(n)]

    def logpr(self, i):
        return lib().a00_locus_logpr(self.h, i)

    def initialize(self):
        assert lib().a00_initialize(self.h), bpp_amd.lib().bpa_last_error()

    def iterate(self):
        assert lib().a00_iterate(self.h), bpp_amd.lib().bpa_last_error()

    def total_lnl(self):
        return lib().a00_total_lnl(self.h)

    def counters(self):
        p, a, s = C.c_ulong(), C.c_ulong(), C.c_ulong()
        lib().a00_counters(self.h, C.byref(p), C.byref(a), C.byref(s))
        return p.value, a.value, s.value

    def tree(self, i):
        t = lib().a00_tree(self.h, i).contents
        n = t.n
        g = lambda p: [p[k] for k in range(n)]
        return dict(tips=t.tips, root=t.root, left=g(t.left), right=g(t.right), parent=g(t.parent),
                    time=g(t.time), clv=g(t.clv), pmat=g(t.pmat), scaler=g(t.scaler), lnl=t.lnl,
                    pop=g(t.pop), logpr=t.logpr)

    def close(self):
        if self.h:
            lib().a00_destroy(self.h)
            self.h = None


def reference_driver(data, seed=1, scaling=False):
    """driver on the REAL reference's locus API (oracle/ref_shim.c: ref_backend_eval)"""
    import oraclelib as O
    import tape
    rls = [tape.ref_locus_for(d, scaling) for d in data]
    arr = (C.c_void_p * len(rls))(*[rl.h for rl in rls])
    fn = C.cast(O.ref().ref_backend_eval, C.c_void_p)
    drv = Driver(data, fn, C.cast(arr, C.c_void_p), seed, scaling)
    drv._keep = (rls, arr)
    drv.set_param_backend(C.cast(O.ref().ref_backend_params, C.c_void_p))
    return drv


def reference_driver_cohorts(data, split, seed=1, scaling=False):
    """the reference-API driver with its loci in two cohorts (a00_set_cohorts; the synchronous backend is its own submit)"""
    import oraclelib as O
    drv = reference_driver(data, seed, scaling)
    L = lib()
    L.a00_set_cohorts.argtypes = [C.c_void_p, C.c_uint] + [C.c_void_p] * 4
    arr = C.cast(drv._keep[1], C.c_void_p)
    assert L.a00_set_cohorts(drv.h, split, C.cast(O.ref().ref_backend_eval, C.c_void_p), C.cast(L.a00_backend_wait_none, C.c_void_p), arr, arr)
    return drv


def hip_driver(engine, loci, data, seed=1, scaling=False):
    """driver on libbpp_amd.so (a00_backend_hip)"""
    arr = (C.c_void_p * len(loci))(*[l.h for l in loci])
    ctx = HipCtx(engine.h, arr)
    fn = C.cast(lib().a00_backend_hip, C.c_void_p)
    drv = Driver(data, fn, C.cast(C.pointer(ctx), C.c_void_p), seed, scaling)
    drv._keep = (arr, ctx)
    drv.set_param_backend(C.cast(lib().a00_backend_hip_params, C.c_void_p))
    return drv


def hip_driver_cohorts(engines, loci, data, split, seed=1, scaling=False):
    """driver on TWO engines (a00_set_cohorts): loci[:split] were made on engines[0], loci[split:] on engines[1] — a
    per-locus step of one cohort is proposed while the other cohort's launch runs"""
    L = lib()
    arr = (C.c_void_p * len(loci))(*[l.h for l in loci])
    ctxs = [HipCtx(e.h, arr) for e in engines]
    drv = Driver(data, C.cast(L.a00_backend_hip, C.c_void_p), C.cast(C.pointer(ctxs[0]), C.c_void_p), seed, scaling)
    drv._keep = (arr, ctxs)
    drv.set_param_backend(C.cast(L.a00_backend_hip_params, C.c_void_p))
    L.a00_set_cohorts.argtypes = [C.c_void_p, C.c_uint] + [C.c_void_p] * 4
    if not L.a00_set_cohorts(drv.h, split, C.cast(L.a00_backend_hip_submit, C.c_void_p), C.cast(L.a00_backend_hip_wait, C.c_void_p),
                             C.cast(C.pointer(ctxs[0]), C.c_void_p), C.cast(C.pointer(ctxs[1]), C.c_void_p)):
        raise RuntimeError("a00_set_cohorts failed")
    return drv


def prior_driver(data, seed=1):
    """driver with lnL = 0 (a00_backend_prior): samples gene trees from the MSC prior"""
    fn = C.cast(lib().a00_backend_prior, C.c_void_p)
    return Driver(data, fn, None, seed, False)
