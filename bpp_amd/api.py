"""ctypes binding of libbpp_amd.so (include/bpp_amd.h) plus a thin host-side
mirror of the reference's locus API so that tests read like the reference's own
call sites (method.c:4137-4300, gtree.c:5447-5467).

The library is the product; this module is plumbing.  It fails loudly when the
HIP extension is missing or no GPU is visible — there is no CPU fallback.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbpp_amd.so")

DATA_DNA, DATA_AA = 0, 1
MODEL_JC69, MODEL_GTR, MODEL_LG = 0, 7, 10
ATTRIB_ARCH_HIP = 1 << 6
SCALE_BUFFER_NONE = -1
PARAM_FREQS, PARAM_SUBST, PARAM_RATES = 1, 2, 4


class BpaError(RuntimeError):
    pass


class Op(C.Structure):
    """bpa_op_t: one pll_core_update_partial_ii call (core_partials.c:585)."""
    _fields_ = [("parent_clv", C.c_uint32), ("parent_scaler", C.c_int32),
                ("left_clv", C.c_uint32), ("left_pmatrix", C.c_uint32), ("left_scaler", C.c_int32),
                ("right_clv", C.c_uint32), ("right_pmatrix", C.c_uint32), ("right_scaler", C.c_int32)]


OP_DTYPE = np.dtype([("parent_clv", "<u4"), ("parent_scaler", "<i4"), ("left_clv", "<u4"),
                     ("left_pmatrix", "<u4"), ("left_scaler", "<i4"), ("right_clv", "<u4"),
                     ("right_pmatrix", "<u4"), ("right_scaler", "<i4")])


class GTreeView(C.Structure):
    """bpa_gtree_view_t: the gnode_t fields the reference's all-nodes recursions read, one entry per node"""
    _fields_ = [("nodes", C.c_uint), ("root", C.c_int), ("left", C.POINTER(C.c_int)), ("right", C.POINTER(C.c_int)),
                ("parent", C.POINTER(C.c_int)), ("time", C.POINTER(C.c_double)),
                ("clv_index", C.POINTER(C.c_uint)), ("scaler_index", C.POINTER(C.c_int)),
                ("pmatrix_index", C.POINTER(C.c_uint)), ("rate_mui", C.c_double)]


class Batch(C.Structure):
    _fields_ = [("nloci", C.c_uint), ("loci", C.POINTER(C.c_void_p)),
                ("mat_off", C.POINTER(C.c_uint)), ("mat_pmatrix", C.POINTER(C.c_uint)),
                ("mat_length", C.POINTER(C.c_double)),
                ("op_off", C.POINTER(C.c_uint)), ("ops", C.POINTER(Op)),
                ("root_clv", C.POINTER(C.c_uint)), ("root_scaler", C.POINTER(C.c_int))]


_lib = None


def lib():
    """Load libbpp_amd.so; raise if it has not been built (python -m bpp_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BpaError(f"{LIB_PATH} is missing: run `python -m bpp_amd.build` (hipcc, gfx950). "
                       "bpp_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u, i, d = C.c_void_p, C.c_uint, C.c_int, C.c_double
    dp, up = C.POINTER(C.c_double), C.POINTER(C.c_uint)
    sig = {
        "bpa_version": (C.c_char_p, []),
        "bpa_last_error": (C.c_char_p, []),
        "bpa_experimental_build": (i, []),
        "bpa_device_count": (i, []),
        "bpa_engine_create": (vp, [i, vp]),
        "bpa_engine_destroy": (None, [vp]),
        "bpa_engine_synchronize": (i, [vp]),
        "bpa_engine_set_options": (None, [vp, i, d]),
        "bpa_locus_create": (vp, [vp] + [u] * 11),
        "bpa_locus_destroy": (None, [vp]),
        "bpa_set_tip_states": (i, [vp, u, up, C.c_char_p]),
        "bpa_set_pattern_weights": (None, [vp, up]),
        "bpa_set_frequencies": (None, [vp, u, dp]),
        "bpa_set_subst_params": (None, [vp, u, dp]),
        "bpa_set_category_rates": (None, [vp, dp]),
        "bpa_set_category_weights": (None, [vp, dp]),
        "bpa_set_param_indices": (None, [vp, up]),
        "bpa_set_diploid": (i, [vp, i, C.POINTER(C.c_ulong), C.POINTER(C.c_ulong), C.c_ulong, up]),
        "bpa_map_nt": (up, []),
        "bpa_map_aa": (up, []),
        "bpa_locus_update_matrices": (i, [vp, up, dp, u]),
        "bpa_locus_update_partials": (i, [vp, C.POINTER(Op), u]),
        "bpa_locus_root_loglikelihood": (d, [vp, u, i, up, dp]),
        "bpa_locus_update_all_matrices": (i, [vp, C.POINTER(GTreeView), dp]),
        "bpa_locus_update_all_partials": (i, [vp, C.POINTER(GTreeView)]),
        "bpa_core_update_pmatrix": (i, [vp, C.POINTER(dp), u, u, dp, dp, up, up, C.POINTER(dp),
                                        C.POINTER(dp), C.POINTER(dp), u, u]),
        "bpa_update_eigen": (i, [vp, dp, dp, dp, dp, dp, u]),
        "bpa_compute_gamma_cats": (i, [d, d, u, dp]),
        "bpa_compress_site_patterns": (i, [C.POINTER(C.c_char_p), up, i, C.POINTER(i), i, up]),
        "bpa_locus_get_clv": (i, [vp, u, dp]),
        "bpa_locus_set_clv": (i, [vp, u, dp]),
        "bpa_locus_get_pmatrix": (i, [vp, u, dp]),
        "bpa_locus_set_pmatrix": (i, [vp, u, dp]),
        "bpa_locus_get_scaler": (i, [vp, u, up]),
        "bpa_locus_set_scaler": (i, [vp, u, up]),
        "bpa_locus_get_eigen": (i, [vp, u, dp, dp, dp]),
        "bpa_plan_create": (vp, [vp, C.POINTER(Batch)]),
        "bpa_plan_destroy": (None, [vp]),
        "bpa_plan_set_lengths": (i, [vp, dp]),
        "bpa_plan_launch": (i, [vp]),
        "bpa_plans_launch": (i, [C.POINTER(vp), u]),
        "bpa_plan_get_lnl": (i, [vp, dp]),
        "bpa_plan_lnl_device": (vp, [vp]),
        "bpa_plan_enable_sum": (i, [vp, vp]),
        "bpa_plan_get_sum": (i, [vp, dp]),
        "bpa_plan_enable_partial_sums": (i, [vp, vp, C.POINTER(C.c_uint)]),
        "bpa_p2p_create": (vp, [vp, i, i, u, vp]),
        "bpa_p2p_connect": (i, [vp, vp]),
        "bpa_p2p_allreduce": (i, [vp, vp, u]),
        "bpa_p2p_status": (i, [vp]),
        "bpa_p2p_set_timeout": (None, [vp, u]),
        "bpa_plans_launch_exchange": (i, [vp, u, vp, vp, u]),
        "bpa_p2p_destroy": (None, [vp]),
        "bpa_batch_evaluate": (i, [vp, C.POINTER(Batch), dp]),
        "bpa_batch_begin": (i, [vp, C.POINTER(Batch)]),
        "bpa_batch_fill": (i, [vp, C.POINTER(Batch), u, u]),
        "bpa_batch_end": (i, [vp, C.POINTER(Batch), dp]),
        "bpa_batch_end_async": (i, [vp, C.POINTER(Batch)]),
        "bpa_batch_wait": (i, [vp, dp]),
        "bpa_engine_stage": (vp, [vp, vp, C.c_size_t]),
        "bpa_plan_set_params": (i, [vp, i, dp]),
        "bpa_plan_set_params_device": (i, [vp, i, vp]),
        "bpa_plan_work": (i, [vp, dp, dp, dp, C.POINTER(C.c_ulong), C.POINTER(C.c_ulong)]),
        "bpa_sampler_create": (vp, [vp, C.POINTER(vp), u, C.c_ulong]),
        "bpa_sampler_destroy": (None, [vp]),
        "bpa_sampler_set_tree": (i, [vp, u, C.POINTER(i), C.POINTER(i), dp, i]),
        "bpa_sampler_set_species_tree": (i, [vp, i, C.POINTER(i), dp, dp]),
        "bpa_sampler_set_tip_species": (i, [vp, u, C.POINTER(i)]),
        "bpa_sampler_set_finetune": (None, [vp, d, d, d, d]),
        "bpa_finetune_onestep": (d, [d, d]),
        "bpa_sampler_adapt_finetune": (i, [vp, dp, dp]),
        "bpa_sampler_burnin": (i, [vp, u, dp]),
        "bpa_burnin_schedule": (u, [u, C.POINTER(u), u]),
        "bpa_sampler_set_tau_prior": (None, [vp, d, d]),
        "bpa_sampler_set_theta_prior": (None, [vp, d, d, d]),
        "bpa_sampler_get_thetas": (i, [vp, dp]),
        "bpa_sampler_set_allreduce": (i, [vp, vp, vp, vp, u]),
        "bpa_sampler_get_taus": (i, [vp, dp]),
        "bpa_sampler_get_tree_msc": (i, [vp, u, C.POINTER(i), dp]),
        "bpa_sampler_initialize": (i, [vp]),
        "bpa_sampler_iterate": (i, [vp, u]),
        "bpa_sampler_get_tree": (i, [vp, u, C.POINTER(i), C.POINTER(i), C.POINTER(i), dp, C.POINTER(i),
                                     C.POINTER(i), C.POINTER(i), dp]),
        "bpa_sampler_summary": (i, [vp, dp, C.POINTER(C.c_ulong), C.POINTER(C.c_ulong), C.POINTER(C.c_ulong)]),
        "bpa_sampler_enable_timing": (i, [vp, u]),
        "bpa_sampler_set_subst_model": (i, [vp, u, dp, dp, d]),
        "bpa_sampler_get_subst_model": (i, [vp, u, dp, dp, dp]),
        "bpa_sampler_set_subst_moves": (None, [vp, d, d, d, d, d]),
        "bpa_sampler_timing": (i, [vp, dp, C.POINTER(C.c_ulong), dp, C.POINTER(C.c_ulong)]),
        "bpa_sampler_work": (i, [vp, dp, C.POINTER(C.c_ulong), C.POINTER(C.c_ulong), C.POINTER(C.c_ulong)]),
        "bpa_sampler_kind": (i, [vp]),
        "bpa_sampler_streams": (i, [vp]),
        "bpa_sampler_set_p2p": (i, [vp, vp, u]),
        "bpa_sampler_set_proposal_kernel": (i, [vp, i]),
        "bpa_sampler_set_program_moves": (i, [vp, i, d]),
        "bpa_sampler_gibbs_counters": (i, [vp, C.POINTER(C.c_ulong), C.POINTER(C.c_ulong)]),
        "bpa_engine_enable_timing": (None, [vp, i]),
        "bpa_engine_set_timing_stride": (None, [vp, u]),
        "bpa_engine_timing": (i, [vp, dp, dp, dp, C.POINTER(C.c_ulong)]),
        "bpa_engine_timing_work": (i, [vp, C.POINTER(C.c_ulong), dp]),
        "bpa_engine_timing_work_codes": (i, [vp, dp]),
        "bpa_plan_work_codes": (i, [vp, dp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


EXPORTED = ["bpa_version", "bpa_last_error", "bpa_experimental_build", "bpa_device_count", "bpa_engine_create",
            "bpa_engine_destroy", "bpa_engine_synchronize", "bpa_engine_set_options",
            "bpa_locus_create", "bpa_locus_destroy", "bpa_set_tip_states",
            "bpa_set_pattern_weights", "bpa_set_frequencies", "bpa_set_subst_params",
            "bpa_set_category_rates", "bpa_set_category_weights", "bpa_set_param_indices",
            "bpa_set_diploid", "bpa_map_nt", "bpa_map_aa", "bpa_locus_update_matrices",
            "bpa_locus_update_partials", "bpa_locus_root_loglikelihood",
            "bpa_locus_update_all_matrices", "bpa_locus_update_all_partials",
            "bpa_core_update_pmatrix", "bpa_update_eigen", "bpa_compute_gamma_cats",
            "bpa_compress_site_patterns", "bpa_locus_get_clv", "bpa_locus_set_clv",
            "bpa_locus_get_pmatrix", "bpa_locus_set_pmatrix", "bpa_locus_get_scaler", "bpa_locus_set_scaler",
            "bpa_locus_get_eigen", "bpa_plan_create", "bpa_plan_destroy", "bpa_plan_set_lengths",
            "bpa_plan_launch", "bpa_plan_get_lnl", "bpa_plan_lnl_device", "bpa_batch_evaluate", "bpa_batch_begin", "bpa_batch_fill", "bpa_batch_end", "bpa_batch_end_async", "bpa_batch_wait",
            "bpa_plan_enable_sum", "bpa_plan_enable_partial_sums", "bpa_plan_get_sum",
            "bpa_p2p_create", "bpa_p2p_connect", "bpa_p2p_allreduce", "bpa_p2p_status", "bpa_p2p_destroy", "bpa_p2p_set_timeout", "bpa_plans_launch_exchange", "bpa_plans_launch",
            "bpa_plan_set_params", "bpa_plan_set_params_device", "bpa_engine_stage",
            "bpa_plan_work", "bpa_engine_enable_timing", "bpa_engine_timing", "bpa_engine_set_timing_stride", "bpa_engine_timing_work", "bpa_engine_timing_work_codes", "bpa_plan_work_codes",
            "bpa_sampler_create", "bpa_sampler_destroy", "bpa_sampler_set_tree", "bpa_sampler_initialize",
            "bpa_sampler_set_species_tree", "bpa_sampler_set_tip_species", "bpa_sampler_set_finetune",
            "bpa_finetune_onestep", "bpa_sampler_adapt_finetune", "bpa_sampler_burnin", "bpa_burnin_schedule",
            "bpa_sampler_set_tau_prior", "bpa_sampler_get_taus", "bpa_sampler_get_tree_msc",
            "bpa_sampler_set_theta_prior", "bpa_sampler_get_thetas", "bpa_sampler_set_allreduce",
            "bpa_sampler_iterate", "bpa_sampler_get_tree", "bpa_sampler_summary",
            "bpa_sampler_enable_timing", "bpa_sampler_timing", "bpa_sampler_work", "bpa_sampler_kind", "bpa_sampler_streams", "bpa_sampler_set_p2p", "bpa_sampler_set_proposal_kernel",
            "bpa_sampler_set_program_moves", "bpa_sampler_gibbs_counters",
            "bpa_sampler_set_subst_model", "bpa_sampler_get_subst_model", "bpa_sampler_set_subst_moves"]


def _err():
    return lib().bpa_last_error().decode()


def _chk(ok):
    if not ok:
        raise BpaError(_err())


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def map_nt():
    return np.ctypeslib.as_array(lib().bpa_map_nt(), shape=(256,)).copy()


def map_aa():
    return np.ctypeslib.as_array(lib().bpa_map_aa(), shape=(256,)).copy()


def compute_gamma_cats(alpha, beta, categories):
    """pll_compute_gamma_cats, PLL_GAMMA_RATES_MEAN (gamma.c:221)."""
    out = np.zeros(categories)
    _chk(lib().bpa_compute_gamma_cats(alpha, beta, categories, _dp(out)))
    return out


def compress_site_patterns(seqs, dna=True, jc69=False):
    """compress_site_patterns (compress.c:218): returns (compressed seqs, weights)."""
    L = lib()
    count, length = len(seqs), len(seqs[0])
    bufs = [C.create_string_buffer(s.encode() if isinstance(s, str) else s, length + 1) for s in seqs]
    arr = (C.c_char_p * count)(*[C.cast(b, C.c_char_p) for b in bufs])
    ln = C.c_int(length)
    w = np.zeros(length, dtype=np.uint32)
    n = L.bpa_compress_site_patterns(arr, L.bpa_map_nt() if dna else L.bpa_map_aa(), count,
                                     C.byref(ln), int(jc69), _up(w))
    if not n:
        raise BpaError("compress_site_patterns failed")
    return [b.raw[:n].decode() for b in bufs], w[:n].copy()


class Engine:
    """One per process/GPU.  Raises when no GPU is visible."""

    def __init__(self, device=0, stream=None):
        L = lib()
        self.h = L.bpa_engine_create(device, stream)
        if not self.h:
            raise BpaError(_err())
        self.loci = []

    def set_options(self, usedata=1, bfbeta=1.0):
        lib().bpa_engine_set_options(self.h, int(usedata), float(bfbeta))

    def stage(self, array):
        """copy a host array into engine-owned device memory (a tape resident in HBM); returns the device address"""
        return _stage(self, array)

    def synchronize(self):
        _chk(lib().bpa_engine_synchronize(self.h))

    def enable_timing(self, on=True, stride=1):
        lib().bpa_engine_set_timing_stride(self.h, int(stride))
        lib().bpa_engine_enable_timing(self.h, int(on))

    def timing(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        n = C.c_ulong()
        _chk(lib().bpa_engine_timing(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        st, by = C.c_ulong(), C.c_double()
        _chk(lib().bpa_engine_timing_work(self.h, C.byref(st), C.byref(by)))
        bc = C.c_double()
        _chk(lib().bpa_engine_timing_work_codes(self.h, C.byref(bc)))
        return {"pmatrix_ms": a.value, "partials_ms": b.value, "reduce_ms": c.value,
                "launches": n.value, "steps": st.value, "bytes": by.value, "bytes_codes": bc.value}

    def update_eigen(self, freqs, subst, states):
        ev, iev, evals = np.zeros((states, states)), np.zeros((states, states)), np.zeros(states)
        _chk(lib().bpa_update_eigen(self.h, _dp(ev), _dp(iev), _dp(evals), _dp(_f64(freqs)),
                                    _dp(_f64(subst)), states))
        return ev, iev, evals

    def core_update_pmatrix(self, states, rates, branch_lengths, evals, evecs, ievecs,
                            param_indices=None):
        """pll_core_update_pmatrix (core_pmatrix.c:785) for one eigensystem."""
        rates, bl = _f64(rates), _f64(branch_lengths)
        R, n = len(rates), len(bl)
        out = np.zeros((n, R, states, states))
        dp = C.POINTER(C.c_double)
        pm = (dp * n)(*[_dp(out[i]) for i in range(n)])
        evals, evecs, ievecs = _f64(evals), _f64(evecs), _f64(ievecs)
        one = lambda a: (dp * 1)(_dp(a))
        pi = _u32(np.zeros(R) if param_indices is None else param_indices)
        mi = _u32(np.arange(n))
        _chk(lib().bpa_core_update_pmatrix(self.h, pm, states, R, _dp(rates), _dp(bl), _up(mi),
                                           _up(pi), one(evals), one(evecs), one(ievecs), n, 0))
        return out

    def close(self):
        if self.h:
            lib().bpa_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _stage(engine, array):
    a = np.ascontiguousarray(array)
    p = lib().bpa_engine_stage(engine.h, a.ctypes.data_as(C.c_void_p), a.nbytes)
    if not p:
        raise BpaError(_err())
    return p


class Locus:
    """Device twin of locus_t; constructor = locus_create (locus.c:622)."""

    def __init__(self, engine, dtype, model, tips, clv_buffers, states, sites, rate_matrices,
                 prob_matrices, rate_cats, scale_buffers, attributes=ATTRIB_ARCH_HIP):
        self.engine = engine
        self.dtype, self.model, self.tips, self.clv_buffers = dtype, model, tips, clv_buffers
        self.states, self.sites, self.rate_matrices = states, sites, rate_matrices
        self.prob_matrices, self.rate_cats, self.scale_buffers = prob_matrices, rate_cats, scale_buffers
        self.h = lib().bpa_locus_create(engine.h, dtype, model, tips, clv_buffers, states, sites,
                                        rate_matrices, prob_matrices, rate_cats, scale_buffers,
                                        attributes)
        if not self.h:
            raise BpaError(_err())
        engine.loci.append(self)

    # --- setters (pll_set_* of locus.c) ---
    def set_tip_states(self, tip_index, sequence, map_=None):
        L = lib()
        if map_ is None:
            mp = L.bpa_map_nt() if self.states == 4 else L.bpa_map_aa()
        else:
            self._map = _u32(map_)
            mp = _up(self._map)
        seq = sequence.encode() if isinstance(sequence, str) else sequence
        if len(seq) != self.sites:
            raise BpaError("sequence length != sites")
        _chk(L.bpa_set_tip_states(self.h, tip_index, mp, seq))

    def set_pattern_weights(self, w):
        w = _u32(w)
        assert len(w) == self.sites
        lib().bpa_set_pattern_weights(self.h, _up(w))

    def set_frequencies(self, index, f):
        lib().bpa_set_frequencies(self.h, index, _dp(_f64(f)))

    def set_subst_params(self, index, p):
        lib().bpa_set_subst_params(self.h, index, _dp(_f64(p)))

    def set_category_rates(self, rates):
        lib().bpa_set_category_rates(self.h, _dp(_f64(rates)))

    def set_category_weights(self, w):
        lib().bpa_set_category_weights(self.h, _dp(_f64(w)))

    def set_diploid(self, resolution_count, mapping, unphased_weights):
        rc = np.ascontiguousarray(resolution_count, dtype=np.uint64)
        mp = np.ascontiguousarray(mapping, dtype=np.uint64)
        uw = _u32(unphased_weights)
        ulp = C.POINTER(C.c_ulong)
        _chk(lib().bpa_set_diploid(self.h, len(rc), rc.ctypes.data_as(ulp), mp.ctypes.data_as(ulp),
                                   len(mp), _up(uw)))

    # --- update API with explicit indices ---
    def update_matrices(self, pmatrix_indices, branch_lengths):
        pi, bl = _u32(pmatrix_indices), _f64(branch_lengths)
        _chk(lib().bpa_locus_update_matrices(self.h, _up(pi), _dp(bl), len(pi)))

    def update_partials(self, ops):
        ops = np.ascontiguousarray(ops, dtype=OP_DTYPE)
        _chk(lib().bpa_locus_update_partials(self.h, ops.ctypes.data_as(C.POINTER(Op)), len(ops)))

    def root_loglikelihood(self, root_clv, root_scaler=SCALE_BUFFER_NONE, persite=False):
        ps = np.zeros(self.sites) if persite else None
        v = lib().bpa_locus_root_loglikelihood(self.h, root_clv, root_scaler, None,
                                               _dp(ps) if persite else None)
        if v != v:
            raise BpaError(_err())
        return (v, ps) if persite else v

    # --- buffer access in the reference's layouts ---
    def get_clv(self, idx):
        out = np.zeros((self.sites, self.rate_cats, self.states))
        _chk(lib().bpa_locus_get_clv(self.h, idx, _dp(out)))
        return out

    def set_clv(self, idx, clv):
        clv = _f64(clv)
        assert clv.shape == (self.sites, self.rate_cats, self.states)
        _chk(lib().bpa_locus_set_clv(self.h, idx, _dp(clv)))

    def get_pmatrix(self, idx):
        out = np.zeros((self.rate_cats, self.states, self.states))
        _chk(lib().bpa_locus_get_pmatrix(self.h, idx, _dp(out)))
        return out

    def set_pmatrix(self, idx, p):
        p = _f64(p)
        assert p.shape == (self.rate_cats, self.states, self.states)
        _chk(lib().bpa_locus_set_pmatrix(self.h, idx, _dp(p)))

    def get_scaler(self, idx):
        out = np.zeros(self.sites, dtype=np.uint32)
        _chk(lib().bpa_locus_get_scaler(self.h, idx, _up(out)))
        return out

    def set_scaler(self, idx, values):
        v = np.ascontiguousarray(values, dtype=np.uint32)
        assert v.size == self.sites
        _chk(lib().bpa_locus_set_scaler(self.h, idx, _up(v)))

    def get_eigen(self, index=0):
        S = self.states
        ev, iev, evals = np.zeros((S, S)), np.zeros((S, S)), np.zeros(S)
        _chk(lib().bpa_locus_get_eigen(self.h, index, _dp(ev), _dp(iev), _dp(evals)))
        return ev, iev, evals


# ---------------------------------------------------------------------------
# gene-tree side of the boundary: the gnode_t / gtree_t fields the path reads
# (bpp.h:692-774) and the reference-named update calls on them.
# ---------------------------------------------------------------------------
class GNode:
    __slots__ = ("left", "right", "parent", "time", "length", "clv_index", "scaler_index",
                 "pmatrix_index", "node_index")

    def __init__(self, node_index):
        self.left = self.right = self.parent = None
        self.time = 0.0
        self.length = 0.0
        self.node_index = node_index
        self.clv_index = node_index
        self.scaler_index = SCALE_BUFFER_NONE
        self.pmatrix_index = node_index


class GTree:
    """tips first, then inner nodes (gtree.c:2433-2439); rate_mui as gtree_t."""

    def __init__(self, left, right, times, root, scaling=False):
        n = len(left)
        self.tip_count = (n + 1) // 2
        self.inner_count = self.tip_count - 1
        self.edge_count = 2 * self.tip_count - 2
        self.nodes = [GNode(i) for i in range(n)]
        self.rate_mui = 1.0
        for i in range(n):
            nd = self.nodes[i]
            nd.time = float(times[i])
            if left[i] >= 0:
                nd.left, nd.right = self.nodes[left[i]], self.nodes[right[i]]
                nd.left.parent = nd.right.parent = nd
                if scaling:
                    nd.scaler_index = i - self.tip_count
        self.root = self.nodes[root]

    def postorder(self):
        out, stack = [], [(self.root, 0)]
        while stack:
            nd, st = stack.pop()
            if nd.left is None:
                continue
            if st == 0:
                stack += [(nd, 1), (nd.right, 0), (nd.left, 0)]
            else:
                out.append(nd)
        return out

    def branches(self):
        return [nd for nd in self.nodes if nd.parent is not None]


def node_op(node):
    l, r = node.left, node.right
    return (node.clv_index, node.scaler_index, l.clv_index, l.pmatrix_index, l.scaler_index,
            r.clv_index, r.pmatrix_index, r.scaler_index)


def branch_length(gtree, node):
    """strict clock, locus.c:2350; also stored in node.length as the reference does."""
    node.length = (node.parent.time - node.time) * gtree.rate_mui
    return node.length


def locus_update_matrices(locus, gtree, traversal, count=None):
    """locus_update_matrices (locus.c:2417): traversal = branches (child nodes)."""
    trav = traversal if count is None else traversal[:count]
    locus.update_matrices([nd.pmatrix_index for nd in trav], [branch_length(gtree, nd) for nd in trav])


def locus_update_partials(locus, traversal, count=None):
    """locus_update_partials (locus.c:2530): traversal children-first."""
    trav = traversal if count is None else traversal[:count]
    locus.update_partials(np.array([node_op(nd) for nd in trav], dtype=OP_DTYPE))


def _view(gtree):
    """bpa_gtree_view_t of a GTree (keeps the arrays alive on the returned object)"""
    nd = gtree.nodes
    idx = lambda x: -1 if x is None else x.node_index
    a = dict(left=np.array([idx(x.left) for x in nd], dtype=np.int32), right=np.array([idx(x.right) for x in nd], dtype=np.int32),
             parent=np.array([idx(x.parent) for x in nd], dtype=np.int32), time=np.array([x.time for x in nd], dtype=np.float64),
             clv=np.array([x.clv_index for x in nd], dtype=np.uint32), scaler=np.array([x.scaler_index for x in nd], dtype=np.int32),
             pmat=np.array([x.pmatrix_index for x in nd], dtype=np.uint32))
    ip = C.POINTER(C.c_int)
    v = GTreeView(len(nd), gtree.root.node_index, a["left"].ctypes.data_as(ip), a["right"].ctypes.data_as(ip),
                  a["parent"].ctypes.data_as(ip), _dp(a["time"]), _up(a["clv"]), a["scaler"].ctypes.data_as(ip),
                  _up(a["pmat"]), float(gtree.rate_mui))
    v._keep = a
    return v


def locus_update_all_matrices(locus, gtree):
    """locus_update_all_matrices (locus.c:1922): every branch of the gene tree (bpa_locus_update_all_matrices);
    node.length is stored as the reference stores it."""
    v = _view(gtree)
    lengths = np.zeros(len(gtree.nodes))
    _chk(lib().bpa_locus_update_all_matrices(locus.h, C.byref(v), _dp(lengths)))
    for nd in gtree.nodes:
        if nd.parent is not None:
            nd.length = float(lengths[nd.node_index])


def locus_update_all_partials(locus, gtree):
    """locus_update_all_partials (locus.c:2523, the post-order recursion of 2482): every inner node
    (bpa_locus_update_all_partials)."""
    v = _view(gtree)
    _chk(lib().bpa_locus_update_all_partials(locus.h, C.byref(v)))


def locus_root_loglikelihood(locus, root, persite=False):
    """locus_root_loglikelihood (locus.c:2573)."""
    return locus.root_loglikelihood(root.clv_index, root.scaler_index, persite)


class Sampler:
    """device-resident per-locus proposal control (bpa_sampler_t)"""

    def __init__(self, engine, loci, data, seed=1):
        L = lib()
        self.engine, self.n = engine, len(loci)
        self._arr = (C.c_void_p * self.n)(*[l.h for l in loci])
        self.h = L.bpa_sampler_create(engine.h, self._arr, self.n, seed)
        if not self.h:
            raise BpaError(_err())
        ip = C.POINTER(C.c_int)
        for k, d in enumerate(data):
            l = np.ascontiguousarray(d["left"], dtype=np.int32)
            r = np.ascontiguousarray(d["right"], dtype=np.int32)
            t = _f64(d["times"])
            _chk(L.bpa_sampler_set_tree(self.h, k, l.ctypes.data_as(ip), r.ctypes.data_as(ip), _dp(t), int(d["root"])))
        self.ntips = [len(d["seqs"]) for d in data]

    def set_species_tree(self, parent, tau, theta):
        """a00_set_species_tree's arrays (bpp_amd.synth.species_tree_arrays makes them)"""
        par = np.ascontiguousarray(parent, dtype=np.int32)
        self._npop = len(par)
        _chk(lib().bpa_sampler_set_species_tree(self.h, (len(par) + 1) // 2, par.ctypes.data_as(C.POINTER(C.c_int)),
                                                _dp(_f64(tau)), _dp(_f64(theta))))

    def set_tip_species(self, k, species):
        sp = np.ascontiguousarray(species, dtype=np.int32)
        _chk(lib().bpa_sampler_set_tip_species(self.h, k, sp.ctypes.data_as(C.POINTER(C.c_int))))

    def set_finetune(self, gage, gspr, tau, mix):
        lib().bpa_sampler_set_finetune(self.h, gage, gspr, tau, mix)

    FT_NAMES = ("gage", "gspr", "tau", "mix", "theta")

    def adapt_finetune(self):
        """the burn-in's step-length rule on the acceptance proportions since the last call (reset_finetune, method.c:1508-1516);
        -> (pjump, finetune) dicts by move type; pjump < 0: never proposed"""
        pj, ft = (C.c_double * 5)(), (C.c_double * 5)()
        _chk(lib().bpa_sampler_adapt_finetune(self.h, pj, ft))
        return dict(zip(self.FT_NAMES, pj)), dict(zip(self.FT_NAMES, ft))

    def burnin(self, iterations):
        """`iterations` iterations with the program's step-length resets (after every quarter and at the end, method.c:5364)"""
        ft = (C.c_double * 5)()
        _chk(lib().bpa_sampler_burnin(self.h, int(iterations), ft))
        return dict(zip(self.FT_NAMES, ft))

    def set_proposal_kernel(self, kind):
        """0 uniform windows on our streams (default), 1 BPP's legacy_rndu + Bactrian-Laplace (before initialize)"""
        _chk(lib().bpa_sampler_set_proposal_kernel(self.h, int(kind)))

    def set_program_moves(self, on, slide_prob=0.1):
        """THETA / TAU / MIX as the program runs them (BPP kernel): sliding window with probability slide_prob and the
        metropolized Gibbs draw otherwise, thetas re-drawn inside the rubber-band and the mixing proposals"""
        _chk(lib().bpa_sampler_set_program_moves(self.h, int(bool(on)), float(slide_prob)))

    def gibbs_counters(self):
        a, b = C.c_ulong(), C.c_ulong()
        _chk(lib().bpa_sampler_gibbs_counters(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_tau_prior(self, alpha, beta):
        lib().bpa_sampler_set_tau_prior(self.h, alpha, beta)

    def set_theta_prior(self, alpha, beta, finetune):
        lib().bpa_sampler_set_theta_prior(self.h, alpha, beta, finetune)

    def set_allreduce(self, fn, device_sum_ptr, first_locus):
        """fn(device_ptr:int, count:int, stream:int) -> truthy; enqueues the sum all-reduce of `count` device doubles
        (section 8e); device_sum_ptr: device memory for SAMPLER_SUMS doubles"""
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p)
        self._ar = proto(lambda ctx, p, n, st: 1 if fn(p, n, st) else 0)
        _chk(lib().bpa_sampler_set_allreduce(self.h, C.cast(self._ar, C.c_void_p), None,
                                             C.c_void_p(device_sum_ptr), first_locus))

    def set_allreduce_native(self, exchange, device_sum_ptr, first_locus):
        """the all-reduce as native code: `exchange` is an RcclExchange (libbpp_amd_rccl.so) — no Python runs inside iterate"""
        self._ar = exchange                       # (keeps the communicator alive)
        _chk(lib().bpa_sampler_set_allreduce(self.h, exchange.callback, exchange.h, C.c_void_p(device_sum_ptr), first_locus))

    def set_p2p(self, p2p, first_locus):
        """several GPUs with the sums exchanged inside the persistent kernel over the mailboxes of a connected P2P (None: off)"""
        self._p2p = p2p
        _chk(lib().bpa_sampler_set_p2p(self.h, p2p.h if p2p is not None else None, first_locus))

    def taus(self):
        out = np.zeros(getattr(self, "_npop", 0))
        if len(out):
            _chk(lib().bpa_sampler_get_taus(self.h, _dp(out)))
        return list(out)

    def thetas(self):
        out = np.zeros(getattr(self, "_npop", 0))
        if len(out):
            _chk(lib().bpa_sampler_get_thetas(self.h, _dp(out)))
        return list(out)

    def initialize(self):
        _chk(lib().bpa_sampler_initialize(self.h))

    def iterate(self, iterations=1):
        _chk(lib().bpa_sampler_iterate(self.h, iterations))

    def tree(self, k):
        n = 2 * self.ntips[k] - 1
        ip = C.POINTER(C.c_int)
        a = [np.zeros(n, dtype=np.int32) for _ in range(5)]
        t = np.zeros(n)
        root, lnl = C.c_int(), C.c_double()
        _chk(lib().bpa_sampler_get_tree(self.h, k, a[0].ctypes.data_as(ip), a[1].ctypes.data_as(ip),
                                        a[2].ctypes.data_as(ip), _dp(t), a[3].ctypes.data_as(ip),
                                        a[4].ctypes.data_as(ip), C.byref(root), C.byref(lnl)))
        pop, logpr = np.zeros(n, dtype=np.int32), C.c_double()
        _chk(lib().bpa_sampler_get_tree_msc(self.h, k, pop.ctypes.data_as(ip), C.byref(logpr)))
        return dict(left=list(a[0]), right=list(a[1]), parent=list(a[2]), time=list(t), clv=list(a[3]),
                    pmat=list(a[4]), root=root.value, lnl=lnl.value, pop=list(pop), logpr=logpr.value)

    def summary(self):
        tot = C.c_double()
        p, a, l = C.c_ulong(), C.c_ulong(), C.c_ulong()
        _chk(lib().bpa_sampler_summary(self.h, C.byref(tot), C.byref(p), C.byref(a), C.byref(l)))
        return dict(total_lnl=tot.value, proposals=p.value, accepted=a.value, launches=l.value)

    def set_subst_model(self, k, freqs, qrates, alpha):
        """starting values of locus k's substitution parameters (the moves of set_subst_moves change them)"""
        _chk(lib().bpa_sampler_set_subst_model(self.h, k, _dp(_f64(freqs)), _dp(_f64(qrates)), float(alpha)))

    def get_subst_model(self, k):
        f, q, a = np.zeros(4), np.zeros(6), C.c_double()
        _chk(lib().bpa_sampler_get_subst_model(self.h, k, _dp(f), _dp(q), C.byref(a)))
        return f, q, a.value

    def set_subst_moves(self, ft_freqs, ft_qrates, ft_alpha, alpha_a=1.0, alpha_b=1.0):
        """window widths of the per-locus frequency / exchangeability / alpha moves (0: off), gamma prior of alpha"""
        lib().bpa_sampler_set_subst_moves(self.h, ft_freqs, ft_qrates, ft_alpha, alpha_a, alpha_b)

    def enable_timing(self, stride=1):
        """HIP events on every stride-th sweep / all-loci launch (0: off)"""
        _chk(lib().bpa_sampler_enable_timing(self.h, int(stride)))

    def timing(self):
        a, b = C.c_double(), C.c_double()
        na, nb = C.c_ulong(), C.c_ulong()
        _chk(lib().bpa_sampler_timing(self.h, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
        return dict(sweep_ms=a.value, sweep_launches=na.value, allloci_ms=b.value, allloci_launches=nb.value)

    def work(self):
        """algorithmic work of the sweep launches so far"""
        by = C.c_double()
        nu, pu, sw = C.c_ulong(), C.c_ulong(), C.c_ulong()
        _chk(lib().bpa_sampler_work(self.h, C.byref(by), C.byref(nu), C.byref(pu), C.byref(sw)))
        return dict(bytes=by.value, node_updates=nu.value, pattern_updates=pu.value, sweeps=sw.value)

    def kind(self):
        """'sweep' (one launch per step), 'generic', 'persistent' (the whole iteration(s) of a call as one launch) or 'hybrid'
        (several ranks: the persistent kernel's sweep + one launch per all-loci step)"""
        k = lib().bpa_sampler_kind(self.h)
        if k < 0:
            raise BpaError(_err())
        return ("sweep", "generic", "persistent", "hybrid", "big", "composite")[k]

    def streams(self):
        """2 when the generic sampler runs its per-locus steps as two overlapping half-batch launches, else 1"""
        k = lib().bpa_sampler_streams(self.h)
        if k < 0:
            raise BpaError(_err())
        return k

    def close(self):
        if self.h and self.engine.h:
            lib().bpa_sampler_destroy(self.h)
        self.h = None


class RcclExchange:
    """include/bpp_amd_rccl.h: a native RCCL communicator whose sum all-reduce is a bpa_allreduce_fn.  rank 0 makes the
    id (RcclExchange.unique_id()), every rank gets its 128 bytes somehow and constructs with it (collective)."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbpp_amd_rccl.so")
            if not os.path.exists(path):
                raise BpaError("libbpp_amd_rccl.so is not built (python -m bpp_amd.build)")
            L = C.CDLL(path)
            L.bpa_rccl_unique_id.restype = C.c_int; L.bpa_rccl_unique_id.argtypes = [C.c_char_p]
            L.bpa_rccl_create.restype = C.c_void_p; L.bpa_rccl_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
            L.bpa_rccl_destroy.restype = None; L.bpa_rccl_destroy.argtypes = [C.c_void_p]
            L.bpa_rccl_allreduce_sum.restype = C.c_int; L.bpa_rccl_allreduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
            L.bpa_rccl_calls.restype = C.c_ulong; L.bpa_rccl_calls.argtypes = [C.c_void_p]
            L.bpa_rccl_last_error.restype = C.c_char_p
            cls._lib = L
        return cls._lib

    @classmethod
    def unique_id(cls):
        buf = C.create_string_buffer(128)
        if not cls.lib().bpa_rccl_unique_id(buf):
            raise BpaError(cls.lib().bpa_rccl_last_error().decode())
        return buf.raw

    def __init__(self, unique_id, nranks, rank, device):
        L = self.lib()
        self.h = L.bpa_rccl_create(C.c_char_p(bytes(unique_id)), nranks, rank, device)
        if not self.h:
            raise BpaError(L.bpa_rccl_last_error().decode())
        self.callback = C.cast(L.bpa_rccl_allreduce, C.c_void_p)

    def allreduce(self, device_ptr, count, stream=None):
        if not self.lib().bpa_rccl_allreduce_sum(self.h, C.c_void_p(device_ptr), count, C.c_void_p(stream)):
            raise BpaError(self.lib().bpa_rccl_last_error().decode())

    def calls(self):
        return self.lib().bpa_rccl_calls(self.h)

    def close(self):
        if self.h:
            self.lib().bpa_rccl_destroy(self.h)
        self.h = None


class PlanSequence:
    """several resident plans launched back to back with one host call (bpa_plans_launch)"""

    def __init__(self, plans):
        self.plans = list(plans)
        self.arr = (C.c_void_p * len(self.plans))(*[p.h for p in self.plans])
        self._fn = lib().bpa_plans_launch

    def launch(self):
        if not self._fn(self.arr, len(self.plans)):
            raise BpaError(_err())

    def launch_exchange(self, p2p, device_ptr, n):
        """launch, then the p2p all-reduce of the n doubles at device_ptr — one host call"""
        if not lib().bpa_plans_launch_exchange(self.arr, len(self.plans), p2p.h, device_ptr, n):
            raise BpaError(_err())


SAMPLER_SUMS = 16            # BPA_SAMPLER_SUMS
P2P_HANDLE_BYTES = 64        # BPA_P2P_HANDLE_BYTES


class P2P:
    """one-shot sum all-reduce of up to 512 doubles between the GPUs of a node (bpa_p2p_*): create on every rank,
    exchange the `handle` bytes, connect with all of them in rank order"""

    def __init__(self, engine, rank, world, max_doubles=512):
        self.engine = engine
        buf = C.create_string_buffer(P2P_HANDLE_BYTES)
        self.h = lib().bpa_p2p_create(engine.h, rank, world, max_doubles, buf)
        if not self.h:
            raise BpaError(_err())
        self.handle = buf.raw

    def connect(self, handles):
        blob = b"".join(handles)
        _chk(lib().bpa_p2p_connect(self.h, blob))

    def allreduce(self, device_ptr, n):
        if not lib().bpa_p2p_allreduce(self.h, device_ptr, n):
            raise BpaError(_err())

    def status(self):
        return lib().bpa_p2p_status(self.h)

    def set_timeout_ms(self, ms):
        """bound of a wait inside an exchange (default 3 s)"""
        lib().bpa_p2p_set_timeout(self.h, int(ms))

    def close(self):
        if self.h:
            lib().bpa_p2p_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """A resident batched proposal step (bpa_plan_t)."""

    def __init__(self, engine, loci, mat_off, mat_pmatrix, mat_length, op_off, ops, root_clv,
                 root_scaler=None):
        L = lib()
        self.engine = engine
        self.n = len(loci)
        self._keep = dict(
            loci=(C.c_void_p * self.n)(*[l.h for l in loci]),
            mat_off=_u32(mat_off), mat_pmatrix=_u32(mat_pmatrix), mat_length=_f64(mat_length),
            op_off=_u32(op_off), ops=np.ascontiguousarray(ops, dtype=OP_DTYPE),
            root_clv=_u32(root_clv),
            root_scaler=np.ascontiguousarray(
                np.full(self.n, SCALE_BUFFER_NONE) if root_scaler is None else root_scaler,
                dtype=np.int32))
        k = self._keep
        b = Batch(self.n, k["loci"], _up(k["mat_off"]), _up(k["mat_pmatrix"]), _dp(k["mat_length"]),
                  _up(k["op_off"]), k["ops"].ctypes.data_as(C.POINTER(Op)), _up(k["root_clv"]),
                  k["root_scaler"].ctypes.data_as(C.POINTER(C.c_int)))
        self.h = L.bpa_plan_create(engine.h, C.byref(b))
        if not self.h:
            raise BpaError(_err())

    def set_lengths(self, lengths):
        lengths = _f64(lengths)
        _chk(lib().bpa_plan_set_lengths(self.h, _dp(lengths)))

    def launch(self):
        _chk(lib().bpa_plan_launch(self.h))

    def lnl(self):
        out = np.zeros(self.n)
        _chk(lib().bpa_plan_get_lnl(self.h, _dp(out)))
        return out

    def set_params(self, which, values):
        """batched substitution-parameter proposal: which = PARAM_FREQS | PARAM_SUBST | PARAM_RATES, values[locus][...]"""
        v = _f64(values)
        _chk(lib().bpa_plan_set_params(self.h, which, _dp(v)))

    def set_params_device(self, which, device_ptr):
        _chk(lib().bpa_plan_set_params_device(self.h, which, C.c_void_p(device_ptr)))

    def enable_sum(self, device_ptr=None):
        _chk(lib().bpa_plan_enable_sum(self.h, device_ptr))

    def enable_partial_sums(self, device_ptr=None, capacity=0):
        """the plan's sum as per-workgroup partial sums written by the step kernel itself; returns their number"""
        n = C.c_uint(capacity)
        _chk(lib().bpa_plan_enable_partial_sums(self.h, device_ptr, C.byref(n)))
        return n.value

    def lnl_sum(self):
        v = C.c_double()
        _chk(lib().bpa_plan_get_sum(self.h, C.byref(v)))
        return v.value

    def work(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        n, p = C.c_ulong(), C.c_ulong()
        lib().bpa_plan_work(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(n), C.byref(p))
        bc = C.c_double()
        lib().bpa_plan_work_codes(self.h, C.byref(bc))
        return {"bytes_partials": a.value, "flops_partials": b.value, "bytes_pmatrix": c.value,
                "node_updates": n.value, "pattern_updates": p.value, "bytes_codes": bc.value}

    def close(self):
        if self.h and self.engine.h:
            lib().bpa_plan_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
