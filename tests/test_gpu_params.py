"""Batched substitution-parameter proposals (bpa_plan_set_params: frequencies / exchangeabilities / category rates
of all loci of a plan in one transfer + one kernel that also refreshes the eigensystems on the device, K6) against
the per-locus setters and against the oracle (whose eigensystem and P-matrices are pinned to the reference)."""
import numpy as np
import pytest

import bpp_amd
from bpp_amd import synth
from bpp_amd.api import PARAM_FREQS, PARAM_SUBST, PARAM_RATES
import oraclelib as O
import tape
from common import rel, lg_model
from test_gpu_parity import build_batch

pytestmark = pytest.mark.gpu


def full_plan(engine, loci, data):
    trees = [bpp_amd.GTree(d["left"], d["right"], d["times"], d["root"]) for d in data]
    return bpp_amd.Plan(engine, loci, *build_batch(loci, trees))


@pytest.mark.parametrize("taxa,model,R", [(8, "gtr", 4), (4, "gtr", 1), (6, "lg", 4)])
def test_batched_params_equal_per_locus_setters_and_oracle(engine, taxa, model, R):
    nloci = 40 if model != "lg" else 6
    data = synth.make_dataset(nloci, 300 if model != "lg" else 120, taxa, model, R, seed=13)
    a = tape.make_engine_loci(engine, data)
    b = tape.make_engine_loci(engine, data)
    pa, pb = full_plan(engine, a, data), full_plan(engine, b, data)
    rng = np.random.default_rng(3)
    S = 4 if model != "lg" else 20
    for rnd in range(3):
        freqs = rng.dirichlet(np.full(S, 20.0), nloci)
        exch = np.exp(rng.normal(0, 0.3, (nloci, S * (S - 1) // 2)))
        exch[:, -1] = 1.0
        alpha = rng.uniform(0.3, 2.0, nloci)
        rates = np.array([bpp_amd.compute_gamma_cats(x, x, R) for x in alpha])
        if rnd != 1:
            pa.set_params(PARAM_FREQS, freqs)
        if rnd != 2:
            pa.set_params(PARAM_SUBST, exch)
        if R > 1:
            pa.set_params(PARAM_RATES, rates)
        for i, l in enumerate(b):
            if rnd != 1:
                l.set_frequencies(0, freqs[i])
            if rnd != 2:
                l.set_subst_params(0, exch[i])
            if R > 1:
                l.set_category_rates(rates[i])
        pa.launch(); pb.launch()
        la, lb = pa.lnl(), pb.lnl()
        assert (la == lb).all(), np.max(np.abs(la - lb))
        # the eigensystems the device refreshed are the per-locus path's, bit for bit
        ea, eb = a[0].get_eigen(0), b[0].get_eigen(0)
        assert all((x == y).all() for x, y in zip(ea, eb))
        if model == "gtr":
            cur_f = freqs if rnd != 1 else cur_f
            cur_e = exch if rnd != 2 else cur_e
            for i in range(0, nloci, 9):
                d = data[i]
                ol = O.OracleLocus(4, R, d["seqs"], d["weights"], model="gtr", freqs=cur_f[i], qrates=cur_e[i],
                                   rates=rates[i] if R > 1 else d["rates"])
                assert rel(la[i], ol.full_lnl(d["left"], d["right"], d["times"], d["root"])) < 1e-12
        else:
            cur_f, cur_e = freqs, exch
    # a per-locus setter after batched updates starts from what the device holds
    a[1].set_category_rates(np.ones(R)); b[1].set_category_rates(np.ones(R))
    pa.launch(); pb.launch()
    assert (pa.lnl() == pb.lnl()).all()
    pa.close(); pb.close()


def test_device_resident_values(engine):
    """the _device form: values already in HBM (a tape), nothing crosses PCIe"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")                  # the runtime libbpp_amd.so itself is linked against
    data = synth.make_dataset(30, 200, 8, "gtr", 4, seed=5)
    a = tape.make_engine_loci(engine, data)
    b = tape.make_engine_loci(engine, data)
    pa, pb = full_plan(engine, a, data), full_plan(engine, b, data)
    rng = np.random.default_rng(1)
    freqs = rng.dirichlet(np.full(4, 30.0), 30)
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(freqs.nbytes)) == 0
    assert hip.hipMemcpy(dptr, freqs.ctypes.data_as(C.c_void_p), C.c_size_t(freqs.nbytes), 1) == 0      # hipMemcpyHostToDevice
    pa.set_params_device(PARAM_FREQS, dptr.value)
    pb.set_params(PARAM_FREQS, freqs)
    pa.launch(); pb.launch()
    assert (pa.lnl() == pb.lnl()).all()
    exch = np.exp(rng.normal(0, 0.2, (30, 6)))
    for i, l in enumerate(a):                       # a host-side setter now: the mirror must catch up with the device first
        l.set_subst_params(0, exch[i])
    pb.set_params(PARAM_SUBST, exch)
    pa.launch(); pb.launch()
    assert (pa.lnl() == pb.lnl()).all()
    pa.close(); pb.close()
    hip.hipFree(dptr)
