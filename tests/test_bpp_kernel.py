"""The reference's own proposal kernel in the C host driver (a00_set_proposal_kernel(d, A00_KERNEL_BPP),
include/bpp_amd_host.h): legacy_rndu (random.c:104-122) and the Bactrian-Laplace window of legacy_rnd_symmetrical
(random.c:192-238) restated, and BPP's acceptance rule (gtree.c:5476).

 * the generator and the window variate are BIT-EQUAL to the reference's functions on the same state (libbppref.so);
 * with that kernel the gene trees sampled without data reproduce direct MSC simulation, and the host driver on the
   reference's likelihood reproduces the unmodified program's posterior (tests/golden/a00_posterior.json) — the same
   checks the default kernel passes (tests/test_msc_density.py, tests/test_a00_posterior.py).
BPP's trajectory itself (byte-identical mcmc.txt) is what oracle/_ref/bpp_hip shows for the LIKELIHOOD under the
reference's own control flow (tests/test_gpu_bpp_hip.py); a batched driver cannot walk it: see a00_driver.c's header."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from bpp_amd import synth
import oraclelib as O
import hostdrv
from test_a00_posterior import compare, dataset
from test_msc_density import topology_key

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", [1, 12345, 4294967295, 2654435769])
def test_generator_and_window_bit_equal_to_the_reference(seed):
    R = O.ref()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    R.legacy_rndu.restype = C.c_double
    R.legacy_rndu.argtypes = [C.c_long]
    R.legacy_rnd_symmetrical.restype = C.c_double
    R.legacy_rnd_symmetrical.argtypes = [C.c_long]
    R.set_legacy_rndu_array.argtypes = [C.c_void_p]
    L = hostdrv.lib()
    L.a00_bpp_kernel_sequence.argtypes = [C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_double)]
    n = 5000
    for symmetrical in (0, 1):
        z = libc.malloc(16)                                     # the reference takes ownership (set_legacy_rndu_array frees the old one)
        C.cast(z, C.POINTER(C.c_uint))[0] = seed
        R.set_legacy_rndu_array(z)
        want = np.array([R.legacy_rnd_symmetrical(0) if symmetrical else R.legacy_rndu(0) for _ in range(n)])
        got = np.zeros(n)
        L.a00_bpp_kernel_sequence(seed, symmetrical, n, got.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    x = got
    assert abs(x.mean()) < 0.05 and abs(x.var() - 1.0) < 0.08                    # mean 0, variance 1 (random.c:203-205)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", [7, 2654435769])
def test_gibbs_theta_pieces_bit_equal_to_the_reference(seed):
    """the three pieces of the program's metropolized Gibbs draw of a theta (propose_theta_gibbs, stree.c:3645): rndNormal
    (random.c:215), legacy_rndgamma (random.c:240) and the inverse-gamma fit get_gamma_conditional_approx (stree.c:3384,
    opt_theta_prop = MG_INVG under a gamma prior) — restated in include/bpp_amd_host.h for the host driver AND the device
    kernel, bit-equal to the reference's functions on the same state / arguments"""
    R = O.ref()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    R.rndNormal.restype = C.c_double; R.rndNormal.argtypes = [C.c_long]
    R.legacy_rndgamma.restype = C.c_double; R.legacy_rndgamma.argtypes = [C.c_long, C.c_double]
    R.set_legacy_rndu_array.argtypes = [C.c_void_p]
    L = hostdrv.lib()
    L.a00_bpp_kernel_sequence.argtypes = [C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.a00_bpp_gamma_sequence.argtypes = [C.c_uint, C.c_double, C.c_int, C.POINTER(C.c_double)]
    n = 3000

    def reseed():
        z = libc.malloc(16)
        C.cast(z, C.POINTER(C.c_uint))[0] = seed
        R.set_legacy_rndu_array(z)
    reseed()
    want = np.array([R.rndNormal(0) for _ in range(n)])
    got = np.zeros(n)
    L.a00_bpp_kernel_sequence(seed, 2, n, got.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    for shape in (0.3, 1.0, 2.5, 31.0, 1234.5, 30001.0):
        reseed()
        want = np.array([R.legacy_rndgamma(0, shape) for _ in range(n)])
        L.a00_bpp_gamma_sequence(seed, shape, n, got.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), shape
        assert abs(got.mean() / shape - 1) < 0.1
    # the conditional's inverse-gamma fit
    C.c_long.in_dll(R, "opt_theta_prior").value = 2                     # BPP_THETA_PRIOR_GAMMA (bpp.h:275)
    C.c_long.in_dll(R, "opt_theta_prop").value = 1                      # BPP_THETA_PROP_MG_INVG (bpp.h:277)
    R.get_gamma_conditional_approx.argtypes = [C.c_double, C.c_double, C.c_long, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.a00_theta_conditional.argtypes = [C.c_double, C.c_double, C.c_long, C.c_double, C.POINTER(C.c_double)]
    rng = np.random.default_rng(seed)
    for _ in range(300):
        a, b = float(rng.choice([2.0, 3.0, 21.0])), float(rng.choice([100.0, 1000.0, 2000.0]))
        k = int(rng.integers(0, 40000))
        T = float(rng.choice([0.0, rng.uniform(0, 1e-3), rng.uniform(0, 80.0)]))
        a1, b1 = C.c_double(), C.c_double()
        R.get_gamma_conditional_approx(a, b, k, T, C.byref(a1), C.byref(b1))
        mine = np.zeros(2)
        L.a00_theta_conditional(a, b, k, T, mine.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(mine.view(np.uint64), np.array([a1.value, b1.value]).view(np.uint64)), (a, b, k, T)
    # ... and the comparison-driven form the device samplers run (a00_theta_conditional_invgamma_fast): the same bits as the
    # plain bisection on a wide sweep of cases, incl. tiny data sets (the plain loop's domain) and huge ones
    L.a00_theta_conditional_fast.argtypes = L.a00_theta_conditional.argtypes
    fast, plain = np.zeros(2), np.zeros(2)
    nfast = 0
    for i in range(100000):
        a, b = float(rng.choice([2.0, 3.0, 21.0])), float(rng.choice([100.0, 1000.0, 2000.0, 7.0]))
        k = int(rng.choice([rng.integers(0, 6), rng.integers(0, 300), rng.integers(0, 40000), rng.integers(0, 3000000)]))
        T = float(rng.choice([0.0, rng.uniform(0, 1e-3), rng.uniform(0, 80.0), rng.uniform(0, 1e4), 10.0**rng.uniform(-12, 3)]))
        L.a00_theta_conditional(a, b, k, T, plain.ctypes.data_as(C.POINTER(C.c_double)))
        L.a00_theta_conditional_fast(a, b, k, T, fast.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(fast.view(np.uint64), plain.view(np.uint64)), (a, b, k, T, fast, plain)


def test_prior_sampling_with_the_bpp_kernel_matches_direct_msc_simulation():
    taxa, theta = 4, 0.004
    nloci, iters, burn = 400, 60, 10
    rng = np.random.default_rng(6)
    parent, tau, thetas = synth.species_tree_arrays(taxa, theta)
    start = [synth._msc_gene_tree(synth.SPECIES_TREES[taxa], theta, rng) for _ in range(nloci)]
    data = [dict(seqs=["A"] * taxa, left=l, right=r, times=t, root=rt) for l, r, t, rt in start]
    drv = hostdrv.prior_driver(data, seed=12)
    drv.set_proposal_kernel(1)
    drv.set_species_tree(parent, tau, thetas)
    drv.set_finetune(1.2 * theta, 1.2 * theta, 0.0, 0.0)           # variance-1 window variate: ~ the uniform kernel's 4 theta
    drv.initialize()
    ages, tops = [], {}
    for it in range(iters):
        drv.iterate()
        if it < burn:
            continue
        for i in range(nloci):
            t = drv.tree(i)
            ages.append(sorted(t["time"][taxa:]))
            k = topology_key(t, taxa)
            tops[k] = tops.get(k, 0) + 1
    ages = np.array(ages)
    sim, stops = [], {}
    for _ in range(40000):
        l, r, t, rt = synth._msc_gene_tree(synth.SPECIES_TREES[taxa], theta, rng)
        sim.append(sorted(t[taxa:]))
        k = topology_key(dict(left=l, right=r), taxa)
        stops[k] = stops.get(k, 0) + 1
    sim = np.array(sim)
    for j in range(taxa - 1):
        assert abs(ages[:, j].mean() - sim[:, j].mean()) < 0.04 * sim[:, j].mean(), (j, ages[:, j].mean(), sim[:, j].mean())
        assert abs(ages[:, j].std() - sim[:, j].std()) < 0.08 * sim[:, j].std() + 1e-12, j
    n_mc, n_sim = sum(tops.values()), sum(stops.values())
    for k, c in stops.items():
        f_sim, f_mc = c / n_sim, tops.get(k, 0) / n_mc
        if f_sim > 0.02:
            assert abs(f_mc - f_sim) < 0.03 + 0.15 * f_sim, (f_mc, f_sim)
    p, a, _ = drv.counters()
    assert 0.1 < a / p < 0.95
    drv.close()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("slide_prob", [None, 0.1, 1.0])
def test_host_driver_with_the_bpp_kernel_reproduces_bpp_posterior(slide_prob):
    """slide_prob not None: THETA / TAU / MIX as the program runs them (a00_set_program_moves) — 0.1 is the program's own THETA mix
    (stree.c:3957: sliding window 1 time in 10, metropolized Gibbs draw otherwise); thetas re-drawn inside TAU and MIX"""
    gold = json.load(open(os.path.join(HERE, "golden", "a00_posterior.json")))
    c = gold["config"]
    data = dataset(gold)
    drv = hostdrv.reference_driver(data, seed=6)
    drv.set_proposal_kernel(1)
    drv.set_program_moves(slide_prob is not None, slide_prob or 0.0)     # THETA / TAU / MIX as the program runs them
    parent, tau, thetas = synth.species_tree_arrays(c["taxa"], c["theta"])
    drv.set_species_tree(parent, tau, thetas)
    drv.set_tau_prior(*c["tau_prior"])
    drv.set_theta_prior(c["theta_prior"][0], c["theta_prior"][1], 0.0012)
    drv.set_finetune(0.0012, 0.0012, 0.0004, 0.3)                 # step lengths of the order BPP's burn-in tuning settles on
    drv.initialize()
    S = []
    for it in range(16000):
        drv.iterate()
        if it >= 3000 and it % 2 == 0:
            S.append(drv.thetas()[4:] + drv.taus()[4:] + [drv.total_lnl()])
    compare(S, gold)
    p, a, _ = drv.counters()
    assert 0.15 < a / p < 0.9
    gp, ga = drv.gibbs_counters()
    if slide_prob is not None and slide_prob < 1:
        assert 0.85 < gp / (16000 * 3 * 0.9) < 1.15 and 0.3 < ga / gp <= 1.0, (gp, ga)
    else:
        assert gp == 0
    drv.close()
