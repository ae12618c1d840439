"""End-to-end known answer: the posterior of the UNMODIFIED reference program (bpp A00, JC69; fixture
tests/golden/a00_posterior.json from tests/golden/make_golden_a00.py) on a synthetic 30-locus data set,
against this repo's sampler on the same data and priors (thetaprior gamma 2 500, tauprior gamma 2 400).
Different proposal kernels and random numbers, same target: means and spreads of every theta, every tau
and the log-likelihood must agree within Monte-Carlo error.

 * CPU: the C host driver on the REAL reference's locus API (skipped where oracle/_ref is absent);
 * GPU: the device-resident sampler (bpa_sampler_t) — every proposal, density and decision on the device.
"""
import json
import os

import numpy as np
import pytest

from bpp_amd import synth
import oraclelib as O
import hostdrv

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["theta_AB", "theta_ABC", "theta_root", "tau_AB", "tau_ABC", "tau_root", "lnL"]


def setup(drv, gold):
    c = gold["config"]
    parent, tau, thetas = synth.species_tree_arrays(c["taxa"], c["theta"])
    drv.set_species_tree(parent, tau, thetas)
    drv.set_tau_prior(*c["tau_prior"])
    drv.set_theta_prior(c["theta_prior"][0], c["theta_prior"][1], 0.004)
    drv.set_finetune(0.004, 0.004, 0.0012, 0.3)


def compare(samples, gold):
    S = np.array(samples)
    for k, nm in enumerate(NAMES):
        ref = gold["posterior"][nm]
        x = S[:, k]
        tol = 0.25 * ref["sd"] if nm != "lnL" else 0.2 * ref["sd"]      # a quarter of a posterior sd: MC error of both chains
        assert abs(x.mean() - ref["mean"]) < tol, (nm, x.mean(), ref["mean"])
        assert abs(x.std() - ref["sd"]) < 0.2 * ref["sd"], (nm, x.std(), ref["sd"])


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(HERE, "golden", "a00_posterior.json")))


def dataset(gold):
    c = gold["config"]
    return synth.make_dataset(c["nloci"], c["sites"], c["taxa"], "jc69", 1, seed=c["seed"], theta=c["theta"])


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_host_driver_on_reference_backend_reproduces_bpp_posterior(gold):
    data = dataset(gold)
    drv = hostdrv.reference_driver(data, seed=5)
    setup(drv, gold)
    drv.initialize()
    S = []
    for it in range(16000):
        drv.iterate()
        if it >= 3000 and it % 2 == 0:
            S.append(drv.thetas()[4:] + drv.taus()[4:] + [drv.total_lnl()])
    compare(S, gold)
    drv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("program", [False, True])
def test_device_sampler_reproduces_bpp_posterior(gold, program):
    """program: BPP's own generator and windows and its THETA / TAU / MIX move for move (bpa_sampler_set_program_moves)"""
    import bpp_amd
    import tape
    data = dataset(gold)
    eng = bpp_amd.Engine(0)
    loci = tape.make_engine_loci(eng, data)
    dev = bpp_amd.Sampler(eng, loci, data, seed=9)
    if program:
        dev.set_proposal_kernel(1)
        dev.set_program_moves(True, 0.1)
    setup(dev, gold)
    if program:
        c = gold["config"]
        dev.set_theta_prior(c["theta_prior"][0], c["theta_prior"][1], 0.0012)
        dev.set_finetune(0.0012, 0.0012, 0.0004, 0.3)            # (step lengths of the order BPP's burn-in tuning settles on)
    dev.initialize()
    dev.iterate(3000)
    S = []
    for _ in range(6500):
        dev.iterate(2)
        S.append(dev.thetas()[4:] + dev.taus()[4:] + [dev.summary()["total_lnl"]])
    compare(S, gold)
    dev.close(); eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("program", [False, True])
def test_device_sampler_reproduces_bpp_posterior_at_benchmark_scale(program):
    """BASELINE config 2 itself — 10 000 loci x 1 000 sites: the unmodified program (8 threads, ~3 minutes) and the
    device-resident sampler (seconds) land on the same posterior; with this much data it is 1-4 % wide, so the
    comparison is tight in absolute terms.  program: BPP's kernel with the program's moves and the step lengths the
    program's own burn-in arrives at on such data"""
    import bpp_amd
    import tape
    gold = json.load(open(os.path.join(HERE, "golden", "a00_posterior_10k.json")))
    c = gold["config"]
    data = synth.make_dataset(c["nloci"], c["sites"], c["taxa"], "jc69", 1, seed=c["seed"])
    eng = bpp_amd.Engine(0)
    loci = tape.make_engine_loci(eng, data)
    dev = bpp_amd.Sampler(eng, loci, data, seed=3)
    if program:
        dev.set_proposal_kernel(1)
        dev.set_program_moves(True, 0.1)
    dev.set_species_tree(*synth.species_tree_arrays(c["taxa"]))
    dev.set_theta_prior(c["theta_prior"][0], c["theta_prior"][1], 3e-5 if program else 8e-5)
    dev.set_tau_prior(*c["tau_prior"])
    if program:
        dev.set_finetune(18.5, 0.0019, 1.9e-5, 0.0059)
    else:
        dev.set_finetune(0.004, 0.004, 4e-5, 0.006)
    dev.initialize()
    dev.iterate(1000)
    S = []
    for _ in range(6000):
        dev.iterate(1)
        S.append(dev.thetas()[4:] + dev.taus()[4:])
    S = np.array(S)
    for k, nm in enumerate(NAMES[:6]):
        ref = gold["posterior"][nm]
        assert abs(S[:, k].mean() - ref["mean"]) < 0.4 * ref["sd"], (nm, S[:, k].mean(), ref["mean"])       # both chains: ESS ~ 150
        assert abs(S[:, k].std() - ref["sd"]) < 0.3 * ref["sd"], (nm, S[:, k].std(), ref["sd"])
        assert abs(S[:, k].mean() - ref["mean"]) < 0.012 * ref["mean"], nm
    lnl = dev.summary()["total_lnl"]
    assert abs(lnl - gold["posterior"]["lnL"]["mean"]) < 5 * gold["posterior"]["lnL"]["sd"]
    dev.close(); eng.close()
