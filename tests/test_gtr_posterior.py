"""End-to-end known answer for the substitution-parameter moves (base frequencies, exchangeabilities, alpha:
locus.c:2782-3419, prop_gamma.c:52-224) next to the tree moves: the posterior of the UNMODIFIED reference program on a
synthetic 20-locus 8-species GTR + Gamma4 data set (fixture tests/golden/gtr_posterior.json from
tests/golden/make_golden_gtr.py: thetaprior gamma 2 500, tauprior gamma 2 300, alphaprior 1 1 4) against this repo's
samplers on the same data and priors — every theta, every tau and the log-likelihood within Monte-Carlo error.

 * CPU: the C host driver on the REAL reference's locus API (skipped where oracle/_ref is absent);
 * GPU: the generic device-resident sampler (csrc/gsampler.hpp) — every proposal, parameter, eigensystem, density and
   decision on the device.
"""
import json
import os

import numpy as np
import pytest

from bpp_amd import synth
import oraclelib as O
import hostdrv

HERE = os.path.dirname(os.path.abspath(__file__))
POP_OF = {"A,B": 8, "C,D": 9, "A,B,C,D": 10, "E,F": 11, "G,H": 12, "E,F,G,H": 13, "A,B,C,D,E,F,G,H": 14}


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(HERE, "golden", "gtr_posterior.json")))


def dataset(gold):
    c = gold["config"]
    return synth.make_dataset(c["nloci"], c["sites"], c["taxa"], "gtr", c["rate_cats"], seed=c["seed"], theta=c["theta"])


def setup(drv, gold, data, host):
    c = gold["config"]
    parent, tau, thetas = synth.species_tree_arrays(c["taxa"], c["theta"])
    drv.set_species_tree(parent, tau, thetas)
    drv.set_tau_prior(*c["tau_prior"])
    drv.set_theta_prior(c["theta_prior"][0], c["theta_prior"][1], 0.004)
    drv.set_finetune(0.004, 0.004, 0.0012, 0.2)
    for i, d in enumerate(data):
        # BPP's start: equal frequencies and rates, alpha at its prior mean (locus.c:901; locus.c:695)
        if host:
            drv.set_subst_model(i, [0.25]*4, [1.0]*6, c["alpha_prior"][0]/c["alpha_prior"][1], c["rate_cats"])
        else:
            drv.set_subst_model(i, [0.25]*4, [1.0]*6, c["alpha_prior"][0]/c["alpha_prior"][1])
    drv.set_subst_moves(0.5, 0.6, 1.2, c["alpha_prior"][0], c["alpha_prior"][1])


def compare(samples, gold, tol_mean=0.3, tol_sd=0.25):
    S = np.array(samples)                  # columns: thetas[8..14], taus[8..14], lnL
    for name, ref in gold["posterior"].items():
        if name == "lnL":
            x = S[:, -1]
        else:
            kind, _, label = name.split(":")
            x = S[:, POP_OF[label] - 8 + (0 if kind == "theta" else 7)]
        assert abs(x.mean() - ref["mean"]) < tol_mean*ref["sd"], (name, x.mean(), ref["mean"], ref["sd"])
        assert abs(x.std() - ref["sd"]) < tol_sd*ref["sd"], (name, x.std(), ref["sd"])


def start_params(drv_or_loci, data, gold):
    return


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_host_driver_with_parameter_moves_reproduces_bpp_posterior(gold):
    data = dataset(gold)
    c = gold["config"]
    # the loci start from BPP's values, not from the values the data were simulated with
    for d in data:
        d["freqs"] = np.full(4, 0.25); d["exch"] = np.ones(6)
        d["rates"] = np.asarray(__import__("bpp_amd").compute_gamma_cats(1.0, 1.0, c["rate_cats"]))
    drv = hostdrv.reference_driver(data, seed=5)
    setup(drv, gold, data, True)
    drv.initialize()
    S = []
    for it in range(9000):
        drv.iterate()
        if it >= 2000 and it % 2 == 0:
            S.append(drv.thetas()[8:] + drv.taus()[8:] + [drv.total_lnl()])
    compare(S, gold)
    p, a, _ = drv.counters()
    assert 0.15 < a / p < 0.9
    f, q, alpha = drv.get_subst_model(0)
    assert abs(sum(f) - 1) < 1e-12 and min(f) > 0 and alpha > 0 and f != [0.25]*4 and q != [1.0]*6
    drv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("moves", ["uniform", "program"])
def test_device_sampler_with_parameter_moves_reproduces_bpp_posterior(gold, moves):
    """moves = "program": BPP's own iteration on the generic sampler — its generator, Bactrian-Laplace windows and acceptance rule
    in every per-locus move (tree and parameter moves), THETA by the metropolized Gibbs draw, the thetas re-drawn inside the
    rubber band and the mixing step (decided on the host from the device's sums), the step lengths tuned by its burn-in rule
    from its defaults (bpa_sampler_burnin) — must sample the same posterior as the program"""
    import bpp_amd
    import tape
    data = dataset(gold)
    c = gold["config"]
    for d in data:
        d["freqs"] = np.full(4, 0.25); d["exch"] = np.ones(6)
        d["rates"] = np.asarray(bpp_amd.compute_gamma_cats(1.0, 1.0, c["rate_cats"]))
    eng = bpp_amd.Engine(0)
    loci = tape.make_engine_loci(eng, data)
    dev = bpp_amd.Sampler(eng, loci, data, seed=9)
    if moves == "program":
        dev.set_proposal_kernel(1)
        dev.set_program_moves(True, 0.1)
    setup(dev, gold, data, False)
    if moves == "program":
        dev.set_theta_prior(c["theta_prior"][0], c["theta_prior"][1], 0.001)       # the program's defaults (bpp.c:530-549)
        dev.set_finetune(5.0, 0.001, 0.001, 0.3)
    dev.initialize()
    if moves == "program":
        ft = dev.burnin(2000)
        assert ft["gage"] != 5.0 and ft["mix"] != 0.3
        g = dev.gibbs_counters()
        assert g[0] > 0 and g[1] > 0.8 * g[0]
    else:
        dev.iterate(2000)
    S = []
    for _ in range(4000):
        dev.iterate(2)
        S.append(dev.thetas()[8:] + dev.taus()[8:] + [dev.summary()["total_lnl"]])
    compare(S, gold)
    f, q, alpha = dev.get_subst_model(0)
    assert abs(f.sum() - 1) < 1e-12 and f.min() > 0 and alpha > 0 and not np.allclose(f, 0.25) and not np.allclose(q, 1.0)
    dev.close(); eng.close()
