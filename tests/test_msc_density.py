"""MSC density and the sampler's validity on the CPU (no likelihood back-end needed).

 * a00_msc_contrib / the driver's tree density == the REAL reference's gtree_logprob
   (gtree.c:3957, 3859) bit for bit on the golden trees (tests/golden/msc_density.json, generated
   through oracle/ref_shim_input.c: ref_msc_logpr), including several sequences per species;
 * with lnL = 0 (a00_backend_prior, BPP's usedata = 0) the sampler's gene trees follow the
   multispecies coalescent: moments of the node ages and the topology frequencies agree with direct
   simulation from the MSC (bpp_amd.synth) — which the GAGE / GSPR / TAU / MIX moves can only do if
   their bounds, population bookkeeping, Hastings ratios and Jacobians are right.
"""
import json
import os

import numpy as np
import pytest

from bpp_amd import synth
import hostdrv

HERE = os.path.dirname(os.path.abspath(__file__))
fh = float.fromhex


def one_locus_driver(case):
    tips = case["tips"]
    data = [dict(seqs=["A"] * tips, left=case["left"], right=case["right"], times=[fh(x) for x in case["time"]],
                 root=case["root"])]
    drv = hostdrv.prior_driver(data, seed=3)
    drv.set_species_tree(case["parent"], [fh(x) for x in case["tau"]], [fh(x) for x in case["theta"]])
    drv.set_tip_species(0, case["tip_species"])
    drv.initialize()
    return drv


def test_density_bit_exact_against_reference_golden():
    cases = json.load(open(os.path.join(HERE, "golden", "msc_density.json")))
    assert len(cases) >= 19
    for c in cases:
        drv = one_locus_driver(c)
        t = drv.tree(0)
        assert t["pop"] == c["pop"]
        assert t["logpr"] == fh(c["logpr"]) == drv.logpr(0), (t["logpr"], fh(c["logpr"]))
        assert sum(fh(x) for x in c["contrib"]) == pytest.approx(fh(c["logpr"]), rel=1e-15)
        drv.close()


def test_incompatible_tree_is_refused():
    parent, tau, theta = synth.species_tree_arrays(4)
    # A and D coalesce at 0.0005, below the root divergence 0.003
    data = [dict(seqs=["A"] * 4, left=[-1, -1, -1, -1, 0, 4, 5], right=[-1, -1, -1, -1, 3, 1, 2],
                 times=[0, 0, 0, 0, 0.0005, 0.004, 0.005], root=6)]
    drv = hostdrv.prior_driver(data)
    drv.set_species_tree(parent, tau, theta)
    assert not hostdrv.lib().a00_initialize(drv.h)
    drv.close()
    drv = hostdrv.prior_driver(data)
    assert not hostdrv.lib().a00_initialize(drv.h)             # no species tree at all
    drv.close()


def topology_key(t, tips):
    def clade(v):
        return frozenset([v]) if v < tips else clade(t["left"][v]) | clade(t["right"][v])
    return frozenset(clade(v) for v in range(tips, 2 * tips - 1))


@pytest.mark.parametrize("taxa,theta", [(4, 0.002), (4, 0.008)])
def test_prior_sampling_matches_direct_msc_simulation(taxa, theta):
    """gene trees only (no TAU/MIX: the species tree is fixed so the target is exactly the MSC)"""
    nloci, iters, burn = 400, 60, 10
    rng = np.random.default_rng(5)
    parent, tau, thetas = synth.species_tree_arrays(taxa, theta)
    start = [synth._msc_gene_tree(synth.SPECIES_TREES[taxa], theta, rng) for _ in range(nloci)]
    data = [dict(seqs=["A"] * taxa, left=l, right=r, times=t, root=rt) for l, r, t, rt in start]
    drv = hostdrv.prior_driver(data, seed=11)
    drv.set_species_tree(parent, tau, thetas)
    drv.set_finetune(4 * theta, 4 * theta, 0.0, 0.0)               # tau window 0, mix window 0: identity moves
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
    assert drv.taus() == list(tau)
    ages = np.array(ages)
    # direct simulation
    sim, stops = [], {}
    for _ in range(40000):
        l, r, t, rt = synth._msc_gene_tree(synth.SPECIES_TREES[taxa], theta, rng)
        sim.append(sorted(t[taxa:]))
        k = topology_key(dict(left=l, right=r), taxa)
        stops[k] = stops.get(k, 0) + 1
    sim = np.array(sim)
    # samples of one locus are autocorrelated: compare with a tolerance of a few per cent of the mean
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


def test_tau_and_mix_moves_leave_the_joint_prior_invariant():
    """with a flat prior on the taus the joint density is the MSC density itself; after TAU and MIX moves
    the carried densities must still equal from-scratch ones and every gene node must sit in the
    population its age puts it in (the moves re-scale ages across population boundaries)"""
    taxa, theta, nloci = 8, 0.002, 50
    rng = np.random.default_rng(9)
    parent, tau, thetas = synth.species_tree_arrays(taxa, theta)
    start = [synth._msc_gene_tree(synth.SPECIES_TREES[taxa], theta, rng) for _ in range(nloci)]
    data = [dict(seqs=["A"] * taxa, left=l, right=r, times=t, root=rt) for l, r, t, rt in start]
    drv = hostdrv.prior_driver(data, seed=2)
    drv.set_species_tree(parent, tau, thetas)
    drv.set_finetune(0.004, 0.004, 0.0004, 0.05)
    drv.initialize()
    for _ in range(30):
        drv.iterate()
    new = drv.taus()
    assert new != list(tau) and all(new[parent[p]] > new[p] for p in range(2 * taxa - 2))
    for i in range(nloci):
        t = drv.tree(i)
        assert t["logpr"] == pytest.approx(drv.logpr(i), rel=1e-11)
        for v in range(taxa, 2 * taxa - 1):
            pv = t["pop"][v]
            assert new[pv] <= t["time"][v] and (parent[pv] < 0 or t["time"][v] < new[parent[pv]])
            assert t["time"][v] > max(t["time"][t["left"][v]], t["time"][t["right"][v]])
    drv.close()


def test_joint_prior_of_taus_and_gene_trees():
    """BPP's tau prior (gamma on the root, uniform below) x MSC, no data: the gene trees integrate out, so
    the taus must come out of the sampler with their prior moments — which needs the rubber-band
    Jacobian, the density ratio over all loci and the mixing Jacobian to be right"""
    taxa, theta, nloci, alpha, beta = 4, 0.004, 4, 12.0, 3000.0
    rng = np.random.default_rng(17)
    parent, tau, thetas = synth.species_tree_arrays(taxa, theta)
    start = [synth._msc_gene_tree(synth.SPECIES_TREES[taxa], theta, rng) for _ in range(nloci)]
    data = [dict(seqs=["A"] * taxa, left=l, right=r, times=t, root=rt) for l, r, t, rt in start]
    drv = hostdrv.prior_driver(data, seed=23)
    drv.set_species_tree(parent, tau, thetas)
    drv.set_tau_prior(alpha, beta)
    drv.set_finetune(0.01, 0.01, 0.003, 0.8)
    drv.initialize()
    samples = []
    for it in range(12000):
        drv.iterate()
        if it >= 500:
            samples.append(drv.taus()[taxa:])
    s = np.array(samples)                                   # columns: tau_AB, tau_ABC, tau_root
    root = s[:, 2]
    assert abs(root.mean() - alpha / beta) < 0.04 * alpha / beta, root.mean()
    assert abs(root.std() - np.sqrt(alpha) / beta) < 0.12 * np.sqrt(alpha) / beta, root.std()
    assert abs((s[:, 1] / root).mean() - 2 / 3) < 0.03 and abs((s[:, 0] / root).mean() - 1 / 3) < 0.03
    drv.close()


def test_theta_moves_reproduce_the_theta_prior():
    """no data: the gene trees integrate out of MSC(G | theta, tau), so every theta that can hold a coalescence
    must come out with its gamma prior's moments (the THETA step's density ratio over all loci and the prior
    ratio), jointly with moving taus and gene trees"""
    taxa, nloci, a, b = 4, 3, 8.0, 2000.0
    rng = np.random.default_rng(4)
    parent, tau, thetas = synth.species_tree_arrays(taxa, 0.004)
    start = [synth._msc_gene_tree(synth.SPECIES_TREES[taxa], 0.004, rng) for _ in range(nloci)]
    data = [dict(seqs=["A"] * taxa, left=l, right=r, times=t, root=rt) for l, r, t, rt in start]
    drv = hostdrv.prior_driver(data, seed=31)
    drv.set_species_tree(parent, tau, thetas)
    drv.set_tau_prior(12.0, 3000.0)
    drv.set_theta_prior(a, b, 0.006)
    drv.set_finetune(0.01, 0.01, 0.003, 0.8)
    drv.initialize()
    th = []
    for it in range(16000):
        drv.iterate()
        if it >= 500:
            th.append(drv.thetas())
    th = np.array(th)
    assert (th[:, :taxa] == 0.004).all()                    # one sequence per species: no theta there, never moved
    for p in range(taxa, 2 * taxa - 1):
        assert abs(th[:, p].mean() - a / b) < 0.05 * a / b, (p, th[:, p].mean())
        assert abs(th[:, p].std() - np.sqrt(a) / b) < 0.12 * np.sqrt(a) / b, (p, th[:, p].std())
    drv.close()
