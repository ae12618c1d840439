"""usedata = 0 (lnL = 0 for every locus: bpa_engine_set_options, locus.c:2581): what the device samplers leave is the
multispecies coalescent itself, so the gene trees they visit must look like direct MSC simulation — node-age moments and
how many coalescences fall inside each species.  With SEVERAL sequences per species the tip populations hold
coalescences, prunings regraft across the divergences in both directions and the rubber band moves nodes of tip
populations: bounds, population bookkeeping, Hastings ratios and Jacobians of every move are in this check, on each of the
device implementations (persistent kernel: 6 tips; generic: 12 tips and 16 tips on an 8-species tree; big-tree: 24 tips)."""
import numpy as np
import pytest

import bpp_amd
from bpp_amd import synth
import tape

pytestmark = pytest.mark.gpu

CASES = {
    # species tree (stree->nodes order), sequences per species
    "persistent-2x3": dict(parent=[2, 2, -1], tau=[0, 0, 0.003], per=3, kind="persistent", nloci=1600, burn=40, snaps=14),
    "generic-3x4": dict(parent=[3, 3, 4, 4, -1], tau=[0, 0, 0, 0.002, 0.004], per=4, kind="generic", nloci=1200, burn=40, snaps=12),
    "generic-8x2": dict(parent=[8, 8, 9, 9, 11, 11, 12, 12, 10, 10, 14, 13, 13, 14, -1],
                        tau=[0]*8 + [0.001, 0.0012, 0.0022, 0.0009, 0.0011, 0.002, 0.004], per=2, kind="generic", nloci=900, burn=40, snaps=10),
    "persistent-4x2": dict(parent=[4, 4, 5, 6, 5, 6, -1], tau=[0, 0, 0, 0, 0.0015, 0.003, 0.0045], per=2, kind="persistent", nloci=1200, burn=40, snaps=12),
    "persistent-2x4": dict(parent=[2, 2, -1], tau=[0, 0, 0.003], per=4, kind="persistent", nloci=1200, burn=40, snaps=12),
    "big-4x6": dict(parent=[4, 4, 5, 6, 5, 6, -1], tau=[0, 0, 0, 0, 0.0015, 0.003, 0.0045], per=6, kind="big", nloci=500, burn=30, snaps=10),
}


def stats(trees, tips, species, nsp):
    ages = np.array([sorted(t[taxa_slice(tips)]) for t, _ in trees])
    inside = np.array([[sum(1 for k in range(tips, 2 * tips - 1) if p[k] == s) for s in range(nsp)] for _, p in trees])
    return ages, inside


def taxa_slice(tips):
    return slice(tips, 2 * tips - 1)


@pytest.mark.parametrize("name", list(CASES))
def test_device_samplers_draw_gene_trees_from_the_msc(name):
    c = CASES[name]
    theta = 0.002
    nsp = (len(c["parent"]) + 1) // 2
    tips = nsp * c["per"]
    species = [k // c["per"] for k in range(tips)]
    thetas = [theta] * len(c["parent"])
    rng = np.random.default_rng(12)
    data = []
    for _ in range(c["nloci"]):
        left, right, times, root = synth.msc_start_tree(species, c["parent"], c["tau"], thetas, rng)
        seqs = ["ACGT"] * tips
        pats, w = bpp_amd.compress_site_patterns(seqs, True, True)
        data.append(dict(seqs=pats, weights=w, left=left, right=right, times=times, root=root, states=4, rate_cats=1, model="jc69", rates=np.ones(1)))
    eng = bpp_amd.Engine(0)
    eng.set_options(usedata=0, bfbeta=1.0)
    dev = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=5)
    dev.set_species_tree(c["parent"], c["tau"], thetas)
    for i in range(c["nloci"]):
        dev.set_tip_species(i, species)
    dev.set_finetune(3 * theta, 3 * theta, 0.0, 0.0)              # gene-tree moves only: taus and thetas stay
    dev.initialize()
    assert dev.kind() == c["kind"]
    dev.iterate(c["burn"])
    got = []
    for _ in range(c["snaps"]):
        dev.iterate(4)
        for i in range(c["nloci"]):
            t = dev.tree(i)
            got.append((list(t["time"]), [int(x) for x in t["pop"]]))
    sm = dev.summary()
    assert 0.1 < sm["accepted"] / sm["proposals"] < 0.95
    sim = []
    for _ in range(6000):
        left, right, times, root = synth.msc_start_tree(species, c["parent"], c["tau"], thetas, rng)
        # populations of the simulated tree's inner nodes: the youngest population that has started by the node's age on the
        # path up from its descendants' species (what the samplers keep in `pop`)
        pop = list(species) + [0] * (tips - 1)
        for k in range(tips, 2 * tips - 1):
            a, b = pop[left[k]], pop[right[k]]
            anc = set()
            x = a
            while x >= 0:
                anc.add(x); x = c["parent"][x]
            x = b
            while x not in anc:
                x = c["parent"][x]
            while c["parent"][x] >= 0 and c["tau"][c["parent"][x]] <= times[k]:
                x = c["parent"][x]
            pop[k] = x
        sim.append((times, pop))
    ga, gi = stats(got, tips, species, nsp)
    sa, si = stats(sim, tips, species, nsp)
    for j in range(tips - 1):
        m, s_ = sa[:, j].mean(), sa[:, j].std()
        assert abs(ga[:, j].mean() - m) < 0.05 * m + 3 * s_ / np.sqrt(len(sa)), (name, j, ga[:, j].mean(), m)
        assert abs(ga[:, j].std() - s_) < 0.12 * s_ + 1e-12, (name, j, ga[:, j].std(), s_)
    # coalescences inside each species: mean count and the chance of none
    for s in range(nsp):
        assert abs(gi[:, s].mean() - si[:, s].mean()) < 0.06 * max(si[:, s].mean(), 0.2), (name, s, gi[:, s].mean(), si[:, s].mean())
        assert abs((gi[:, s] == 0).mean() - (si[:, s] == 0).mean()) < 0.03, (name, s)
    eng.set_options(usedata=1, bfbeta=1.0)
    dev.close(); eng.close()


def _batch_se(x, nb=40):
    m = len(x) // nb
    bm = np.array([x[i * m:(i + 1) * m].mean() for i in range(nb)])
    return bm.std(ddof=1) / np.sqrt(nb)


@pytest.mark.parametrize("taxa,model,R,mode,samples,thin,kind", [(4, "jc69", 1, "uniform", 4000, 4, "persistent"),
                                                                 (4, "jc69", 1, "program", 4000, 4, "persistent"),
                                                                 (8, "gtr", 4, "uniform", 1100, 2, "generic")])
def test_all_loci_moves_leave_the_priors_of_theta_and_tau(taxa, model, R, mode, samples, thin, kind):
    """usedata = 0 with ALL moves on: the joint is p(taus) p(thetas) p(G | taus, thetas), so the marginal of every theta is
    its gamma prior and that of the root tau its gamma prior — exactly, whatever the loci.  The THETA windows / Gibbs
    draws, the rubber band with its Jacobian (and the program's theta re-draws inside it), the mixing step: a wrong factor
    in any of them moves these means (tools/prior_marginals.py prints the table)."""
    data = synth.make_dataset(3, 100, taxa, model, R, seed=3)
    eng = bpp_amd.Engine(0)
    eng.set_options(usedata=0, bfbeta=1.0)
    dev = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=17)
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    a_th, b_th = 3.0, 3.0 / 0.002
    a_tau, b_tau = 4.0, 4.0 / tau0[-1]
    dev.set_species_tree(parent, tau0, thetas)
    dev.set_tau_prior(a_tau, b_tau)
    dev.set_theta_prior(a_th, b_th, 0.002)
    dev.set_finetune(0.004, 0.004, 0.5 * tau0[-1], 0.6)
    if mode == "program":
        dev.set_proposal_kernel(1)
        dev.set_program_moves(True, 0.1)
    dev.initialize()
    assert dev.kind() == kind
    dev.iterate(2000)
    S = []
    for _ in range(samples):
        dev.iterate(thin)
        S.append(dev.thetas() + dev.taus())
    S = np.array(S)
    npop = len(parent)
    for p in range(taxa, npop):                                  # the inner populations' thetas (one sequence per species)
        x = S[:, p]
        assert abs(x.mean() - a_th / b_th) < 4.5 * _batch_se(x), (p, x.mean(), _batch_se(x))
        assert 0.8 < x.std() / (np.sqrt(a_th) / b_th) < 1.2, (p, x.std())
    x = S[:, 2 * npop - 1]
    assert abs(x.mean() - a_tau / b_tau) < 4.5 * _batch_se(x), (x.mean(), _batch_se(x))
    assert 0.85 < x.std() / (np.sqrt(a_tau) / b_tau) < 1.15
    if mode == "program":
        g = dev.gibbs_counters()
        assert g[0] > samples and g[1] > 0.5 * g[0]          # (three loci: the inverse-gamma fit is rougher than with thousands)
    eng.set_options(usedata=1, bfbeta=1.0)
    dev.close(); eng.close()


@pytest.mark.parametrize("name,samples,thin", [("big-4x6", 700, 2), ("generic-3x4", 1000, 2), ("persistent-4x2+program", 3000, 3), ("persistent-2x4+program", 3000, 3)])
def test_all_loci_moves_leave_the_priors_with_several_sequences_per_species(name, samples, thin):
    """the same with several sequences per species (the tip populations have thetas too, the rubber band moves nodes inside
    them) — on the big-tree sampler (24 tips), on the generic one (12 tips), and on the persistent kernel with 8 tips (4 species
    x 2, 2 species x 4) under BPP's move kernel and the program's moves: Gibbs thetas of tip populations, re-draws inside the
    rubber band and the mixing step"""
    program = name.endswith("+program")
    name = name.split("+")[0]
    c = CASES[name]
    nsp = (len(c["parent"]) + 1) // 2
    tips = nsp * c["per"]
    species = [k // c["per"] for k in range(tips)]
    npop = len(c["parent"])
    a_th, b_th = 3.0, 3.0 / 0.002
    a_tau, b_tau = 4.0, 4.0 / c["tau"][-1]
    thetas = [a_th / b_th] * npop
    rng = np.random.default_rng(5)
    data = []
    for _ in range(2):
        left, right, times, root = synth.msc_start_tree(species, c["parent"], c["tau"], thetas, rng)
        pats, w = bpp_amd.compress_site_patterns(["ACGT"] * tips, True, True)
        data.append(dict(seqs=pats, weights=w, left=left, right=right, times=times, root=root, states=4, rate_cats=1, model="jc69", rates=np.ones(1)))
    eng = bpp_amd.Engine(0)
    eng.set_options(usedata=0, bfbeta=1.0)
    dev = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=23)
    dev.set_species_tree(c["parent"], c["tau"], thetas)
    for i in range(len(data)):
        dev.set_tip_species(i, species)
    dev.set_tau_prior(a_tau, b_tau)
    dev.set_theta_prior(a_th, b_th, 0.002)
    dev.set_finetune(0.004, 0.004, 0.4 * c["tau"][-1], 0.5)
    if program:
        dev.set_proposal_kernel(1)
        dev.set_program_moves(True, 0.1)
    dev.initialize()
    assert dev.kind() == c["kind"]
    dev.iterate(600)
    S = []
    for _ in range(samples):
        dev.iterate(thin)
        S.append(dev.thetas() + dev.taus())
    S = np.array(S)
    for p in range(npop):
        x = S[:, p]
        assert abs(x.mean() - a_th / b_th) < 4.5 * _batch_se(x), (name, p, x.mean(), _batch_se(x))
        assert 0.75 < x.std() / (np.sqrt(a_th) / b_th) < 1.25, (name, p, x.std())
    x = S[:, 2 * npop - 1]
    assert abs(x.mean() - a_tau / b_tau) < 4.5 * _batch_se(x), (name, x.mean(), _batch_se(x))
    assert 0.8 < x.std() / (np.sqrt(a_tau) / b_tau) < 1.2
    eng.set_options(usedata=1, bfbeta=1.0)
    dev.close(); eng.close()
