"""The C host driver (host MCMC control in C, include/bpp_amd_host.h) on the REAL
reference's locus API: it keeps trees valid, its incremental log-likelihoods equal a
from-scratch evaluation (the reference's check_logl invariant, method.c:4699-4717), and it
accepts/rejects.  GPU twin: tests/test_gpu_host_driver.py."""
import numpy as np
import pytest

from bpp_amd import synth
import oraclelib as O
import hostdrv
from common import rel

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")


@pytest.mark.parametrize("taxa,model,R,scaling", [(4, "jc69", 1, False), (8, "gtr", 4, False), (6, "jc69", 2, True)])
def test_driver_on_reference_backend(taxa, model, R, scaling):
    data = synth.make_dataset(10, 300, taxa, model, R, seed=31, theta=0.004 if taxa == 6 else None)
    drv = hostdrv.reference_driver(data, seed=7, scaling=scaling)
    theta = 0.004 if taxa == 6 else 0.002
    parent, tau0, thetas = synth.species_tree_arrays(taxa, theta)
    drv.set_species_tree(parent, tau0, thetas)
    if taxa == 6:
        drv.set_finetune(0.02, 0.02, 0.005, 0.3)
    drv.initialize()
    l0 = drv.total_lnl()
    want0 = sum(O.OracleLocus(d["states"], R, d["seqs"], d["weights"], model=model,
                              freqs=None if model == "jc69" else d["freqs"],
                              qrates=None if model == "jc69" else d["exch"], rates=d["rates"],
                              scaling=scaling).full_lnl(d["left"], d["right"], d["times"], d["root"]) for d in data)
    assert rel(l0, want0) < 1e-13
    for _ in range(4):
        drv.iterate()
    props, acc, steps = drv.counters()
    assert 1 + 4 * ((taxa - 1) + 1) <= steps <= 1 + 4 * ((taxa - 1) + (2 * taxa - 2) + 1 + (taxa - 1))
    new = drv.taus()
    assert len(new) == 2 * taxa - 1 and new != list(tau0) and new[:taxa] == [0.0] * taxa
    assert all(new[parent[p]] > new[p] for p in range(2 * taxa - 2))
    assert 0.05 < acc / props < 0.98
    total = 0.0
    for i, d in enumerate(data):
        t = drv.tree(i)
        # valid tree, root object still the root, buffers in their own pairs
        assert t["root"] == 2 * taxa - 2 and t["parent"][t["root"]] == -1
        for v in range(taxa, 2 * taxa - 1):
            assert t["time"][v] > max(t["time"][t["left"][v]], t["time"][t["right"][v]])
            assert t["clv"][v] in (v, v + taxa - 1)
        ol = O.OracleLocus(d["states"], R, d["seqs"], d["weights"], model=model,
                           freqs=None if model == "jc69" else d["freqs"],
                           qrates=None if model == "jc69" else d["exch"], rates=d["rates"], scaling=scaling)
        full = ol.full_lnl(t["left"], t["right"], t["time"], t["root"])
        assert rel(t["lnl"], full) < 1e-12
        # every node sits in the population its age and descendants put it in; the carried MSC density
        # is the from-scratch one (check_logpr, method.c:4719)
        for v in range(taxa, 2 * taxa - 1):
            pv = t["pop"][v]
            assert new[pv] <= t["time"][v] and (parent[pv] < 0 or t["time"][v] < new[parent[pv]])
        assert t["pop"][:taxa] == list(range(taxa))
        assert rel(t["logpr"], drv.logpr(i)) < 1e-12 and np.isfinite(t["logpr"])
        total += t["lnl"]
    assert rel(drv.total_lnl(), total) < 1e-13
    assert drv.total_lnl() > l0 - 50        # a likelihood-driven sampler does not run away downhill
    drv.close()


@pytest.mark.parametrize("threads", [2, 5])
def test_thread_count_does_not_change_the_trajectory(threads):
    """the per-locus loops of a step run on worker threads (a00_set_threads): every draw of a per-locus proposal comes
    from that locus's own stream and the sums of the all-loci steps are taken in locus order afterwards, so a
    threaded run walks the one-thread trajectory exactly — decisions, trees, populations, densities, taus, thetas"""
    data = synth.make_dataset(37, 200, 6, "jc69", 1, seed=5)
    runs = []
    for n in (1, threads):
        drv = hostdrv.reference_driver(data, seed=11)
        drv.set_threads(n)
        parent, tau0, thetas = synth.species_tree_arrays(6)
        drv.set_species_tree(parent, tau0, thetas)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        drv.set_theta_prior(2.0, 1000.0, 0.0004)
        drv.initialize()
        for _ in range(6):
            drv.iterate()
        runs.append((drv.counters(), drv.taus(), drv.thetas(), [drv.tree(i) for i in range(len(data))], drv.total_lnl()))
    a, b = runs
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[4] == b[4]
    for x, y in zip(a[3], b[3]):
        assert x == y


@pytest.mark.parametrize("split,threads", [(1, 1), (18, 3), (36, 2)])
def test_cohorts_do_not_change_the_trajectory(split, threads):
    """a00_set_cohorts: the per-locus steps of the two cohorts of loci are proposed, evaluated and decided in turns (on
    the GPU: one cohort's launch runs while the other is proposed for — tests/test_gpu_host_driver.py), all-loci steps
    are evaluated in two shares: the one-cohort trajectory exactly, substitution-parameter moves included"""
    data = synth.make_dataset(37, 200, 6, "gtr", 2, seed=5)
    runs = []
    for co in (False, True):
        drv = hostdrv.reference_driver_cohorts(data, split, seed=11) if co else hostdrv.reference_driver(data, seed=11)
        drv.set_threads(threads if co else 1)
        parent, tau0, thetas = synth.species_tree_arrays(6)
        drv.set_species_tree(parent, tau0, thetas)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        drv.set_theta_prior(2.0, 1000.0, 0.0004)
        drv.set_subst_moves(0.3, 0.4, 0.8, 1.0, 1.0)
        for i, d in enumerate(data):
            drv.set_subst_model(i, list(d["freqs"]), list(d["exch"]), 0.5, 2)
        drv.initialize()
        for _ in range(5):
            drv.iterate()
        runs.append((drv.counters()[:2], drv.taus(), drv.thetas(), [drv.tree(i) for i in range(len(data))], drv.total_lnl(),
                     [tuple(map(tuple, map(np.atleast_1d, drv.get_subst_model(i)))) for i in range(len(data))]))
        drv.close()
    a, b = runs
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[4] == b[4] and a[5] == b[5]
    for x, y in zip(a[3], b[3]):
        assert x == y


def _batch_se(x, nb=40):
    m = len(x) // nb
    bm = np.array([x[i * m:(i + 1) * m].mean() for i in range(nb)])
    return bm.std(ddof=1) / np.sqrt(nb)


@pytest.mark.parametrize("mode", ["uniform", "bpp", "program"])
def test_prior_only_run_leaves_the_priors_of_theta_and_tau(mode):
    """lnL = 0 (a00_backend_prior) with ALL moves on: the joint is p(taus) p(thetas) p(G | taus, thetas), so the marginal
    of every theta is its gamma prior and that of the root tau its gamma prior, exactly — the THETA windows / Gibbs
    draws, the rubber band with its Jacobian (and the program's theta re-draws), the mixing step all enter (the device
    samplers' twin: tests/test_gpu_prior.py)"""
    taxa = 4
    data = synth.make_dataset(3, 60, taxa, "jc69", 1, seed=3)
    drv = hostdrv.prior_driver(data, seed=17)
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    a_th, b_th, a_tau, b_tau = 3.0, 3.0 / 0.002, 4.0, 4.0 / tau0[-1]
    drv.set_species_tree(parent, tau0, thetas)
    drv.set_tau_prior(a_tau, b_tau)
    drv.set_theta_prior(a_th, b_th, 0.002)
    drv.set_finetune(0.004, 0.004, 0.5 * tau0[-1], 0.6)
    if mode != "uniform":
        drv.set_proposal_kernel(1)
    if mode == "program":
        drv.set_program_moves(True, 0.1)
    drv.initialize()
    for _ in range(2000):
        drv.iterate()
    S = []
    for _ in range(4000):
        for _ in range(4):
            drv.iterate()
        S.append(drv.thetas() + drv.taus())
    S = np.array(S)
    npop = len(parent)
    for p in range(taxa, npop):
        x = S[:, p]
        assert abs(x.mean() - a_th / b_th) < 4.5 * _batch_se(x), (p, x.mean(), _batch_se(x))
        assert 0.8 < x.std() / (np.sqrt(a_th) / b_th) < 1.2, (p, x.std())
    x = S[:, 2 * npop - 1]
    assert abs(x.mean() - a_tau / b_tau) < 4.5 * _batch_se(x), (x.mean(), _batch_se(x))
    assert 0.85 < x.std() / (np.sqrt(a_tau) / b_tau) < 1.15
    drv.close()
