"""Device-resident proposal control (bpa_sampler_t) walks the same trajectory as the C host
driver on libbpp_amd.so with the same seed — which in turn equals the host driver on the real
reference (test_gpu_host_driver.py).  Same proposals (integer random streams), same
accept/reject history, same trees; ages and log-likelihoods to ~1e-12 (device exp/log vs glibc
in the root-age and mixing multipliers)."""
import os

import numpy as np
import pytest

import bpp_amd
from bpp_amd import synth
import oraclelib as O
import hostdrv
import tape
from common import rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("taxa,nloci,iters", [(4, 300, 6), (8, 60, 4)])
def test_sampler_equals_host_driver(taxa, nloci, iters):
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(nloci, 400, taxa, "jc69", 1, seed=17)
    loci_a = tape.make_engine_loci(eng, data)
    loci_b = tape.make_engine_loci(eng, data)
    host = hostdrv.hip_driver(eng, loci_a, data, seed=23)
    dev = bpp_amd.Sampler(eng, loci_b, data, seed=23)
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    for drv in (host, dev):
        drv.set_species_tree(parent, tau0, thetas)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        drv.set_theta_prior(2.0, 1000.0, 0.001)
        drv.set_finetune(0.003, 0.005, 0.0008, 0.2)
    host.initialize(); dev.initialize()
    s = dev.summary()
    assert rel(s["total_lnl"], host.total_lnl()) < 1e-13
    for it in range(iters):
        host.iterate(); dev.iterate(1)
        s = dev.summary()
        hp, ha, _ = host.counters()
        assert (s["proposals"], s["accepted"]) == (hp, ha), it
        assert rel(s["total_lnl"], host.total_lnl()) < 1e-11, it
    assert np.allclose(dev.taus(), host.taus(), rtol=1e-12, atol=0) and dev.taus() != list(tau0)
    assert np.allclose(dev.thetas(), host.thetas(), rtol=1e-12, atol=0) and dev.thetas() != list(thetas)
    assert dev.thetas()[:taxa] == list(thetas[:taxa])          # one sequence per species: those thetas never move
    for i in range(nloci):
        a, b = dev.tree(i), host.tree(i)
        assert a["root"] == b["root"]
        for key in ("left", "right", "parent", "clv", "pmat", "pop"):
            assert [int(x) for x in a[key]] == [int(x) for x in b[key]], (i, key)
        assert np.allclose(a["time"], b["time"], rtol=1e-12, atol=0)
        assert rel(a["lnl"], b["lnl"]) < 1e-11 and rel(a["logpr"], b["logpr"]) < 1e-11
    # the device state is the state of the explicit-index API: buffers hold what the indices say
    for i in range(0, nloci, max(1, nloci // 10)):
        t = dev.tree(i)
        d = data[i]
        have = loci_b[i].root_loglikelihood(int(t["clv"][t["root"]]), -1)
        full = O.OracleLocus(4, 1, d["seqs"], d["weights"]).full_lnl(list(t["left"]), list(t["right"]), list(t["time"]), t["root"])
        assert rel(have, full) < 1e-12 and rel(t["lnl"], full) < 1e-12
        assert rel(t["logpr"], host.logpr(i)) < 1e-11
    # 4 launches per iteration instead of 3 tips - 2 host round trips
    assert dev.summary()["launches"] <= 1 + (2 + 2 * (taxa - 1)) * iters + 2 * (iters + 2) + nloci
    host.close(); dev.close(); eng.close()


def _sampler(eng, data, taxa, seed, v1, monkeypatch):
    if v1:
        monkeypatch.setenv("BPA_SMP_V1", "1")
    else:
        monkeypatch.delenv("BPA_SMP_V1", raising=False)
    smp = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=seed)
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    smp.set_species_tree(parent, tau0, thetas)
    smp.set_tau_prior(3.0, 3.0 / tau0[-1])
    smp.set_theta_prior(3.0, 1500.0, 8e-5)
    smp.set_finetune(0.004, 0.004, 4e-5, 0.006)
    smp.initialize()
    return smp


@pytest.mark.parametrize("taxa,nloci,iters", [(4, 3000, 12), (8, 150, 6)])
def test_persistent_kernel_equals_one_launch_per_step(taxa, nloci, iters, monkeypatch):
    """the persistent iteration kernel (csrc/sweep2.hpp: state in LDS for the whole launch, all-loci decisions from
    fixed-point device accumulators over several workgroups) and the one-launch-per-step path (BPA_SMP_V1=1) walk the
    same trajectory: every tree, every tau and theta, bit for bit — also when one call runs several iterations"""
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(nloci, 600, taxa, "jc69", 1, seed=5)
    new = _sampler(eng, data, taxa, 11, False, monkeypatch)
    old = _sampler(eng, data, taxa, 11, True, monkeypatch)
    for it in range(iters):
        n = 1 if it % 3 else 4
        new.iterate(n); old.iterate(n)
        assert new.taus() == old.taus() and new.thetas() == old.thetas(), it
        a, b = new.summary(), old.summary()
        assert (a["proposals"], a["accepted"]) == (b["proposals"], b["accepted"]), it
        assert a["launches"] < b["launches"]
    assert new.taus() != list(synth.species_tree_arrays(taxa)[1])
    for i in range(nloci):
        x, y = new.tree(i), old.tree(i)
        for key in ("left", "right", "parent", "clv", "pmat", "pop", "time"):
            assert list(x[key]) == list(y[key]), (i, key)
        assert x["root"] == y["root"] and x["lnl"] == y["lnl"] and x["logpr"] == y["logpr"], i
    assert new.kind() == "persistent" and old.kind() == "sweep"
    assert new.work()["node_updates"] > old.work()["node_updates"]      # (the persistent kernel counts the all-loci steps' too)
    new.close(); old.close(); eng.close()


def test_sweeps_of_the_persistent_kernel_between_one_launch_all_loci_steps(monkeypatch):
    """several ranks: an all-reduce callback is installed, so the all-loci steps run one launch each (their sums go
    through the callback) while the per-locus sweep is the persistent kernel's — which then applies the pending decision
    of the previous iteration's last step while it loads (rejected: the trees come from the pre-step snapshot).  One rank
    with an identity callback, several iterations per call (no download in between): the one-launch-per-step trajectory"""
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(400, 400, 4, "jc69", 1, seed=3)

    def make(v1):
        if v1:
            monkeypatch.setenv("BPA_SMP_V1", "1")
        else:
            monkeypatch.delenv("BPA_SMP_V1", raising=False)
        smp = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=7)
        smp.set_allreduce(lambda p, n, st: True, None, 0)          # (the sums stay in the sampler's own device memory)
        parent, tau0, thetas = synth.species_tree_arrays(4)
        smp.set_species_tree(parent, tau0, thetas)
        smp.set_tau_prior(3.0, 1000.0); smp.set_theta_prior(2.0, 1000.0, 0.001); smp.set_finetune(0.003, 0.005, 0.0008, 0.2)
        smp.initialize()
        return smp
    hyb, old = make(False), make(True)
    assert hyb.kind() == "hybrid" and old.kind() == "sweep"
    for call in range(8):
        hyb.iterate(3); old.iterate(3)
        a, b = hyb.summary(), old.summary()
        assert (a["proposals"], a["accepted"]) == (b["proposals"], b["accepted"]), call
        assert hyb.taus() == old.taus() and hyb.thetas() == old.thetas(), call
    for i in range(len(data)):
        x, y = hyb.tree(i), old.tree(i)
        for key in ("left", "right", "parent", "clv", "pmat", "pop", "time"):
            assert list(x[key]) == list(y[key]), (i, key)
        assert x["lnl"] == y["lnl"] and x["logpr"] == y["logpr"], i
    hyb.close(); old.close(); eng.close()


@pytest.mark.parametrize("program", [False, True])
def test_several_sequences_per_species(program):
    """two species with three sequences each: tip populations hold coalescences (and a theta that moves),
    gene nodes cross the species boundary both ways; device == host driver, step for step.  program: BPP's kernel with the
    program's moves — the rubber band then re-draws the thetas of the two TIP populations next to the root's"""
    eng = bpp_amd.Engine(0)
    rng = np.random.default_rng(8)
    nloci, tips = 120, 6
    species = [0, 0, 0, 1, 1, 1]
    parent, tau0, thetas = [2, 2, -1], [0.0, 0.0, 0.003], [0.002, 0.003, 0.004]
    data = []
    for _ in range(nloci):
        # ((a1,a2),a3) and ((b1,b2),b3) inside their species or deeper, joined above the divergence
        t = sorted(rng.uniform(0.0002, 0.0028, 4))
        left = [-1] * 6 + [0, 6, 3, 8, 7]
        right = [-1] * 6 + [1, 2, 4, 5, 9]
        times = [0.0] * 6 + [t[0], t[2], t[1], t[3], 0.003 + rng.uniform(0.0005, 0.004)]
        seqs = ["".join(rng.choice(list("ACGT"), 60)) for _ in range(2)]
        seqs = [seqs[0]] * 3 + [seqs[1]] * 3
        seqs = ["".join(c if rng.random() > 0.05 else rng.choice(list("ACGT")) for c in s) for s in seqs]
        pats, w = bpp_amd.compress_site_patterns(seqs, True, True)
        data.append(dict(seqs=pats, weights=w, left=left, right=right, times=times, root=10, states=4, rate_cats=1,
                         model="jc69", rates=np.ones(1)))
    loci_a = tape.make_engine_loci(eng, data)
    loci_b = tape.make_engine_loci(eng, data)
    host = hostdrv.hip_driver(eng, loci_a, data, seed=4)
    dev = bpp_amd.Sampler(eng, loci_b, data, seed=4)
    for drv in (host, dev):
        if program:
            drv.set_proposal_kernel(1)
            drv.set_program_moves(True, 0.5)
        drv.set_species_tree(parent, tau0, thetas)
        for i in range(nloci):
            drv.set_tip_species(i, species)
        drv.set_tau_prior(3.0, 1000.0)
        drv.set_theta_prior(2.0, 700.0, 0.002)
        drv.set_finetune(0.003, 0.004, 0.0008, 0.2)
    host.initialize(); dev.initialize()
    for it in range(8):
        host.iterate(); dev.iterate(1)
        s = dev.summary()
        hp, ha, _ = host.counters()
        assert (s["proposals"], s["accepted"]) == (hp, ha), it
    tol = 1e-10 if program else 1e-12          # (the program's theta re-draws go through libm's log / lgamma on both sides)
    assert np.allclose(dev.thetas(), host.thetas(), rtol=tol, atol=0) and all(a != b for a, b in zip(dev.thetas(), thetas))
    assert np.allclose(dev.taus(), host.taus(), rtol=tol, atol=0)
    moved = 0
    for i in range(nloci):
        a, b = dev.tree(i), host.tree(i)
        for key in ("left", "right", "parent", "clv", "pmat", "pop"):
            assert [int(x) for x in a[key]] == [int(x) for x in b[key]], (i, key)
        assert np.allclose(a["time"], b["time"], rtol=tol, atol=0)
        assert rel(a["logpr"], b["logpr"]) < 10*tol and rel(a["lnl"], b["lnl"]) < 10*tol
        moved += sum(int(x) == 2 for x in a["pop"][6:]) != 1          # more than the root node above the divergence
    assert moved > 0
    host.close(); dev.close(); eng.close()


def test_device_sampler_draws_from_the_msc_prior():
    """usedata = 0 (bpa_engine_set_options: lnL = 0, locus.c:2581): the device sampler's gene trees must
    follow the multispecies coalescent — node-age moments against direct simulation"""
    taxa, theta, nloci = 4, 0.002, 2000
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(nloci, 40, taxa, "jc69", 1, seed=3, theta=theta)
    loci = tape.make_engine_loci(eng, data)
    eng.set_options(usedata=0, bfbeta=1.0)
    dev = bpp_amd.Sampler(eng, loci, data, seed=5)
    dev.set_species_tree(*synth.species_tree_arrays(taxa, theta))
    dev.set_finetune(4 * theta, 4 * theta, 0.0, 0.0)
    dev.initialize()
    dev.iterate(15)
    ages = []
    for _ in range(8):
        dev.iterate(3)
        ages += [sorted(dev.tree(i)["time"][taxa:]) for i in range(nloci)]
    ages = np.array(ages)
    rng = np.random.default_rng(1)
    sim = np.array([sorted(synth._msc_gene_tree(synth.SPECIES_TREES[taxa], theta, rng)[2][taxa:]) for _ in range(40000)])
    for j in range(taxa - 1):
        assert abs(ages[:, j].mean() - sim[:, j].mean()) < 0.04 * sim[:, j].mean(), (j, ages[:, j].mean(), sim[:, j].mean())
        assert abs(ages[:, j].std() - sim[:, j].std()) < 0.08 * sim[:, j].std(), j
    eng.set_options(usedata=1, bfbeta=1.0)
    dev.close(); eng.close()


def test_sampler_rejects_unsupported_loci(engine):
    """GTR+G loci are the generic sampler's (tests/test_gpu_gsampler.py), loci with scale buffers the big-tree sampler's
    (tests/test_gpu_bigsampler.py); amino-acid loci with scalers are refused loudly; a mix of one- and several-category loci —
    refused until round 4 — is dealt to a part per kind (csrc/composite.hpp; trajectory: tests/test_gpu_gsampler.py), and with
    BPA_SMP_NO_COMPOSITE=1 refused as before"""
    data = synth.make_dataset(2, 200, 8, "gtr", 4, seed=1)
    loci = tape.make_engine_loci(engine, data, True)                  # with scale buffers
    smp = bpp_amd.Sampler(engine, loci, data)
    parent, tau0, thetas = synth.species_tree_arrays(8)
    smp.set_species_tree(parent, tau0, thetas)
    smp.initialize()
    assert smp.kind() == "big"
    smp.close()
    aa = synth.make_dataset(2, 100, 6, "lg", 1, seed=3)
    with pytest.raises(bpp_amd.BpaError, match="without scalers"):
        bpp_amd.Sampler(engine, tape.make_engine_loci(engine, aa, True), aa)
    mixed = data[:1] + synth.make_dataset(1, 200, 8, "jc69", 1, seed=2)
    loci = tape.make_engine_loci(engine, mixed)
    smp = bpp_amd.Sampler(engine, loci, mixed)
    smp.set_species_tree(parent, tau0, thetas)
    smp.initialize()
    assert smp.kind() == "composite"
    with pytest.raises(bpp_amd.BpaError, match="several kinds"):
        smp.set_proposal_kernel(1)
    smp.close()
    os.environ["BPA_SMP_NO_COMPOSITE"] = "1"
    try:
        with pytest.raises(bpp_amd.BpaError, match="all be JC69"):
            bpp_amd.Sampler(engine, tape.make_engine_loci(engine, mixed), mixed)
    finally:
        os.environ.pop("BPA_SMP_NO_COMPOSITE", None)


def test_native_rccl_callback(monkeypatch):
    """the several-GPU exchange as native code (include/bpp_amd_rccl.h, libbpp_amd_rccl.so): ncclAllReduce on the
    engine's stream behind bpa_allreduce_fn — no Python inside bpa_sampler_iterate.  A one-rank communicator (two ranks
    cannot share the test box's single GPU under RCCL; the two-process tests use gloo): the sums come back unchanged, so
    the sampler walks the single-GPU trajectory, and every all-loci step went through the collective"""
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(300, 400, 4, "jc69", 1, seed=13)
    x = bpp_amd.RcclExchange(bpp_amd.RcclExchange.unique_id(), 1, 0, 0)

    def make(v1, native):
        if v1:
            monkeypatch.setenv("BPA_SMP_V1", "1")
        else:
            monkeypatch.delenv("BPA_SMP_V1", raising=False)
        smp = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=5)
        if native:
            smp.set_allreduce_native(x, None, 0)
        parent, tau0, thetas = synth.species_tree_arrays(4)
        smp.set_species_tree(parent, tau0, thetas)
        smp.set_tau_prior(3.0, 1000.0); smp.set_theta_prior(2.0, 1000.0, 0.001); smp.set_finetune(0.003, 0.005, 0.0008, 0.2)
        smp.initialize()
        return smp
    nat, ref = make(False, True), make(True, False)
    c0 = x.calls()
    for call in range(4):
        nat.iterate(3); ref.iterate(3)
        a, b = nat.summary(), ref.summary()
        assert (a["proposals"], a["accepted"]) == (b["proposals"], b["accepted"]), call
        assert nat.taus() == ref.taus() and nat.thetas() == ref.thetas(), call
    assert x.calls() - c0 == 12 * (1 + 3 + 1)            # THETA (all populations in one collective), 3 TAU, MIX per iteration
    nat.close(); ref.close(); x.close(); eng.close()


@pytest.mark.parametrize("taxa,nloci,iters,slide_prob", [(4, 400, 8, None), (8, 60, 4, None), (4, 400, 12, 0.1), (8, 60, 8, 0.1), (4, 300, 8, 0.0), (4, 300, 8, 1.0)])
def test_bpp_proposal_kernel_on_the_device(taxa, nloci, iters, slide_prob):
    """bpa_sampler_set_proposal_kernel(BPA_KERNEL_BPP): the reference's own generator (legacy_rndu, random.c:104-122) and
    window (Bactrian-Laplace, random.c:192-238) and its acceptance rule (a number drawn only when lnacc < -1e-10) inside the
    persistent kernel — the trajectory of the host driver with A00_KERNEL_BPP (whose generator and variate are bit-equal
    to the reference's functions: tests/test_bpp_kernel.py), per-locus and global streams alike"""
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(nloci, 400, taxa, "jc69", 1, seed=21)
    host = hostdrv.hip_driver(eng, tape.make_engine_loci(eng, data), data, seed=31)
    dev = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=31)
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    for drv in (host, dev):
        drv.set_proposal_kernel(1)
        # slide_prob not None: the program's moves — THETA by the sliding window with that probability, else the metropolized
        # Gibbs draw (stree.c:3957, 3645); thetas re-drawn inside TAU (stree.c:5840) and MIX (prop_mixing.c:272) —, decided in
        # the kernel from the coalescence counts and the T2h sums of all loci
        drv.set_program_moves(slide_prob is not None, slide_prob or 0.0)     # THETA / TAU / MIX as the program runs them
        drv.set_species_tree(parent, tau0, thetas)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        drv.set_theta_prior(2.0, 1000.0, 0.0003)
        drv.set_finetune(0.0012, 0.0015, 0.0002, 0.05)
    host.initialize(); dev.initialize()
    assert dev.kind() == "persistent"
    for it in range(iters):
        host.iterate(); dev.iterate(1)
        s = dev.summary()
        hp, ha, _ = host.counters()
        assert (s["proposals"], s["accepted"]) == (hp, ha), it
        assert rel(s["total_lnl"], host.total_lnl()) < 1e-10, it
    assert np.allclose(dev.taus(), host.taus(), rtol=1e-11, atol=0) and dev.taus() != list(tau0)
    assert np.allclose(dev.thetas(), host.thetas(), rtol=1e-11, atol=0)
    for i in range(nloci):
        a, b = dev.tree(i), host.tree(i)
        for key in ("left", "right", "parent", "clv", "pmat", "pop"):
            assert [int(x) for x in a[key]] == [int(x) for x in b[key]], (i, key)
        assert np.allclose(a["time"], b["time"], rtol=1e-11, atol=0)
    # several iterations per call: the global stream's state stays on the device between launches
    host.iterate(); host.iterate(); host.iterate(); dev.iterate(3)
    assert np.allclose(dev.taus(), host.taus(), rtol=1e-11, atol=0) and np.allclose(dev.thetas(), host.thetas(), rtol=1e-11, atol=0)
    hp, ha, _ = host.counters(); s = dev.summary()
    assert (s["proposals"], s["accepted"]) == (hp, ha)
    assert dev.gibbs_counters() == host.gibbs_counters()
    gp, ga = dev.gibbs_counters()
    ntheta = taxa - 1                                        # one sequence per species: only the inner populations hold coalescences
    if slide_prob == 0.0:
        assert gp == (iters + 3) * ntheta and ga > 0
    elif slide_prob is not None and slide_prob < 1:
        assert 0 < gp < (iters + 3) * ntheta and ga > 0
    else:
        assert gp == 0
    host.close(); dev.close(); eng.close()


@pytest.mark.parametrize("taxa", [4, 6, 8])
@pytest.mark.parametrize("program", [False, True])
def test_persistent_kernel_at_edge_sizes(taxa, program):
    """1, 2, 13, 64, 65 loci (one lane group, one partial wave, a workgroup boundary) on the persistent kernel, several
    iterations per launch, with our kernel and with the program's moves: the host driver's decisions, trees, taus and thetas"""
    for nloci in (1, 2, 13, 64, 65):
        eng = bpp_amd.Engine(0)
        data = synth.make_dataset(nloci, 300, taxa, "jc69", 1, seed=100 + nloci)
        host = hostdrv.hip_driver(eng, tape.make_engine_loci(eng, data), data, seed=5)
        dev = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=5)
        parent, tau0, thetas = synth.species_tree_arrays(taxa)
        for drv in (host, dev):
            if program:
                drv.set_proposal_kernel(1)
                drv.set_program_moves(True, 0.3)
            drv.set_species_tree(parent, tau0, thetas)
            drv.set_tau_prior(3.0, 3.0 / tau0[-1])
            drv.set_theta_prior(2.0, 1000.0, 0.0004)
            drv.set_finetune(0.003, 0.004, 0.0004, 0.1)
        host.initialize(); dev.initialize()
        assert dev.kind() == "persistent"
        for chunk in (1, 4, 7):
            for _ in range(chunk):
                host.iterate()
            dev.iterate(chunk)
            s = dev.summary(); hp, ha, _ = host.counters()
            assert (s["proposals"], s["accepted"]) == (hp, ha), (nloci, chunk)
        assert np.allclose(dev.taus(), host.taus(), rtol=1e-10, atol=0) and np.allclose(dev.thetas(), host.thetas(), rtol=1e-10, atol=0), nloci
        for i in range(nloci):
            a, b = dev.tree(i), host.tree(i)
            assert [int(x) for x in a["parent"]] == [int(x) for x in b["parent"]] and np.allclose(a["time"], b["time"], rtol=1e-10, atol=0), (nloci, i)
        host.close(); dev.close(); eng.close()


@pytest.mark.parametrize("moves", ["uniform", "bpp", "program"])
@pytest.mark.parametrize("inject", ["2", "3,w"])
def test_a_launch_that_gives_up_is_run_again_on_the_same_random_numbers(moves, inject, monkeypatch):
    """A persistent launch whose workgroups are not all resident together (a shared device) gives up and leaves the loci as it
    found them; the launches queued behind it give up at once and the iterations run again, in order, from the state and the
    global stream the first of them started from: the chain is the one a run without the time-out walks, to the bit.
    BPA_SMP_INJECT=k makes the k-th launch give up at its first wait (",w": workgroup 0 alone, the others then meet the real
    0.5 s time-out at the next exchange and workgroup 0's copy of the decision is thrown away with theirs)."""
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(1100, 400, 4, "jc69", 1, seed=29)
    parent, tau0, thetas = synth.species_tree_arrays(4)

    def run(env):
        if env:
            monkeypatch.setenv("BPA_SMP_INJECT", env)
        else:
            monkeypatch.delenv("BPA_SMP_INJECT", raising=False)
        smp = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=31)
        if moves != "uniform":
            smp.set_proposal_kernel(1)
        if moves == "program":
            smp.set_program_moves(True, 0.1)
        smp.set_species_tree(parent, tau0, thetas)
        smp.set_tau_prior(3.0, 3.0 / tau0[-1])
        smp.set_theta_prior(2.0, 1000.0, 0.001)
        smp.set_finetune(0.003, 0.005, 0.0008, 0.2)
        smp.initialize()
        assert smp.kind() == "persistent"
        for n in (2, 3, 1, 4):                 # four launches queued before anything is read back
            smp.iterate(n)
        sm = smp.summary()
        out = (sm, smp.taus(), smp.thetas(), [smp.tree(i) for i in range(0, len(data), 37)])
        smp.close()
        return out

    want, got = run(None), run(inject)
    assert got[0]["proposals"] == want[0]["proposals"] and got[0]["accepted"] == want[0]["accepted"]
    assert got[0]["total_lnl"] == want[0]["total_lnl"]
    assert got[1] == want[1] and got[2] == want[2] and got[1] != list(tau0)
    for a, b in zip(got[3], want[3]):
        for key in ("left", "right", "parent", "clv", "pmat", "pop", "time"):
            assert list(a[key]) == list(b[key]), key
        assert a["lnl"] == b["lnl"] and a["root"] == b["root"]
    eng.close()
