"""The generic device-resident sampler (csrc/gsampler.hpp: loci outside the LDS sweep kernel's scope — several rate
categories, GTR, up to 16 tips — proposed on the device as records for the engine's step kernels) walks the trajectory
of the C host driver on libbpp_amd.so with the same seeds, which is the host driver's on the REAL reference
(test_gpu_host_driver.py): same accept/reject history, same trees, populations, buffer indices, taus and thetas."""
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


def walk(host, dev, iters, nloci, check_every=1, tol=1e-12):
    host.initialize(); dev.initialize()
    s = dev.summary()
    assert rel(s["total_lnl"], host.total_lnl()) < 1e-13
    for it in range(iters):
        host.iterate(); dev.iterate(1)
        if it % check_every:
            continue
        s = dev.summary()
        hp, ha, _ = host.counters()
        assert (s["proposals"], s["accepted"]) == (hp, ha), it
        assert rel(s["total_lnl"], host.total_lnl()) < 1e-11, it
    assert np.allclose(dev.taus(), host.taus(), rtol=tol, atol=0)
    assert np.allclose(dev.thetas(), host.thetas(), rtol=tol, atol=0)
    for i in range(nloci):
        a, b = dev.tree(i), host.tree(i)
        assert a["root"] == b["root"]
        for key in ("left", "right", "parent", "clv", "pmat", "pop"):
            assert [int(x) for x in a[key]] == [int(x) for x in b[key]], (i, key)
        assert np.allclose(a["time"], b["time"], rtol=tol, atol=0)
        assert rel(a["lnl"], b["lnl"]) < 10*tol and rel(a["logpr"], b["logpr"]) < 10*tol


@pytest.mark.parametrize("taxa,model,R,nloci,iters,forced,chain", [(4, "jc69", 1, 300, 5, True, None), (8, "gtr", 4, 60, 3, False, "1"), (8, "gtr", 4, 60, 3, False, "0"),
                                                                   (8, "gtr", 4, 700, 2, False, "0"), (8, "gtr", 4, 700, 2, False, "1"), (8, "jc69", 1, 40, 3, True, "0"),
                                                                   (8, "jc69", 1, 40, 3, True, None), (6, "lg", 4, 40, 3, False, None), (6, "lg", 1, 24, 2, False, None),
                                                                   (6, "lg", 4, 300, 2, False, None)])
def test_generic_sampler_equals_host_driver(taxa, model, R, nloci, iters, forced, chain, monkeypatch):
    """(lg: amino-acid loci — BASELINE config 4's kind —, the steps written on the device as the records of the tiled 20-state
    kernels: pmatrix_wg2_kernel, partials_lnl_pipe20_kernel; 700 GTR loci: enough workgroups of the packing for the per-locus
    steps to run as two half-batches on two streams — more launches, the same trajectory; 300 amino-acid loci: the 20-state
    form of the same, two halves of the loci; chain None: the per-locus steps
    as ONE launch (gsm2::gchain_kernel: the default for small JC69 sets, "1": forced), "0": a launch per step (BPA_GS_CHAIN))"""
    if chain is not None:
        monkeypatch.setenv("BPA_GS_CHAIN", chain)
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(nloci, 300, taxa, model, R, seed=19)
    loci_a = tape.make_engine_loci(eng, data)
    loci_b = tape.make_engine_loci(eng, data)
    host = hostdrv.hip_driver(eng, loci_a, data, seed=29)
    if forced:
        os.environ["BPA_SMP_GENERIC"] = "1"            # these loci would fit the sweep kernel
    try:
        dev = bpp_amd.Sampler(eng, loci_b, data, seed=29)
    finally:
        os.environ.pop("BPA_SMP_GENERIC", None)
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    for drv in (host, dev):
        drv.set_species_tree(parent, tau0, thetas)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        drv.set_theta_prior(2.0, 1000.0, 0.001)
        drv.set_finetune(0.003, 0.005, 0.0008, 0.2)
    walk(host, dev, iters, nloci)
    assert dev.taus() != list(tau0) and dev.thetas() != list(thetas)
    # the device state is the state of the explicit-index API: buffers hold what the indices say
    for i in range(0, nloci, max(1, nloci // 8)):
        t = dev.tree(i)
        d = data[i]
        have = loci_b[i].root_loglikelihood(int(t["clv"][t["root"]]), -1)
        ol = O.OracleLocus(d["states"], R, d["seqs"], d["weights"], model=d["model"], freqs=None if model == "jc69" else d["freqs"],
                           qrates=None if model == "jc69" else d["exch"], rates=d["rates"])
        full = ol.full_lnl(list(t["left"]), list(t["right"]), list(t["time"]), t["root"])
        assert rel(have, full) < 1e-12 and rel(t["lnl"], full) < 1e-12
    w = dev.work()
    assert w["sweeps"] >= iters*(3*taxa - 3) and w["node_updates"] > 0 and w["bytes"] > 0      # (generic path: evaluated steps)
    if model == "gtr" and chain == "0":
        assert (w["sweeps"] >= iters*2*(3*taxa - 3)) == (nloci >= 700) == (dev.streams() == 2)   # two half-batch launches per per-locus step
    if model == "lg":
        assert (dev.streams() == 2) == (nloci >= 128)
    if model != "lg":
        assert (dev.summary()["launches"] < iters*2*(3*taxa - 3)) == (chain != "0")             # the chain: one launch for the per-locus steps (else >= 2 each)
    dev.close(); host.close(); eng.close()


def test_generic_sampler_on_the_anopheles_data():
    """BASELINE config 5's data (examples/anopheles: 100 loci x 12 sequences, 6 species with two sequences each, JC69,
    cleandata = 1; priors and step lengths of anopheles-bpp-msci.ctl, the MSC on the control file's tree): 12 tips are
    beyond the sweep kernel — the generic sampler takes them, and walks the host driver's trajectory"""
    from bpp_amd import seqio
    from test_gpu_host_driver import _msc_start_tree
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    species = ["G", "C", "R", "L", "A", "Q"]
    recs = seqio.load_dataset(os.path.join(G, "anopheles", "loci_realign.txt"), os.path.join(G, "anopheles", "Imap.txt"),
                              species, None, model="jc69", cleandata=True)
    parent = [6, 6, 10, 8, 7, 7, 9, 8, 9, 10, -1]
    tau0 = [0.0] * 6 + [0.004, 0.004, 0.008, 0.012, 0.016]
    thetas = [0.02] * 11
    rng = np.random.default_rng(77)
    data = []
    for r in recs:
        left, right, times, root = _msc_start_tree(r["species"], parent, tau0, thetas, rng)
        data.append(dict(seqs=r["seqs"], weights=r["weights"], left=left, right=right, times=times, root=root, states=4,
                         rate_cats=1, model="jc69", rates=np.ones(1)))
    eng = bpp_amd.Engine(0)
    loci_a = tape.make_engine_loci(eng, data)
    loci_b = tape.make_engine_loci(eng, data)
    host = hostdrv.hip_driver(eng, loci_a, data, seed=21)
    dev = bpp_amd.Sampler(eng, loci_b, data, seed=21)
    for drv in (host, dev):
        drv.set_species_tree(parent, tau0, thetas)
        for i, r in enumerate(recs):
            drv.set_tip_species(i, r["species"])
        drv.set_tau_prior(2.0, 10.0)
        drv.set_theta_prior(2.0, 100.0, 0.002)
        drv.set_finetune(0.003, 0.003, 0.00002, 0.9)
    walk(host, dev, 5, len(data))
    assert dev.thetas() != thetas
    s = dev.summary()
    assert 0.1 < s["accepted"] / s["proposals"] < 0.95
    dev.close(); host.close(); eng.close()


@pytest.mark.parametrize("nloci,iters,fuse", [(40, 3, None), (700, 2, None), (700, 2, "BPA_GS_FUSEA=1"), (700, 2, "BPA_GS_FUSEPM=0")])
def test_generic_sampler_with_parameter_moves_equals_host_driver(nloci, iters, fuse, monkeypatch):
    """(fuse None: the proposal's lane groups fill the step's P-matrices themselves, the eigensystems are refreshed by a launch
    of their own; BPA_GS_FUSEA=1: both inside the node-update launch; BPA_GS_FUSEPM=0: the P-matrices as a launch of their own)
    the per-locus frequency / exchangeability / alpha moves (locus.c:2782-3419, prop_gamma.c:52-224) on the device —
    new values into the loci's parameter blocks, eigensystems and category rates refreshed there — against the host
    driver's param_step over libbpp_amd.so's setters: same decisions, same parameters (the category rates of a proposed
    alpha come from device libm here and from glibc there: equal to ~1e-14, not to the bit)"""
    taxa, R = 8, 4
    if fuse is not None:
        __import__("common").skip_unless_experimental(fuse)
        monkeypatch.setenv(*fuse.split("="))
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(nloci, 300, taxa, "gtr", R, seed=23)
    loci_a = tape.make_engine_loci(eng, data)
    loci_b = tape.make_engine_loci(eng, data)
    host = hostdrv.hip_driver(eng, loci_a, data, seed=31)
    dev = bpp_amd.Sampler(eng, loci_b, data, seed=31)
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    for drv in (host, dev):
        drv.set_species_tree(parent, tau0, thetas)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        drv.set_theta_prior(2.0, 1000.0, 0.001)
        drv.set_finetune(0.003, 0.005, 0.0008, 0.2)
        drv.set_subst_moves(0.3, 0.4, 0.8, 1.0, 1.0)
    for i, d in enumerate(data):
        host.set_subst_model(i, list(d["freqs"]), list(d["exch"]), 0.5, R)
        dev.set_subst_model(i, d["freqs"], d["exch"], 0.5)
    walk(host, dev, iters, nloci)
    moved = 0
    for i in range(nloci):
        fh, qh, ah = host.get_subst_model(i)
        fd, qd, ad = dev.get_subst_model(i)
        assert np.allclose(fd, fh, rtol=1e-11, atol=0) and np.allclose(qd, qh, rtol=1e-11, atol=0) and rel(ad, ah) < 1e-11, i
        moved += (ah != 0.5) + (not np.allclose(fh, data[i]["freqs"])) + (not np.allclose(qh, data[i]["exch"]))
    assert moved > nloci
    # the loci's device state is what the explicit-index API would hold for those parameters
    for i in range(0, nloci, 7):
        t = dev.tree(i)
        f, q, a = dev.get_subst_model(i)
        rates = bpp_amd.compute_gamma_cats(a, a, R)
        ol = O.OracleLocus(4, R, data[i]["seqs"], data[i]["weights"], model="gtr", freqs=f, qrates=q, rates=rates)
        full = ol.full_lnl(list(t["left"]), list(t["right"]), list(t["time"]), t["root"])
        assert rel(t["lnl"], full) < 1e-10
    dev.close(); host.close(); eng.close()


def test_window_widths_change_in_mid_run():
    """BPP adjusts its step lengths during the burn-in: switching the substitution-parameter moves on after two
    iterations and changing their widths after two more must not touch the chain's state (trees, streams, counters,
    taus, thetas, parameters) — the device sampler keeps walking the host driver's trajectory.  Also: any locus's
    parameters can be asked for first (not only locus 0's) and are current"""
    taxa, R, nloci = 8, 4, 30
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(nloci, 300, taxa, "gtr", R, seed=41)
    loci_a = tape.make_engine_loci(eng, data)
    loci_b = tape.make_engine_loci(eng, data)
    host = hostdrv.hip_driver(eng, loci_a, data, seed=37)
    dev = bpp_amd.Sampler(eng, loci_b, data, seed=37)
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    for drv in (host, dev):
        drv.set_species_tree(parent, tau0, thetas)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        drv.set_theta_prior(2.0, 1000.0, 0.001)
        drv.set_finetune(0.003, 0.005, 0.0008, 0.2)
    for i, d in enumerate(data):
        host.set_subst_model(i, list(d["freqs"]), list(d["exch"]), 0.5, R)
        dev.set_subst_model(i, d["freqs"], d["exch"], 0.5)
    host.initialize(); dev.initialize()
    for phase, widths in enumerate([None, (0.3, 0.4, 0.8), (0.1, 0.2, 0.3)]):
        if widths:
            host.set_subst_moves(*widths, 1.0, 1.0); dev.set_subst_moves(*widths, 1.0, 1.0)
        for it in range(2):
            host.iterate(); dev.iterate(1)
            s = dev.summary()
            hp, ha, _ = host.counters()
            assert (s["proposals"], s["accepted"]) == (hp, ha), (phase, it)
            assert rel(s["total_lnl"], host.total_lnl()) < 1e-10, (phase, it)
        # a locus other than 0 first
        fh, qh, ah = host.get_subst_model(nloci - 1)
        fd, qd, ad = dev.get_subst_model(nloci - 1)
        assert np.allclose(fd, fh, rtol=1e-11, atol=0) and np.allclose(qd, qh, rtol=1e-11, atol=0) and rel(ad, ah) < 1e-11, phase
    assert np.allclose(dev.taus(), host.taus(), rtol=1e-12, atol=0) and np.allclose(dev.thetas(), host.thetas(), rtol=1e-12, atol=0)
    assert any(host.get_subst_model(i)[2] != 0.5 for i in range(nloci))
    with pytest.raises(bpp_amd.BpaError):
        dev.set_subst_model(0, data[0]["freqs"], data[0]["exch"], 0.7)          # not while the sampler runs
    dev.close(); host.close(); eng.close()


def test_loci_of_several_kinds_in_one_sampler():
    """A data set no single device sampler takes (BPP: a partition list with a model per locus, loci of any size —
    method.c:3320-3346): JC69 loci that fit the LDS kernels, JC69 loci with more than 64 patterns, GTR+Gamma4 loci.  The
    sampler deals them to a part per kind (csrc/composite.hpp: persistent-kernel sweeps + one launch per all-loci step for the
    first, the generic path's two record formats for the others) and steps the parts together; the whole walks the host
    driver's trajectory — every decision, every tree."""
    eng = bpp_amd.Engine(0)
    fit = synth.make_dataset(96, 300, 8, "jc69", 1, seed=5)
    wide = synth.make_dataset(7, 8000, 8, "jc69", 1, seed=6, divergence=20.0)
    gtr = synth.make_dataset(30, 300, 8, "gtr", 4, seed=7)
    assert all(len(d["weights"]) <= 64 for d in fit) and all(64 < len(d["weights"]) < 256 for d in wide), [len(d["weights"]) for d in wide]
    # interleaved: the parts' loci are not contiguous ranges of the whole set
    data = []
    for i in range(96):
        data.append(fit[i])
        if i % 14 == 0: data.append(wide[i // 14])
        if i % 3 == 0 and i // 3 < 30: data.append(gtr[i // 3])
    nloci = len(data)
    loci_a = tape.make_engine_loci(eng, data)
    loci_b = tape.make_engine_loci(eng, data)
    host = hostdrv.hip_driver(eng, loci_a, data, seed=31)
    dev = bpp_amd.Sampler(eng, loci_b, data, seed=31)
    parent, tau0, thetas = synth.species_tree_arrays(8)
    for drv in (host, dev):
        drv.set_species_tree(parent, tau0, thetas)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        drv.set_theta_prior(2.0, 1000.0, 0.001)
        drv.set_finetune(0.003, 0.005, 0.0008, 0.2)
    assert dev.kind() == "composite"
    walk(host, dev, 6, nloci)
    assert dev.taus() != list(tau0) and dev.thetas() != list(thetas)
    w = dev.work()
    assert w["node_updates"] > 0 and w["bytes"] > 0
    dev.close(); host.close(); eng.close()


@pytest.mark.parametrize("taxa,model,R,nloci,iters,subst,env", [(8, "gtr", 4, 60, 4, True, None), (8, "gtr", 4, 700, 2, True, None), (6, "lg", 4, 40, 3, False, None),
                                                                (8, "jc69", 1, 40, 4, False, None),
                                                                # the paths kept next to the defaults: the sums through a device buffer and a copy /
                                                                # waited for with the stream, 20-state P-matrices a workgroup per branch entry
                                                                (8, "gtr", 4, 60, 3, True, "BPA_GS_PINOUT=0"), (8, "gtr", 4, 60, 3, True, "BPA_GS_PINOUT=1"),
                                                                # the all-loci decisions on the HOST (round 5's form, the trajectory reference of
                                                                # the device's gdec_kernel, which is the default since round 6)
                                                                (8, "gtr", 4, 60, 4, True, "BPA_GS_HOSTDEC=1"), (6, "lg", 4, 40, 3, False, "BPA_GS_HOSTDEC=1"),
                                                                (6, "lg", 4, 40, 3, False, "BPA_S20_PMGROUP=0")])
def test_generic_sampler_with_the_program_s_moves_equals_host_driver(taxa, model, R, nloci, iters, subst, env, monkeypatch):
    """BPP's own iteration on the generic sampler (bpa_sampler_set_proposal_kernel(BPP) + bpa_sampler_set_program_moves): the
    per-locus proposals draw from the reference's generator with its Bactrian-Laplace windows and acceptance rule on the device
    (gsm2::gstep2_kernel<.., BPP>), THETA by the metropolized Gibbs draw, the thetas re-drawn inside the rubber band and the
    mixing step — decided ON THE DEVICE by one wave from the loci's sums (gsm::gdec_kernel: the persistent kernel's control-wave
    functions; round 6) or, BPA_GS_HOSTDEC=1, on the host (gs_prog_theta / _tau / _mix, the statements of theta_step_gibbs /
    tau_step / mix_step of a00_driver.c).  Same trajectory as the C host driver with a00_set_program_moves on the same library."""
    if model == "jc69":
        monkeypatch.setenv("BPA_SMP_GENERIC", "1")
    if env:
        __import__("common").skip_unless_experimental(env)
        monkeypatch.setenv(*env.split("="))
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(nloci, 300, taxa, model, R, seed=41)
    loci_a = tape.make_engine_loci(eng, data)
    loci_b = tape.make_engine_loci(eng, data)
    host = hostdrv.hip_driver(eng, loci_a, data, seed=43)
    dev = bpp_amd.Sampler(eng, loci_b, data, seed=43)
    monkeypatch.delenv("BPA_SMP_GENERIC", raising=False)
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    for drv in (host, dev):
        drv.set_proposal_kernel(1)
        drv.set_program_moves(True, 0.3)
        drv.set_species_tree(parent, tau0, thetas)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        drv.set_theta_prior(2.0, 1000.0, 0.0004)
        drv.set_finetune(0.003, 0.005, 0.0004, 0.05)
        if subst:
            drv.set_subst_moves(0.3, 0.4, 0.8, 1.0, 1.0)
    if subst:
        for i, d in enumerate(data):
            host.set_subst_model(i, list(d["freqs"]), list(d["exch"]), 0.5, R)
            dev.set_subst_model(i, d["freqs"], d["exch"], 0.5)
    # (the Bactrian-Laplace windows and the theta re-draws go through libm's log / sqrt / lgamma on both sides: equal to ~1e-13
    #  per draw, not to the bit — the tolerance of the persistent kernel's program-moves test)
    walk(host, dev, iters, nloci, tol=1e-9)
    assert dev.kind() == "generic"
    assert dev.taus() != list(tau0) and dev.thetas() != list(thetas)
    gh, gd = host.gibbs_counters(), dev.gibbs_counters()
    assert tuple(gd) == tuple(gh) and gd[0] > 0, (gd, gh)
    dev.close(); host.close(); eng.close()


def test_what_the_generic_sampler_refuses_with_bpp_s_kernel():
    """BPP's proposal kernel on a generic sampler comes with the program's moves and a theta prior, on one rank: anything else
    fails loudly when the run starts (no silent fall-back to the uniform kernel)"""
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(12, 200, 8, "gtr", 4, seed=5)
    parent, tau0, thetas = synth.species_tree_arrays(8)

    def make(program, theta_prior):
        smp = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=3)
        smp.set_proposal_kernel(1)
        if program:
            smp.set_program_moves(True, 0.1)
        smp.set_species_tree(parent, tau0, thetas)
        smp.set_tau_prior(3.0, 3.0 / tau0[-1])
        if theta_prior:
            smp.set_theta_prior(2.0, 1000.0, 0.001)
        smp.set_finetune(0.003, 0.005, 0.0008, 0.2)
        smp.initialize()
        return smp

    for program, theta_prior in ((False, True), (True, False)):
        smp = make(program, theta_prior)
        with pytest.raises(bpp_amd.BpaError, match="program's moves"):
            smp.iterate(1)
        smp.close()
    smp = make(True, True)
    smp.iterate(2)                                   # the supported combination runs
    assert smp.gibbs_counters()[0] > 0
    smp.close()
    eng.close()
