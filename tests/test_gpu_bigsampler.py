"""The big-tree device sampler (csrc/bigsampler.hpp: loci of more than 16 tips, with scalers, or unphased diploids — trees in
HBM, one lane per locus, the host driver's proposal code statement for statement, the engine's general 4-state kernels)
walks the trajectory of the C host driver on libbpp_amd.so with the same seeds: same accept/reject history, same trees,
populations, buffer and scaler indices, taus and thetas.  BASELINE config 1's loci (examples/frogs: unphased diploids,
42-60 tips after phasing) run on it."""
import json
import os

import numpy as np
import pytest

import bpp_amd
from bpp_amd import synth, seqio
import hostdrv
import tape
from common import rel
from test_gpu_gsampler import walk
from test_gpu_host_driver import _msc_start_tree

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("taxa,model,R,nloci,iters,scaling", [(4, "jc69", 1, 150, 5, False), (8, "gtr", 4, 30, 3, False), (8, "jc69", 1, 30, 3, True)])
def test_big_sampler_forced_on_small_loci_equals_host_driver(taxa, model, R, nloci, iters, scaling, monkeypatch):
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(nloci, 300, taxa, model, R, seed=19)
    host = hostdrv.hip_driver(eng, tape.make_engine_loci(eng, data, scaling), data, seed=29, scaling=scaling)
    monkeypatch.setenv("BPA_SMP_BIG", "1")
    dev = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data, scaling), data, seed=29)
    monkeypatch.delenv("BPA_SMP_BIG")
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    for drv in (host, dev):
        drv.set_species_tree(parent, tau0, thetas)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        drv.set_theta_prior(2.0, 1000.0, 0.001)
        drv.set_finetune(0.003, 0.005, 0.0008, 0.2)
    walk(host, dev, iters, nloci)
    assert dev.kind() == "big"
    assert dev.taus() != list(tau0) and dev.thetas() != list(thetas)
    host.close(); dev.close(); eng.close()


def test_big_sampler_on_24_tip_loci_with_scalers():
    """four species with six sequences each (24 tips: beyond the generic sampler's 16), scale buffers on"""
    eng = bpp_amd.Engine(0)
    rng = np.random.default_rng(5)
    parent, tau0, thetas = synth.species_tree_arrays(4, 0.004)
    species = [k // 6 for k in range(24)]
    data = []
    for _ in range(12):
        left, right, times, root = _msc_start_tree(species, parent, tau0, thetas, rng)
        base = "".join(rng.choice(list("ACGT"), 200))
        seqs = ["".join(c if rng.random() > 0.04 else rng.choice(list("ACGT")) for c in base) for _ in range(24)]
        pats, w = bpp_amd.compress_site_patterns(seqs, True, True)
        data.append(dict(seqs=pats, weights=w, left=left, right=right, times=times, root=root, states=4, rate_cats=1, model="jc69", rates=np.ones(1)))
    host = hostdrv.hip_driver(eng, tape.make_engine_loci(eng, data, True), data, seed=3, scaling=True)
    dev = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data, True), data, seed=3)
    for drv in (host, dev):
        drv.set_species_tree(parent, tau0, thetas)
        for i in range(len(data)):
            drv.set_tip_species(i, species)
        drv.set_tau_prior(3.0, 3.0 / tau0[-1])
        drv.set_theta_prior(2.0, 500.0, 0.001)
        drv.set_finetune(0.002, 0.003, 0.0004, 0.1)
    walk(host, dev, 2, len(data))
    assert dev.kind() == "big"
    for i in range(len(data)):
        assert [int(x) for x in dev.tree(i)["pop"]] == [int(x) for x in host.tree(i)["pop"]]
    host.close(); dev.close(); eng.close()


def test_big_sampler_on_the_frogs_loci():
    """BASELINE config 1's data from the files the reference ships (5 loci, unphased diploid sequences phased analytically:
    42-60 tips, the per-pattern terms averaged over the phase resolutions) under the MSC on (((K, C), L), H)"""
    gold = json.load(open(os.path.join(G, "input_pipeline.json")))
    recs = seqio.load_dataset(os.path.join(G, "frogs", "frogs.txt"), os.path.join(G, "frogs", "frogs.Imap.txt"), gold["species"], [1, 1, 1, 1], model="jc69")
    eng = bpp_amd.Engine(0)
    parent = [4, 4, 5, 6, 5, 6, -1]                       # K C L H | KC KCL root
    tau0 = [0.0] * 4 + [0.01, 0.02, 0.03]
    thetas = [0.02] * 7
    rng = np.random.default_rng(9)
    data = []
    for r in recs:
        left, right, times, root = _msc_start_tree(r["species"], parent, tau0, thetas, rng)
        data.append(dict(seqs=r["seqs"], weights=r.get("weights", np.ones(len(r["seqs"][0]))), left=left, right=right, times=times, root=root,
                         states=4, rate_cats=1, model="jc69", rates=np.ones(1)))
    assert max(len(r["seqs"]) for r in recs) > 16
    host = hostdrv.hip_driver(eng, [seqio.make_locus(eng, r) for r in recs], data, seed=8)
    dev = bpp_amd.Sampler(eng, [seqio.make_locus(eng, r) for r in recs], data, seed=8)
    for drv in (host, dev):
        drv.set_species_tree(parent, tau0, thetas)
        for i, r in enumerate(recs):
            drv.set_tip_species(i, r["species"])
        drv.set_tau_prior(3.0, 100.0)
        drv.set_theta_prior(3.0, 150.0, 0.003)
        drv.set_finetune(0.004, 0.004, 0.002, 0.1)
    walk(host, dev, 2, len(data))
    assert dev.kind() == "big" and dev.thetas() != list(thetas)
    host.close(); dev.close(); eng.close()
