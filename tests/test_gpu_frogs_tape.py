"""BASELINE config 1 (examples/frogs A00: 5 loci, unphased diploids -> 42-60 tips after phasing, 22-102 patterns,
JC69) end to end through proposals: the files the reference's tests hold -> our reader / phasing -> device loci;
a proposal tape (gene-node ages, prune/regraft with the reference's buffer toggling, rollbacks) replayed step by step
on the GPU and through the REAL reference's locus API (locus_update_matrices / _partials /
locus_root_loglikelihood with its phase-resolution averaging, locus.c:2586-2615) on the reference's own phased
patterns.  Same pattern order on both sides, so the log-likelihoods agree to the last bits."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import bpp_amd
from bpp_amd import seqio
import oraclelib as O
import tape
from common import rand_tree, rel

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref did not travel")]
HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")


def test_frogs_tape_gpu_vs_reference(engine):
    gold = json.load(open(os.path.join(G, "input_pipeline.json")))
    recs = seqio.load_dataset(os.path.join(G, "frogs", "frogs.txt"), os.path.join(G, "frogs", "frogs.Imap.txt"),
                              gold["species"], [1, 1, 1, 1], model="jc69")
    rng = np.random.default_rng(2)
    data, loci = [], []
    for r in recs:
        left, right, times, root = rand_tree(len(r["seqs"]), rng, 0.01)
        data.append(dict(seqs=r["seqs"], weights=np.ones(len(r["weights"]), dtype=np.uint32), left=left, right=right,
                         times=times, root=root, states=4, rate_cats=1, model="jc69", rates=np.ones(1)))
        loci.append(seqio.make_locus(engine, r))
    sch = tape.make_schedule(data, seed=6)
    steps = [sch.initial_step()] + sch.iteration()
    assert len(steps) > 150                          # 59 age + 118 prune/regraft steps for the 60-tip locus alone
    got = []
    for st in steps:
        p = tape.plan_for_step(engine, loci, st)
        p.launch()
        got.append(p.lnl())
        p.close()
    ulp = C.POINTER(C.c_ulong)
    for li, (r, w) in enumerate(zip(recs, gold["frogs_jc69_phased"]["loci"])):
        d = dict(data[li], seqs=w["a3"]["seqs"])                    # the reference's own phased patterns
        rl = tape.ref_locus_for(d)
        rc = np.array(w["resolution_count"], dtype=np.uint64)
        mp = np.array(w["mapping"], dtype=np.uint64)
        uw = np.array(w["a1"]["weights"], dtype=np.uint32)
        rl.L.ref_set_diploid(rl.h, len(rc), rc.ctypes.data_as(ulp), mp.ctypes.data_as(ulp), C.c_ulong(len(mp)),
                             uw.ctypes.data_as(C.POINTER(C.c_uint)))
        sub = tape.locus_subtape(steps, li)
        want, _ = tape.ref_replay(rl, tape.ref_tape_arrays(sub))
        mine = np.array([got[s["step"]][s["task"]] for s in sub])
        assert len(sub) >= 3 * len(r["seqs"]) - 3
        assert np.all(np.isfinite(want)) and np.all(np.abs(mine - want) <= 1e-13 * np.abs(want)), \
            (li, np.max(np.abs(mine - want) / np.abs(want)))
        rl.free()
