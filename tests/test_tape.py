"""The A00 proposal tape (bpp_amd/schedule.py): its workload matches the reference's
measured per-iteration work, it keeps every gene tree valid, and replaying
it with the oracle equals replaying it through the REAL reference's update API
(locus_update_matrices / locus_update_partials / locus_root_loglikelihood with the
reference's own index toggling) bit for bit.  GPU replay: test_gpu_tape.py."""
import numpy as np
import pytest

from bpp_amd import synth
import oraclelib as O
import tape


def build(taxa, model, R, nloci, iters, seed=3, scaling=False):
    data = synth.make_dataset(nloci, 300, taxa, model, R, seed=seed)
    sch = tape.make_schedule(data, seed=5, scaling=scaling,
                             taus=(0.001, 0.002, 0.003) if taxa == 4 else (0.0011, 0.0025, 0.005))
    steps = [sch.initial_step()]
    for _ in range(iters):
        steps += sch.iteration()
    return data, sch, steps


def test_workload_shape_matches_reference_profile():
    data, sch, steps = build(4, "jc69", 1, 40, 4)
    nodes = sum(len(s.ops) for s in steps[1:]) / 40 / 4
    lnls = sum(len(s.loci) for s in steps[1:]) / 40 / 4
    # reference gprof on config 2: 27.9 node updates, 12.7 lnL evaluations per locus-iteration
    assert 24 < nodes < 33 and 11.5 < lnls <= 13
    kinds = [s.kind for s in steps[1:14]]
    assert kinds == ["GAGE"] * 3 + ["GSPR"] * 6 + ["TAU"] * 3 + ["MIX"]


def test_trees_stay_valid():
    data, sch, steps = build(8, "jc69", 1, 10, 5)
    for tr in sch.trees:
        inner = tr.inner_nodes()
        assert len(inner) == tr.inner
        seen = set()
        for v in inner:
            for c in (tr.left[v], tr.right[v]):
                assert tr.parent[c] == v and tr.time[v] > tr.time[c]
                seen.add(c)
        assert tr.parent[tr.root] == -1 and len(seen) == tr.n - 1
        # double buffers: every inner node sits in one of its two CLV / P-matrix slots, no two
        # branches share a P-matrix buffer, and the root object is still the root (gtree.c:6129-6175)
        assert tr.root == tr.n - 1
        for v in range(tr.tips, tr.n):
            assert tr.clv[v] in (v, v + tr.inner)
        pm = [tr.pmat[b] for b in range(tr.n) if tr.parent[b] >= 0]
        assert len(set(pm)) == len(pm) and max(pm) < 2 * tr.edges
        assert all(tr.pmat[b] in (b, b + tr.edges) for b in range(tr.n - 1))


def test_oracle_c_loop_equals_python_replay():
    data, sch, steps = build(4, "jc69", 1, 4, 2)
    for li in range(4):
        sub = tape.locus_subtape(steps, li)
        a = tape.oracle_replay(data[li], sub)
        b, _ = tape.oracle_tape_run(data[li], sub)
        assert (a == b).all()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("taxa,model,R,scaling", [(4, "jc69", 1, False), (8, "gtr", 4, False), (6, "jc69", 2, True)])
def test_tape_oracle_equals_reference(taxa, model, R, scaling):
    if taxa == 6:
        synth.SPECIES_TREES.setdefault(6, synth.SPECIES_TREES[6])
    data = synth.make_dataset(6, 250, taxa, model, R, seed=7, theta=0.002 if taxa != 6 else 0.004)
    sch = tape.make_schedule(data, seed=9, scaling=scaling)
    steps = [sch.initial_step()]
    for _ in range(3):
        steps += sch.iteration()
    for li in range(6):
        sub = tape.locus_subtape(steps, li)
        lo = tape.oracle_replay(data[li], sub, scaling)
        rl = tape.ref_locus_for(data[li], scaling)
        lr, _ = tape.ref_replay(rl, tape.ref_tape_arrays(sub))
        rl.free()
        assert (lo == lr).all()
        assert np.isfinite(lo).all()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("taxa,R", [(8, 4), (4, 1)])
def test_gtr_tape_with_substitution_parameter_proposals(taxa, R):
    """GTR(+Gamma): 3 frequency, 5 exchangeability and (R > 1) 1 alpha proposal per locus and iteration, each a full
    recompute with a fresh eigensystem, rejections putting the old values back — oracle == reference bit for bit"""
    data = synth.make_dataset(5, 250, taxa, "gtr", R, seed=11)
    sch = tape.make_schedule(data, seed=2, subst=True)
    steps = [sch.initial_step()]
    for _ in range(2):
        steps += sch.iteration()
    kinds = [s.kind for s in steps]
    assert kinds.count("FREQ") == 6 and kinds.count("QRATE") == 10 and kinds.count("ALPHA") == (2 if R > 1 else 0)
    assert any(s.params and s.kind in ("TAU", "MIX", "QRATE", "ALPHA", "FREQ") for s in steps)
    for li in range(5):
        sub = tape.locus_subtape(steps, li)
        assert sum(len(x["params"]) for x in sub) >= 16
        lo = tape.oracle_replay(data[li], sub)
        rl = tape.ref_locus_for(data[li])
        lr, _ = tape.ref_replay(rl, tape.ref_tape_arrays(sub))
        rl.free()
        assert np.isfinite(lo).all() and (lo == lr).all(), np.max(np.abs(lo - lr))
    # the parameter moves really move the likelihood
    sub = tape.locus_subtape(steps, 0)
    lo = tape.oracle_replay(data[0], sub)
    fq = [i for i, x in enumerate(sub) if x["kind"] in ("FREQ", "QRATE", "ALPHA")]
    assert len(set(np.round(lo[fq], 9))) > len(fq) // 2
