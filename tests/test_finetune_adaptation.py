"""The burn-in's step-length rule (reset_finetune_onestep, method.c:1122-1136; pj_optimum = 0.3, method.c:45) and the
burn-in built on it (bpa_sampler_burnin: method.c:5364-5377, 1508-1516).

 * CPU: bpa_finetune_onestep against the reference's arithmetic, restated here from the cited lines;
 * GPU: the device sampler with the program's moves, from the program's default step lengths (bpp.c:530-549), burns in to the
   step lengths the unmodified program (oracle/_ref/bpp, finetune = 1) burns in to on the same data — the numbers its
   "finetune = 1 Gage:... " line prints — and its acceptance proportions afterwards sit where the rule aims (0.3).
"""
import math
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import bpp_amd
from bpp_amd import synth
import oraclelib as O


def reference_onestep(pjump, ft):
    """method.c:1122-1136"""
    if pjump < 0.001:
        return ft / 100
    if pjump > 0.999:
        return min(99.0, ft * 100)
    return min(99.0, ft * math.tan(math.pi / 2 * pjump) / math.tan(math.pi / 2 * 0.3))


def test_onestep_rule_is_the_reference_s():
    L = bpp_amd.lib()
    rng = np.random.default_rng(3)
    cases = [(0.0, 5.0), (0.0009, 5.0), (0.001, 5.0), (0.3, 0.001), (0.2999, 7.0), (0.999, 0.3), (0.9991, 0.3), (1.0, 2.0), (0.95, 60.0), (0.5, 98.0)]
    cases += [(float(p), float(f)) for p, f in zip(rng.uniform(0, 1, 200), 10 ** rng.uniform(-6, 2, 200))]
    for pj, ft in cases:
        got, want = L.bpa_finetune_onestep(pj, ft), reference_onestep(pj, ft)
        assert got == pytest.approx(want, rel=1e-14, abs=0), (pj, ft)
    assert L.bpa_finetune_onestep(0.3, 0.123) == pytest.approx(0.123, rel=1e-15)     # at the optimum nothing moves
    assert L.bpa_finetune_onestep(0.6, 98.0) == 99.0                                  # maxstep


def reference_reset_points(burnin):
    """the iterations after which the program's loop resets the step lengths (method.c:5343-5417: i runs from -burnin; the reset
    is at the top of iteration i when i == 0, or i < 0, opt_burnin >= 200, ft_round >= 100 and i % (opt_burnin/4) == 0 — C's
    remainder, which is 0 for the same i as Python's; ++ft_round at the end of the body)"""
    pts, ft_round = [], 0
    for i in range(-burnin, 1):
        if i == 0 or (burnin >= 200 and ft_round >= 100 and i % (burnin // 4) == 0):
            if burnin >= 200:
                pts.append(i + burnin)
            ft_round = 0
        ft_round += 1
    return pts


def test_burnin_resets_where_the_program_s_loop_does():
    import ctypes as C
    L = bpp_amd.lib()
    assert reference_reset_points(400) == [100, 200, 300, 400]
    assert reference_reset_points(300) == [150, 300]          # (a quarter is 75 < 100 iterations: every second one)
    assert reference_reset_points(402) == [102, 202, 302, 402]
    assert reference_reset_points(199) == []
    for burnin in list(range(0, 1300)) + [2000, 4000, 8000, 10007, 100000]:
        buf = (C.c_uint * 16)()
        n = L.bpa_burnin_schedule(burnin, buf, 16)
        assert list(buf[:n]) == reference_reset_points(burnin), burnin


@pytest.mark.gpu
@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_burnin_arrives_where_the_program_s_does():
    import tape
    nloci, burnin = 400, 2000
    data = synth.make_dataset(nloci, 1000, 4, "jc69", 1, seed=77)
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "seqs.txt"), "w") as f:
            for d in data:
                seqs = ["".join(ch * int(w) for ch, w in zip(s, d["weights"])) for s in d["seqs"]]
                f.write(f"4 {len(seqs[0])}\n")
                for nm, s in zip("abcd", seqs):
                    f.write(f"s^{nm}  {s}\n")
                f.write("\n")
        open(os.path.join(td, "imap.txt"), "w").write("a A\nb B\nc C\nd D\n")
        open(os.path.join(td, "a00.ctl"), "w").write(
            "seed = 1\nseqfile = seqs.txt\nImapfile = imap.txt\njobname = out\nspeciesdelimitation = 0\n"
            "speciestree = 0\nspecies&tree = 4  A B C D\n                  1 1 1 1\n                 (((A, B), C), D);\nusedata = 1\n"
            f"nloci = {nloci}\ncleandata = 0\nthetaprior = gamma 2 1000\ntauprior = gamma 2 500\nfinetune = 1\nprint = 1 0 0 0\n"
            f"burnin = {burnin}\nsampfreq = 1\nnsample = 200\n")
        r = subprocess.run([O.REF_BIN, "--cfile", "a00.ctl"], cwd=td, capture_output=True, text=True, timeout=600)
    ft = re.findall(r"finetune = 1 Gage:(\S+) Gspr:(\S+) th1:(\S+) th2:(\S+) tau:(\S+) mix:([0-9.eE+-]+)", r.stdout)
    assert ft, r.stdout[-2000:]
    prog = dict(zip(("gage", "gspr", "th1", "theta", "tau", "mix"), map(float, ft[-1])))

    eng = bpp_amd.Engine(0)
    smp = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=11)
    parent, tau, theta = synth.species_tree_arrays(4)
    smp.set_species_tree(parent, tau, theta)
    smp.set_tau_prior(2.0, 500.0)
    smp.set_proposal_kernel(1)
    smp.set_program_moves(True, 0.1)
    smp.set_theta_prior(2.0, 1000.0, 0.001)           # opt_finetune_theta[0], bpp.c:549
    smp.set_finetune(5.0, 0.001, 0.001, 0.3)           # Gage, Gspr, tau, mix: bpp.c:530-546
    smp.initialize()
    assert smp.kind() == "persistent"
    dev = smp.burnin(burnin)
    # the same rule on the same target from the same start: the two burn-ins end within a factor of each other that their own
    # run-to-run scatter explains (two seeds of the program differ by up to ~25 % in tau / mix; the theta window, proposed one
    # time in ten, by more)
    for k, tol in (("gage", 1.15), ("gspr", 1.15), ("tau", 1.6), ("mix", 1.6), ("theta", 2.0)):
        assert 1 / tol < dev[k] / prog[k] < tol, (k, dev[k], prog[k])
    assert dev["gage"] != 5.0 and dev["tau"] != 0.001
    # ... and the chain then accepts what the rule aims at
    smp.iterate(1500)
    pj, ft2 = smp.adapt_finetune()
    for k in ("gspr", "tau", "mix"):
        assert 0.18 < pj[k] < 0.42, (k, pj)
    assert 0.25 < pj["gage"] < 0.55, pj               # (the age window is capped by its bounds: the program's own pjump there is ~0.4)
    # a second call right away has nothing to go on: the counters were cleared, the step lengths stay
    pj0, ft3 = smp.adapt_finetune()
    assert all(v < 0 for v in pj0.values()) and ft3 == ft2
    smp.close(); eng.close()


@pytest.mark.gpu
def test_the_rule_on_the_generic_sampler():
    """GTR + Gamma loci (the generic sampler, the program's moves decided on the host): the uniform-window iteration has no
    step-length rule; with the program's moves the burn-in from the program's defaults brings every move type to the rule's aim"""
    import tape
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(300, 500, 8, "gtr", 4, seed=3)
    smp = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=1)
    parent, tau, theta = synth.species_tree_arrays(8)
    smp.set_species_tree(parent, tau, theta)
    smp.set_tau_prior(3.0, 3.0 / tau[-1])
    smp.set_theta_prior(2.0, 1000.0, 0.001)
    smp.set_finetune(5.0, 0.001, 0.001, 0.3)
    smp.initialize()
    smp.iterate(2)
    with pytest.raises(bpp_amd.BpaError):
        smp.adapt_finetune()
    smp.close()
    smp = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=1)
    smp.set_proposal_kernel(1)
    smp.set_program_moves(True, 0.1)
    smp.set_species_tree(parent, tau, theta)
    smp.set_tau_prior(3.0, 3.0 / tau[-1])
    smp.set_theta_prior(2.0, 1000.0, 0.001)
    smp.set_finetune(5.0, 0.001, 0.001, 0.3)
    smp.initialize()
    assert smp.kind() == "generic"
    ft = smp.burnin(800)
    assert ft["gage"] != 5.0 and ft["tau"] != 0.001 and ft["mix"] != 0.3
    smp.iterate(300)
    pj, _ = smp.adapt_finetune()
    for k in ("gspr", "tau", "mix"):
        assert 0.15 < pj[k] < 0.45, (k, pj, ft)
    assert 0.2 < pj["gage"] < 0.6, (pj, ft)
    g = smp.gibbs_counters()
    assert g[0] > 0 and g[1] > 0.9 * g[0]             # (the metropolized Gibbs draws are nearly always accepted)
    smp.close(); eng.close()
