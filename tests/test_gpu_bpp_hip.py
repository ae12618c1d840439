"""The drop-in boundary, executed: the reference's OWN program — every object of /root/reference/src unmodified,
method.c's MCMC loop and all proposals included — linked against libbpp_amd.so through integration/locus_hip.c
(oracle/_ref/bpp_hip, recipe in oracle/Makefile) runs A00 next to the unmodified program (oracle/_ref/bpp) on the same
control file and seed.  Checked: log-L0 and the lnL column of mcmc.txt to 1e-10 relative (both are printed with 6 / 3
decimals, so this is equality of what the two programs print), every other column likewise, and whether the sample
file is byte-identical (reported; asserted where it has been observed to hold).

The binaries are built in the build container (the reference sources do not travel); on the GPU box they are used as
they come.  SURVEY.md section 8b; reference call sites method.c:4137-4300, gtree.c:5447-5467, 7484-7566,
stree.c:4727-4749, prop_mixing.c:117-131, locus.c:2704-3295."""
import os
import pytest
import bpphip as B

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not B.have_binaries(), reason="oracle/_ref/bpp{,_hip} not built (needs /root/reference)")]

G = B.GOLDEN
FROGS = {"frogs.txt": os.path.join(G, "frogs", "frogs.txt"), "frogs.Imap.txt": os.path.join(G, "frogs", "frogs.Imap.txt")}
ANOPH = {"loci_realign.txt": os.path.join(G, "anopheles", "loci_realign.txt"), "Imap.txt": os.path.join(G, "anopheles", "Imap.txt")}


@pytest.fixture(scope="module", autouse=True)
def all_program_runs_side_by_side(request):
    """every run of `bpp` / `bpp_hip` this module's selected tests will ask for, started together (the runs share nothing,
    one at a time they are 70 s of mostly waiting) — each test function is called once with its assertions ignored, its
    runs are remembered by tests/bpphip.py, and the test proper then reads them.  tests/conftest.py starts this when the
    collection is done and runs this module last, so the runs overlap the rest of the session; here they are waited for."""
    import conftest
    t = conftest._program_runs.get("thread")
    if t is not None:
        t.join()                  # the early pass: the CPU program's runs, made while the rest of the session ran
    # ... and now, with no persistent-kernel test left to disturb, the bpp_hip runs side by side
    conftest._start_program_runs(request.session, cpu_only=False)
    t = conftest._program_runs.get("thread")
    if t is not None:
        t.join()
    yield


def check(res, logl0=None):
    assert res["logl0_ref"] is not None and res["logl0_hip"] is not None
    assert abs(res["logl0_ref"] - res["logl0_hip"]) <= 1e-10*abs(res["logl0_ref"]), res
    if logl0 is not None:
        assert abs(res["logl0_hip"] - logl0) <= 1e-10*abs(logl0)
    assert res["lnl_err"] <= 1e-10, res
    assert res["all_err"] <= 1e-10, res
    print(f"samples {res['samples']}  log-L0 {res['logl0_hip']}  max rel diff lnL {res['lnl_err']:.1e}  "
          f"all columns {res['all_err']:.1e}  mcmc.txt byte-identical: {res['identical']}")


def test_frogs_a00():
    """BASELINE config 1: examples/frogs A00 (5 loci, unphased diploids: 42-60 tips after phasing, the diploid
    averaging of locus.c:2586-2615), known log-L0 = -7370.742841"""
    res = B.compare_runs(B.FROGS_CTL.format(burnin=40, sampfreq=2, nsample=80, extra=""), FROGS)
    check(res, -7370.742841)
    assert res["identical"]


def test_frogs_a00_scaling():
    """the same with `scaling = 1` (per-pattern scalers through every proposal's index toggling)"""
    res = B.compare_runs(B.FROGS_CTL.format(burnin=10, sampfreq=2, nsample=30, extra="scaling = 1"), FROGS)
    check(res)


def test_anopheles_msci_a00():
    """BASELINE config 5: examples/anopheles MSC-I A00 (100 loci x 12 sequences, cleandata = 1, the introgression
    model of anopheles-bpp-msci.ctl: its species-tree moves are the reference's own code, its likelihood ours);
    log-L0 = -82303.942488 with this control file in the build container)"""
    ctl = B.ANOPHELES_CTL.format(tree=B.ANOPHELES_MSCI_TREE, phiprior="phiprior = 1 1", burnin=10, sampfreq=2, nsample=25, extra="")
    res = B.compare_runs(ctl, ANOPH)
    check(res)
    assert res["identical"]


def test_synthetic_c2_like_200_loci():
    """a config-2-like set from the reference's own simulator: 200 loci x 1 000 sites, 4 species, JC69"""
    files = B.simulate(B.SIM_CTL.format(seed=12345, species=B.SPECIES4_SIM, phase="0 0 0 0", nloci=200, sites=1000, simmodel=0, extra=""))
    ctl = B.A00_CTL.format(species=B.SPECIES4, phase="0 0 0 0", nloci=200, model="jc69", alpha="", taub=500,
                           burnin=20, sampfreq=2, nsample=60, extra="")
    res = B.compare_runs(ctl, files)
    check(res, -299317.884924)
    assert res["identical"]


def test_synthetic_gtr_gamma():
    """a config-3-like set: 40 loci x 500 sites, 8 species, GTR+G4 — the frequency / exchangeability / alpha proposals
    live inside locus.c (locus.c:2782-3419) and prop_gamma.c and reach the library through the same three calls"""
    files = B.simulate(B.SIM_CTL.format(seed=7, species=B.SPECIES8_SIM, phase="0 0 0 0 0 0 0 0", nloci=40, sites=500, simmodel=7,
                                        extra="alpha_siterate = 1 0.5 4\nqrates = 1 1 2 1 0.5 1.5 1\nbasefreqs = 1 0.3 0.2 0.2 0.3\nmodelparafile = syn.para.txt\n"))
    ctl = B.A00_CTL.format(species=B.SPECIES8, phase="0 0 0 0 0 0 0 0", nloci=40, model="gtr", alpha="alphaprior = 1 1 4",
                           taub=300, burnin=10, sampfreq=2, nsample=30, extra="")
    res = B.compare_runs(ctl, files)
    check(res)


def test_simulator_runs_pll_core_update_pmatrix_on_the_device():
    """`--simulate` with GTR + Gamma site rates draws every branch's sequence from a P-matrix made by
    pll_core_update_pmatrix (simulate.c:694-705, one call per branch and site rate): in bpp_hip that is the shim's
    definition -> bpa_core_update_pmatrix on the GPU.  Same seed: the two programs write the same alignment, byte for
    byte, and the same model-parameter file"""
    ctl = B.SIM_CTL.format(seed=11, species=B.SPECIES4_SIM, phase="0 0 0 0", nloci=6, sites=120, simmodel=7,
                           extra="alpha_siterate = 1 0.5 4\nqrates = 1 1 2 1 0.5 1.5 1\nbasefreqs = 1 0.3 0.2 0.2 0.3\nmodelparafile = syn.para.txt\n")
    cpu = B.simulate(ctl)
    gpu = B.simulate(ctl, binary=B.HIP_BIN)
    assert cpu["syn.txt"] == gpu["syn.txt"] and len(cpu["syn.txt"]) > 6 * 4 * 120
    assert cpu.get("syn.para.txt") == gpu.get("syn.para.txt")
    assert len(set(cpu["syn.txt"].split()[3])) > 1         # (sequences evolved, not constant)


def test_threads_2():
    """threads.c shards the loci over pthreads: calls for different loci arrive concurrently"""
    files = B.simulate(B.SIM_CTL.format(seed=3, species=B.SPECIES4_SIM, phase="0 0 0 0", nloci=64, sites=400, simmodel=0, extra=""))
    ctl = B.A00_CTL.format(species=B.SPECIES4, phase="0 0 0 0", nloci=64, model="jc69", alpha="", taub=500,
                           burnin=10, sampfreq=2, nsample=30, extra="threads = 2 1 1")
    res = B.compare_runs(ctl, files)
    check(res)


def _syn4(nloci=24, sites=300, seed=3):
    return B.simulate(B.SIM_CTL.format(seed=seed, species=B.SPECIES4_SIM, phase="0 0 0 0", nloci=nloci, sites=sites, simmodel=0, extra=""))


def _a00(nloci, model="jc69", alpha="", extra="", nsample=30):
    return B.A00_CTL.format(species=B.SPECIES4, phase="0 0 0 0", nloci=nloci, model=model, alpha=alpha, taub=500,
                            burnin=10, sampfreq=2, nsample=nsample, extra=extra)


@pytest.mark.parametrize("clock", ["clock = 2 10 100 5 iid G", "clock = 3 10 100 5 iid G"])
def test_relaxed_clocks(clock):
    """independent- and correlated-rates clocks with estimated locus rates: branch lengths come from the reference's own
    update_branchlength_relaxed_clock (locus.c:1150) inside the shim, the branch-rate and locus-rate proposals
    (stree.c: prop_branch_rates, prop_locusrate_*) call the library through the same three entry points"""
    res = B.compare_runs(_a00(24, extra=clock + "\nlocusrate = 1 5 5 2 iid"), _syn4())
    check(res)


def test_locusrate_and_heredity():
    """strict clock with per-locus rates (gtree->rate_mui in every branch length, locus.c:2347) and heredity scalars"""
    res = B.compare_runs(_a00(24, extra="locusrate = 1 5 5 2 iid\nheredity = 1 4 4"), _syn4())
    check(res)


@pytest.mark.parametrize("model,alpha", [("hky", "alphaprior = 1 1 4"), ("k80", ""), ("tn93", ""), ("f81", ""), ("t92", ""), ("f84", "")])
def test_closed_form_models_through_the_program(model, alpha):
    """K7 (locus.c:1981-2323) as the reference program drives it: model-specific parameter proposals included"""
    res = B.compare_runs(_a00(16, model=model, alpha=alpha, nsample=20), _syn4(16))
    check(res)


def test_species_tree_inference_a01():
    """A01: the species-tree SPR / NNI moves (stree.c) re-evaluate gene trees of all loci through the library"""
    ctl = _a00(16, nsample=25).replace("speciestree = 0\n", "speciestree = 1\n")
    res = B.compare_text_runs(ctl, _syn4(16))
    assert abs(res["logl0_ref"] - res["logl0_hip"]) <= 1e-10*abs(res["logl0_ref"])
    assert res["all_err"] <= 1e-10, res
    print(f"A01: {res['samples']} samples, numbers agree to {res['all_err']:.1e}, byte-identical: {res['identical']}")


def test_species_delimitation_a11():
    """A11: rjMCMC species delimitation + species-tree moves on the library"""
    ctl = _a00(16, nsample=25).replace("speciesdelimitation = 0\n", "speciesdelimitation = 1 1 2 1\n").replace("speciestree = 0\n", "speciestree = 1\nspeciesmodelprior = 1\n")
    res = B.compare_text_runs(ctl, _syn4(16))
    assert abs(res["logl0_ref"] - res["logl0_hip"]) <= 1e-10*abs(res["logl0_ref"])
    assert res["all_err"] <= 1e-10, res
    print(f"A11: {res['samples']} samples, numbers agree to {res['all_err']:.1e}, byte-identical: {res['identical']}")


@pytest.mark.parametrize("first,second", [("hip", "hip"), ("ref", "hip"), ("hip", "ref")])
def test_checkpoint_and_resume(first, second):
    """`checkpoint = 30 1000` + `--resume` (dump.c / load.c:2016-2140): a restored locus gets its tip CLVs and pattern
    weights written straight into the host struct, with no setter call — the shim reads them off the struct before the
    first update.  A run resumed from its checkpoint ends with the sample file of the uninterrupted run, whichever of the
    two programs wrote the checkpoint and whichever resumed it (the checkpoint holds no likelihood state: load.c
    recomputes all matrices and partials through the locus API)."""
    bins = {"ref": B.REF_BIN, "hip": B.HIP_BIN}
    files = B.simulate(B.SIM_CTL.format(seed=9, species=B.SPECIES8_SIM, phase="0 0 0 0 0 0 0 0", nloci=12, sites=300, simmodel=7,
                                        extra="alpha_siterate = 1 0.5 4\nqrates = 1 1 2 1 0.5 1.5 1\nbasefreqs = 1 0.3 0.2 0.2 0.3\nmodelparafile = syn.para.txt\n"))
    ctl = B.A00_CTL.format(species=B.SPECIES8, phase="0 0 0 0 0 0 0 0", nloci=12, model="gtr", alpha="alphaprior = 1 1 4",
                           taub=300, burnin=10, sampfreq=2, nsample=40, extra="checkpoint = 30 1000")
    rc, out, outs = B.run_program(B.REF_BIN, ctl.replace("checkpoint = 30 1000", ""), files)
    assert rc == 0
    want = outs["out.mcmc.txt"]
    m0, m1, out2 = B.run_checkpointed(bins[first], bins[second], ctl, files)
    assert m0 == want                      # the checkpointing run itself
    assert m1 == want                      # ... and the run resumed from the checkpoint
    if second == "hip":
        assert "Likelihood back-end: bpp_amd" in out2


def test_checkpoint_and_resume_diploid():
    """the same on the frogs files (unphased diploids: the restored loci also carry the A1 -> A3 mapping, resolution
    counts and unphased weights, which load.c writes into the struct and the shim hands to bpa_set_diploid)"""
    ctl = B.FROGS_CTL.format(burnin=10, sampfreq=2, nsample=40, extra="checkpoint = 30 1000")
    rc, out, outs = B.run_program(B.REF_BIN, ctl.replace("checkpoint = 30 1000", ""), FROGS)
    assert rc == 0
    m0, m1, out2 = B.run_checkpointed(B.HIP_BIN, B.HIP_BIN, ctl, FROGS)
    assert m0 == outs["out.mcmc.txt"] and m1 == outs["out.mcmc.txt"]
    assert "Likelihood back-end: bpp_amd" in out2


def test_anopheles_mscm_migration():
    """examples/anopheles, the MSC-M model of anopheles-bpp-mscm.ctl (two migration bands): the migration-rate and
    migration-event proposals are the reference's own code, every likelihood they ask for is the library's"""
    tree = "(R, ((C, G) b, ((A, Q) d, L) c ) a) o;"
    ctl = B.ANOPHELES_CTL.format(tree=tree, phiprior="wprior = 20 1\nmigration = 2\n  A b\n  R Q", burnin=10, sampfreq=2, nsample=20, extra="")
    res = B.compare_runs(ctl, ANOPH)
    check(res)
    assert res["identical"]


def test_bayes_factor_beta_and_prior_only():
    """`BayesFactorBeta = b` (opt_bfbeta: locus_root_loglikelihood returns b x lnL, locus.c:2630) and `usedata = 0` (it
    returns 0 and the update calls do nothing, locus.c:2420, 2533, 2580): both reach the library through
    bpa_engine_set_options"""
    f = _syn4(16)
    res = B.compare_runs(_a00(16, nsample=20, extra="BayesFactorBeta = 0.4"), f)
    check(res)
    assert res["identical"]
    res = B.compare_text_runs(_a00(16, nsample=20).replace("usedata = 1\n", "usedata = 0\n"), f)      # (no lnL column)
    assert res["identical"] and res["all_err"] == 0.0


SPECIES6 = """6  A B C D E F
                  1 1 1 1 1 1
                  ((((A, B), C), (D, E)), F);"""


def _aa_files(nloci, sites, seed=3):
    """amino-acid alignments (the reference's simulator makes DNA only): bpp_amd.synth's LG + Gamma data, the patterns
    written out `weight` times each, in the reference's sequential PHYLIP + Imap form"""
    from bpp_amd import synth
    data = synth.make_dataset(nloci, sites, 6, "lg", 4, seed=seed)
    names = "ABCDEF"
    txt = ""
    for d in data:
        cols = [j for j, w in enumerate(d["weights"]) for _ in range(int(w))]
        txt += f"\n6 {len(cols)}\n\n"
        for t in range(6):
            txt += f"{names[t]}^{names[t].lower()}1        " + "".join(d["seqs"][t][j] for j in cols) + "\n"
        txt += "\n"
    return {"syn.txt": txt, "syn.Imap.txt": "".join(f"{c.lower()}1\t{c}\n" for c in names)}


@pytest.mark.parametrize("model,alpha", [("lg", "alphaprior = 1 1 4"), ("wag", ""), ("jtt", "alphaprior = 1 1 4")])
def test_amino_acid_models_through_the_program(model, alpha):
    """config 4's kind of locus as the reference program drives it: 20 states, empirical rate matrices (K6 on the device),
    one or four rate categories with the alpha proposal — the 20-state pipelined K1+K2 kernel and the workgroup-per-branch
    K5 kernel under method.c's MCMC loop"""
    ctl = B.A00_CTL.format(species=SPECIES6, phase="0 0 0 0 0 0", nloci=10, model=model, alpha=alpha, taub=40, burnin=10,
                           sampfreq=2, nsample=25, extra="").replace("thetaprior = gamma 2 1000", "thetaprior = gamma 2 100")
    res = B.compare_runs(ctl, _aa_files(10, 250))
    check(res)
