#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on the configuration it is quoted on (configs[1]): MCMC iterations/s of A00 on
synthetic 10 000 loci x 1 000 sites, 4 taxa, JC69, 1 rate category, at N GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4] [--scaling weak|strong] ...

A "step" is `iterations_per_step` MCMC ITERATIONS of the device-resident A00 sampler over all loci (config c2) — an
iteration: per locus tips-1 gene-node age proposals and 2 tips-2 prune/regraft proposals, a theta step per population, a
rubber-band tau step per divergence and one mixing step, every proposal, density, likelihood and accept/reject on the
GPU (bpa_sampler_t; same posterior as the unmodified program, tests/test_a00_posterior.py).  The count is chosen so that
the timed region lasts >= 0.25 s whatever --steps is (an iteration takes ~0.1 ms).  On one GPU all iterations of a step
are ONE persistent launch (csrc/sweep2.hpp).  `value` is iterations/s; `roofline` is that kernel's; `cpu_baseline` is the
unmodified reference program's whole MCMC iterations/s at its best thread count on this box (like for like with `value`).
Next to it, as before:

  likelihood_only   the hot path alone on a pre-recorded proposal tape (bpp_amd/schedule.py; accept/reject by a seeded
                    coin): resident batched plans, the per-locus steps of an iteration as one chain launch, the all-loci
                    steps one launch each.  This is also what `--config c3|c4` time (`value` then: tape iterations/s of
                    that config's own loci), and what `other_configs` carries for c3 / c4 in the default run.
  cpu_tape_replay   the same tape through the REAL reference's locus API (oracle/_ref, AVX2), one core and all cores
                    (comparable with likelihood_only; `speedups` holds both ratios).
  reference_program_on_host   the unmodified program's whole MCMC iterations/s by thread count (`cpu_baseline` = its best).

N > 1 (torch.distributed, one rank per GPU): loci sharded, no data-path collective; the only exchange is the sum an
all-loci step is decided on.  --scaling weak (default; every rank owns the config's loci, `value` in 10k-locus
iterations/s x N) or strong (ONE data set of the config's loci dealt out by the reference's zig-zag, threads.c:265-353).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # as bpp_amd/__init__.py does (here too: torch may bring the HIP runtime in first when N > 1)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
BASELINE_METRIC = "MCMC iterations/sec (A00) + site-lnL updates/sec, 10k loci, 1/2/4/8 GPU"      # BASELINE.json's `metric`, verbatim

FP64_PEAK_TFLOPS = 78.6      # MI355X_MICROARCH.md: FP64 vector peak (the FP64 matrix rate is the same on gfx950)
HBM_ACHIEVABLE_FRAC = 0.79   # ~6.3 of 8 TB/s is what a streaming kernel reaches (MI355X_MICROARCH.md): a `frac` above it is an accounting artefact
LINE_LIMIT = 6000            # bytes of the final stdout line (the driver's capture parsed 18.5 KB in round 3 and lost a 26 KB line in round 4)


def _rl_compact(r):
    """the roofline object of the compact line: the contract's keys + the two honest companions of `frac`"""
    if not isinstance(r, dict):
        return None
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_pmc", "frac_codes", "flops_frac",
            "avg_kernel_us", "launches", "algorithmic_bytes_per_launch", "codes_bytes_per_launch", "profile_head", "warning", "frac_pmc_time_us", "pmc")
    o = {k: r[k] for k in keep if r.get(k) is not None}
    o.setdefault("traffic", None)
    if isinstance(o.get("kernel"), str) and len(o["kernel"]) > 60:
        o["kernel"] = o["kernel"][:57] + "..."
    return o


def _num(x, nd=3):
    return round(x, nd) if isinstance(x, float) else x


def compact_line(full):
    """The ONE line the driver parses: the contract's keys, `roofline`, `cpu_baseline`, and per BASELINE config a handful of
    numbers.  Everything else is in bench_full.json (written next to this file and under gpurun_out/)."""
    g = full.get
    cb = g("cpu_baseline") or {}
    eff = g("ess_per_s") or {}
    se = g("statistical_efficiency") or {}
    lo = g("likelihood_only") or {}
    out = {
        "metric": g("metric"), "value": g("value"), "unit": "iterations/s", "n_gpus": g("n_gpus"), "steps": g("steps"), "warmup": g("warmup"),
        "ms_per_step": g("ms_per_step"), "iterations_per_step": g("iterations_per_step"), "ms_per_iteration": g("ms_per_iteration"),
        "higher_is_better": True, "scaling": g("scaling"), "vs_baseline": g("vs_baseline"), "dtype": g("dtype"), "data": g("data"),
        "config": {"workload": (g("config") or {}).get("workload_short") or (g("config") or {}).get("workload", "")[:160],
                   "parallelism": ((g("config") or {}).get("parallelism") or "")[:120], "moves": (g("config") or {}).get("moves")},
        "roofline": _rl_compact(g("roofline")),
        "cpu_baseline": ({"value": cb.get("value"), "unit": "iterations/s", "cores": cb.get("cores"), "kind": cb.get("kind"),
                          "sample": (cb.get("sample_short") or cb.get("sample") or "")[:200], "one_thread": cb.get("one_thread"),
                          "host_cpu_quota": cb.get("host_cpu_quota")} if cb else None),
        "value_over_cpu_baseline": (g("speedups") or {}).get("value_over_cpu_baseline"),
        "vs_survey_measurement": (g("vs_survey_measurement").get("ratio") if isinstance(g("vs_survey_measurement"), dict) else g("vs_survey_measurement")),
        "site_lnl_updates_per_s": g("site_lnl_updates_per_s"),
    }
    if g("value_weak") is not None or g("value_strong") is not None:
        out["value_weak"], out["value_strong"] = g("value_weak"), g("value_strong")
        out["allreduce"] = ((g("allreduce") or {}).get("sampler") or "")[:100] or None
        out["allreduce_check"] = g("allreduce_check")
    if eff:
        out["ess_per_s"] = {"ratio": eff.get("ratio"), "device": eff.get("device"), "reference_program": eff.get("reference_program")}
        epi = {}
        for who in ("device", "reference_program"):
            for p in ("tau_root", "theta_root"):
                v = ((se.get(who) or {}).get(p) or {})
                if v.get("ess_per_iteration") is not None:
                    epi.setdefault(who, {})[p] = [v.get("ess_per_iteration"), v.get("ess_per_iteration_se")]
        if epi:
            out["ess_per_iteration"] = epi
        if se.get("ratio_ess_per_iteration"):
            out["ess_per_iteration_ratio"] = {k: [v["value"], v["se"]] for k, v in se["ratio_ess_per_iteration"].items() if isinstance(v, dict)}
            out["ess_per_iteration_ratio"]["samples"] = [(se.get("device") or {}).get("samples"), (se.get("reference_program") or {}).get("samples")]
    # the committed long runs on the SAME data for both chains (tools/ess_device_vs_program.py): [ratio, standard error] — a
    # builder-run measurement inside a driver-run line, labelled as such (the bench's own ESS section is opt-in: --efficiency)
    try:
        import glob
        lr = {}
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "ess_*.json"))):
            e_ = json.load(open(f))
            lr[f"{e_['nloci']}x{e_['samples']}"] = {k: [e_[k + "_ratio"]["value"], e_[k + "_ratio"]["se"]] for k in ("tau_root", "theta_root")}
        if lr:
            out["ess_per_iteration_ratio_committed_runs"] = lr
    except Exception:       # noqa: BLE001
        pass
    if lo:
        r = lo.get("roofline") or {}
        out["likelihood_only"] = {"it_s": lo.get("iterations_per_s"), "site_lnl_updates_per_s": lo.get("site_lnl_updates_per_s"),
                                  "kernel": (r.get("kernel") or "")[:50], "frac": r.get("frac"),
                                  "frac_codes": r.get("frac_codes"), "frac_pmc": r.get("frac_pmc"), "flops_frac": r.get("flops_frac"),
                                  "avg_kernel_us": r.get("avg_kernel_us")}
    hc = g("host_control_in_c") or {}
    if hc.get("iterations_per_s"):
        out["host_control_in_c_it_s"] = hc["iterations_per_s"]
    uk = (g("device_resident_sampler") or {}).get("device_uniform_kernel") or {}
    if uk.get("iterations_per_s"):
        out["uniform_moves_it_s"] = uk["iterations_per_s"]
    proj = g("scale_projection")
    if isinstance(proj, dict) and "error" not in proj:
        out["share_it_s"] = {k: _num(v.get("iterations_per_s"), 1) for k, v in proj.items() if isinstance(v, dict)}
    cfgs = {}
    for key, sec in (g("other_configs") or {}).items():
        if not isinstance(sec, dict):
            continue
        if "error" in sec:
            cfgs[key] = {"error": str(sec["error"])[:80]}
            continue
        smp = sec.get("device_resident_sampler") or {}
        c = {"it_s": _num(smp.get("iterations_per_s", sec.get("iterations_per_s")), 2)}
        if sec.get("iterations_per_s") is not None and smp:
            c["tape_it_s"] = _num(sec.get("iterations_per_s"), 2)
        if isinstance(sec.get("cpu_baseline"), dict):
            c["cpu_it_s"] = _num(sec["cpu_baseline"].get("value"), 3)
            c["cpu_cores"] = sec["cpu_baseline"].get("cores")
            if c["cpu_it_s"] and c["it_s"]:
                c["ratio"] = round(c["it_s"] / c["cpu_it_s"], 2)
        if smp.get("moves_short") or sec.get("moves"):
            c["moves"] = smp.get("moves_short") or str(sec.get("moves"))[:40]
        if smp.get("launches_per_iteration") is not None:
            c["launches_per_it"] = _num(smp.get("launches_per_iteration"), 1)
        r = sec.get("roofline") or {}
        for k in ("frac", "frac_codes", "frac_pmc", "flops_frac", "avg_kernel_us"):
            if r.get(k) is not None:
                c[k] = r[k]
        if isinstance(r.get("pmc"), dict):
            c["wait_any"] = r["pmc"].get("wait_any")          # share of the kernel's wave-cycles spent waiting (committed SQ passes)
        if r.get("kernel"):
            c["kernel"] = r["kernel"].split("<")[0]
        if r.get("warning"):
            c["frac_warning"] = "frac > 0.79 of HBM peak is an accounting artefact (8d charges tip children as CLVs): read frac_codes / frac_pmc"
        rs = smp.get("roofline") or {}
        if rs.get("frac") is not None:
            c["sampler_frac"] = rs.get("frac")
            if rs.get("frac_codes") is not None:
                c["sampler_frac_codes"] = rs.get("frac_codes")
        sp = sec.get("scale_projection")
        if isinstance(sp, dict) and "error" not in sp:
            c["share_it_s"] = {k: _num(v.get("iterations_per_s"), 1) for k, v in sp.items() if isinstance(v, dict)}
        cfgs[key] = c
    if cfgs:
        out["configs"] = cfgs
    out["full_record"] = "bench_full.json"
    # never let the line outgrow the driver's capture: drop the optional parts, least important first
    for victim in ("share_it_s", "host_control_in_c_it_s", "uniform_moves_it_s", "ess_per_iteration_ratio_committed_runs", "ess_per_iteration", "likelihood_only", "configs", "ess_per_s"):
        if len(json.dumps(out, separators=(",", ":"))) <= LINE_LIMIT:
            break
        out.pop(victim, None)
    return out


def write_full_record(full, where=None):
    for path in ((where,) if where else (os.path.join(ROOT, "bench_full.json"), os.path.join(ROOT, "gpurun_out", "bench_full.json"))):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    json.dump(full, f, indent=1)
        except OSError as ex:
            log(f"bench_full.json not written to {path}: {ex}")


CONFIGS = {
    "c2": dict(taxa=4, model="jc69", rate_cats=1, sites=1000, loci=10000, taus=(0.001, 0.002, 0.003),
               name="C2: A00, 10000 loci x 1000 sites, 4 taxa, JC69, 1 rate cat"),
    "c3": dict(taxa=8, model="gtr", rate_cats=4, sites=1000, loci=10000, taus=(0.0011, 0.0025, 0.005),
               name="C3: A00, 10000 loci x 1000 sites, 8 taxa, GTR+G4"),
    # divergence 3: theta 0.06, tau_root 0.15 — ~195 distinct patterns per locus, the per-locus work BASELINE.md's reference
    # row was measured on (its ad-hoc alignment: ~200); SURVEY 8d's literal theta 0.02 / tau_root 0.05 gives 105 (--c4-divergence 1)
    "c4": dict(taxa=6, model="lg", rate_cats=4, sites=500, loci=2000, taus=(0.01, 0.015, 0.02, 0.035, 0.05), divergence=3.0,
               name="C4: A00, 2000 loci x 500 aa sites, 6 taxa, LG+G4"),
}


_T0 = time.perf_counter()


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:6.1f}s]", *a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------ CPU baselines ---
def cpu_quota():
    """CPUs this container may use at once (cgroup v2 cpu.max), or None: the GPU box shows 256 logical cores but grants 16"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:       # noqa: BLE001
        return None


def cpu_baseline(data, steps_init, steps_iter, n_iter, budget_s=10.0, sample=192):
    """The same tape on the host CPU through the REAL reference's update API when oracle/_ref travelled (kind
    "reference"), else through the oracle's C loop (kind "port"), on a bounded sample of the loci: one core, then —
    loci are independent, as threads.c shards them — all host cores at once (one locus per worker at a time)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    import tape
    from concurrent.futures import ThreadPoolExecutor
    sample = min(sample, len(data))
    use_ref = O.have_ref()
    t_start = time.time()

    def prep(li):
        sub_full = tape.locus_subtape(steps_init + steps_iter, li)
        sub_init = tape.locus_subtape(steps_init, li)
        if use_ref:
            return (tape.ref_locus_for(data[li]), tape.ref_tape_arrays(sub_full), tape.ref_tape_arrays(sub_init))
        return (data[li], sub_full, sub_init)

    def run(job, reps):
        if use_ref:
            rl, full, init = job
            _, t_full = tape.ref_replay(rl, full, repeats=reps)
            _, t_init = tape.ref_replay(rl, init, repeats=reps)
        else:
            d, full, init = job
            _, t_full = tape.oracle_tape_run(d, full, repeats=reps)
            _, t_init = tape.oracle_tape_run(d, init, repeats=reps)
        return t_full, t_init

    def run_full(job, reps):
        if use_ref:
            tape.ref_replay(job[0], job[1], repeats=reps)
        else:
            tape.oracle_tape_run(job[0], job[1], repeats=reps)

    jobs = [prep(li) for li in range(sample)]
    # ---- one core: calibrate the repeats on the first locus, then ~budget_s/2
    t_full, t_init = run(jobs[0], 50)
    reps = int(max(5, min(20000, 50 * (0.5 * budget_s) / max((t_full + t_init) * sample, 1e-9))))
    per_locus, legs = [], []
    for job in jobs:
        t_full, t_init = run(job, reps)
        legs.append((t_full, t_init))
        per_locus.append(max(t_full - t_init, 1e-12) / (reps * n_iter))
        if time.time() - t_start > 1.5 * budget_s:
            break
    one = float(np.mean(per_locus))
    # ---- all cores: the C calls release the GIL; every worker replays whole loci (full tape), wall time of the sample;
    # the start-up evaluation's share of a replay is known from the one-core leg
    cores = os.cpu_count() or 1
    share_iter = float(np.sum([max(f - i, 0.0) for f, i in legs]) / max(np.sum([f for f, _ in legs]), 1e-12))
    # worker counts: what the container's CPU quota grants (cgroup cpu.max; more threads than that are only throttled)
    # and every logical core the sample can feed; the better of the two is the baseline, both are reported
    q = cpu_quota()
    cands = sorted({max(1, min(cores, len(jobs)))} | ({max(1, min(int(q), len(jobs)))} if q else set()))
    tried, best = {}, None
    for workers in cands:
        reps_all = int(max(5, min(20000, reps * min(workers, 16))))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(lambda j: run_full(j, reps_all), jobs))
        wall = time.perf_counter() - t0
        spli = wall * share_iter / (len(jobs) * reps_all * n_iter)
        tried[str(workers)] = spli
        if best is None or spli < best[0]:
            best = (spli, workers, reps_all)
    all_sec_per_locus_iter, workers, reps_all = best
    if use_ref:
        for rl, _, _ in jobs:
            rl.free()
    return dict(sec_per_locus_iter=one, kind="reference" if use_ref else "port", cores=1, sampled_loci=len(per_locus),
                repeats=reps, seconds=time.time() - t_start,
                all_cores=dict(sec_per_locus_iter=all_sec_per_locus_iter, workers=workers, host_logical_cores=cores,
                               repeats=reps_all, sampled_loci=len(jobs), tried=tried))


SIM_CTL = """seed = 12345
seqfile = syn.txt
Imapfile = syn.Imap.txt
species&tree = 4  A B C D
                  1 1 1 1
                  (((A #0.002, B #0.002):0.001 #0.002, C #0.002):0.002 #0.002, D #0.002):0.003 #0.002;
phase = 0 0 0 0
loci&length = {nloci} {sites}
clock = 1
locusrate = 0
model = 0
"""
A00_CTL = """seed = 1
seqfile = syn.txt
Imapfile = syn.Imap.txt
jobname = out
speciesdelimitation = 0
speciestree = 0
species&tree = 4  A B C D
                  1 1 1 1
                  (((A, B), C), D);
phase = 0 0 0 0
usedata = 1
nloci = {nloci}
model = jc69
cleandata = 0
thetaprior = gamma 2 1000
tauprior = gamma 2 500
finetune = 1
print = 1 0 0 0
burnin = 0
sampfreq = 1
nsample = {nsample}
{threads}
"""


def bpp_program_baseline(nloci, sites, threads_list, reps=1, budget_s=75.0, chain_samples=0, long_short=(100, 700)):
    """The unmodified reference PROGRAM (oracle/_ref/bpp, built in place from /root/reference) on this box's host
    cores: data from its own simulator, A00 JC69, whole MCMC iterations/s from the differential wall time of a short
    and a long run (start-up — reading and compressing 10 000 loci, the first likelihoods — cancels), per thread count:
    median and spread of `reps` measurements.  The long run has 800 iterations more than the short one with several
    threads (>= 10 s of MCMC on this box; round 2's 80-iteration differentials scattered by 2x), 130 more with one.
    north_star's comparison is with the multi-thread AVX2 program: the best thread count is what counts."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    if not os.path.exists(O.REF_BIN):
        return None
    out = {}
    t_start = time.time()
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "sim.ctl"), "w").write(SIM_CTL.format(nloci=nloci, sites=sites))
        subprocess.run([O.REF_BIN, "--simulate", "sim.ctl"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=300)

        def wall(ns, tl):
            open(os.path.join(d, "a00.ctl"), "w").write(A00_CTL.format(nloci=nloci, nsample=ns, threads=tl))
            t0 = time.perf_counter()
            subprocess.run([O.REF_BIN, "--cfile", "a00.ctl"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=900)
            return time.perf_counter() - t0

        for th in threads_list:
            tl = f"threads = {th} 1 1" if th > 1 else ""
            n1, n2 = long_short if th > 1 else (10, 70)
            rates = []
            for _ in range(reps if th > 1 else 1):
                if time.time() - t_start > budget_s and rates:
                    break
                t1, t2 = wall(n1, tl), wall(n2, tl)
                rates.append((n2 - n1) / max(t2 - t1, 1e-9) * nloci / 10000.0)
            out[th] = dict(median=round(float(np.median(rates)), 2), min=round(min(rates), 2), max=round(max(rates), 2),
                           runs=len(rates), iterations=f"{n2} vs {n1}")
        # ---- one chain of the program at its best thread count WITH its burn-in (finetune = 1: BPP adjusts its step
        # lengths there), for the statistical efficiency: the trace of tau_root / theta_root and the tuned step lengths
        if chain_samples:
            try:
                best = max((k for k in out if k > 1), key=lambda k: out[k]["median"], default=1)
                tl = f"threads = {best} 1 1" if best > 1 else ""
                ctl = A00_CTL.format(nloci=nloci, nsample=chain_samples, threads=tl).replace("burnin = 0", "burnin = 400")
                open(os.path.join(d, "a00.ctl"), "w").write(ctl)
                r = subprocess.run([O.REF_BIN, "--cfile", "a00.ctl"], cwd=d, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                   timeout=900, text=True)
                import re
                ft = re.findall(r"finetune = 1 Gage:(\S+) Gspr:(\S+) th1:(\S+) th2:(\S+) tau:(\S+) mix:([0-9.eE+-]+)", r.stdout)
                rows = [ln.split() for ln in open(os.path.join(d, "out.mcmc.txt")).read().splitlines()]
                head, rows = rows[0], rows[1:]
                col = {h.split(":")[0] + ":" + h.split(":")[1]: i for i, h in enumerate(head) if ":" in h}
                tr = {k: [float(rw[i]) for rw in rows] for k, i in col.items()}
                out["chain"] = dict(threads=best, samples=len(rows), burnin=400,
                                    finetune=dict(zip(("gage", "gspr", "th1", "th2", "tau", "mix"), map(float, ft[-1]))) if ft else None,
                                    trace=tr)
            except Exception as ex:       # noqa: BLE001
                out["chain"] = dict(error=str(ex)[:200])
    return out


def ess(x):
    """effective sample size of a trace: n / (1 + 2 sum of autocorrelations), Geyer's initial positive sequence"""
    x = np.asarray(x, dtype=float)
    n = len(x)
    x = x - x.mean()
    if n < 16 or not np.any(x):
        return float(n)
    f = np.fft.rfft(x, 2 * n)
    acf = np.fft.irfft(f * np.conj(f))[:n]
    acf = acf / acf[0]
    tau = -1.0
    for k in range(n // 2):
        g = acf[2 * k] + acf[2 * k + 1]
        if g <= 0:
            break
        tau += 2.0 * g
    return float(n / max(tau, 1e-9))


def ess_batches(x, batches=5):
    """ESS per iteration as the mean over consecutive batches of the trace, with its standard error"""
    x = np.asarray(x, dtype=float)
    per = [ess(b) / len(b) for b in np.array_split(x, batches)]
    return dict(ess_per_iteration=round(float(np.mean(per)), 4), ess_per_iteration_se=round(float(np.std(per, ddof=1) / np.sqrt(len(per))), 4))


def run_efficiency(eng, cfg, data, loci, chain, prog_rate, steps_hint=4000):
    """BPP's own move kernel on the device (bpa_sampler_set_proposal_kernel: Bactrian-Laplace windows, legacy_rndu) with the
    step lengths the program's burn-in arrived at, priors of the program's control file: effective samples per second of
    tau_root and theta_root next to the program's (equal statistical work per iteration by construction)"""
    import bpp_amd
    from bpp_amd import synth
    ft = chain["finetune"]
    S = cfg["taxa"]

    def chain_on_device(program):
        smp = bpp_amd.Sampler(eng, make_loci(eng, data), data, seed=11)
        parent, tau, theta = synth.species_tree_arrays(cfg["taxa"])
        smp.set_species_tree(parent, tau, theta)
        smp.set_tau_prior(2.0, 500.0)                                  # A00_CTL: tauprior = gamma 2 500
        if program:
            smp.set_proposal_kernel(1)
            smp.set_program_moves(True, 0.1)                           # THETA / TAU / MIX as the program runs them (bpp.c:650, 618, 581)
            smp.set_theta_prior(2.0, 1000.0, 0.001)                    #          thetaprior = gamma 2 1000; the program's default step lengths
            smp.set_finetune(5.0, 0.001, 0.001, 0.3)                   #          (bpp.c:530-549): its burn-in rule tunes them below
        else:
            scale = 1.0 / math.sqrt(len(data))                         # (the headline section's step lengths)
            smp.set_theta_prior(2.0, 1000.0, 0.008 * scale)
            smp.set_finetune(0.004, 0.004, 0.004 * scale, 0.6 * scale)
        smp.initialize()
        own_ft = smp.burnin(400) if program else None                  # (the program's chain: burnin = 400 as well)
        smp.iterate(500)
        eng.synchronize()
        t0 = time.perf_counter()
        smp.iterate(3000)
        eng.synchronize()
        rate = 3000 / (time.perf_counter() - t0)
        tr_tau, tr_theta = [], []
        for _ in range(steps_hint):
            smp.iterate(1)
            tr_tau.append(smp.taus()[-1]); tr_theta.append(smp.thetas()[-1])
        sm = smp.summary()
        gibbs = smp.gibbs_counters()
        smp.close()
        tr_tau, tr_theta = np.round(tr_tau, 6), np.round(tr_theta, 6)       # (the program prints 6 decimals)
        d_ = dict(iterations_per_s=round(rate, 1), samples=steps_hint, acceptance=round(sm["accepted"] / max(sm["proposals"], 1), 3),
                  tau_root=dict(mean=float(np.mean(tr_tau)), sd=float(np.std(tr_tau)), **ess_batches(tr_tau)),
                  theta_root=dict(mean=float(np.mean(tr_theta)), sd=float(np.std(tr_theta)), **ess_batches(tr_theta)))
        if program:
            d_["theta_gibbs_draws"] = dict(proposed=gibbs[0], accepted=gibbs[1])
            d_["step_lengths_after_burnin"] = {k: float(f"{v:.4g}") for k, v in own_ft.items()}
        return d_
    dev = chain_on_device(True)
    dev_uniform = chain_on_device(False)
    root = f"{S + 1}"
    p_tau, p_theta = chain["trace"].get("tau:" + root), chain["trace"].get("theta:" + root)
    prog = dict(iterations_per_s=prog_rate, samples=chain["samples"], threads=chain["threads"],
                tau_root=dict(mean=float(np.mean(p_tau)), sd=float(np.std(p_tau)), **ess_batches(p_tau)),
                theta_root=dict(mean=float(np.mean(p_theta)), sd=float(np.std(p_theta)), **ess_batches(p_theta)))
    for d_ in (dev, dev_uniform, prog):
        for k in ("tau_root", "theta_root"):
            d_[k]["ess_per_s"] = round(d_[k]["ess_per_iteration"] * d_["iterations_per_s"], 2)
    def ratio_it(k):
        a, b = dev[k], prog[k]
        r_ = a["ess_per_iteration"] / max(b["ess_per_iteration"], 1e-12)
        return dict(value=round(r_, 3), se=round(r_ * math.hypot(a["ess_per_iteration_se"] / max(a["ess_per_iteration"], 1e-12),
                                                                 b["ess_per_iteration_se"] / max(b["ess_per_iteration"], 1e-12)), 3))
    return dict(device=dev, device_uniform_kernel=dev_uniform, reference_program=prog, step_lengths=ft,
                ratio_ess_per_iteration=dict(tau_root=ratio_it("tau_root"), theta_root=ratio_it("theta_root"),
                                             longer_runs="profiles/r5/ess_*.json: 1 000 loci x 100 000 samples and 10 000 loci x 20 000 samples, both chains on the SAME data (tools/ess_device_vs_program.py)"),
                ratio_ess_per_s=dict(tau_root=round(dev["tau_root"]["ess_per_s"] / max(prog["tau_root"]["ess_per_s"], 1e-9), 1),
                                     theta_root=round(dev["theta_root"]["ess_per_s"] / max(prog["theta_root"]["ess_per_s"], 1e-9), 1)),
                note_uniform_kernel="device_uniform_kernel: the headline section's sampler (our 64-bit streams, uniform windows, sliding-window THETA, "
                     "TAU and MIX without theta re-draws; step lengths 0.004, 0.004 and 0.008, 0.004, 0.6 over sqrt(loci)) on the program's priors: "
                     "the same work per iteration, fewer effective samples per iteration than the program's moves",
                note="device: the device sampler with BPP's own proposal kernel (Bactrian-Laplace m = 0.9 windows, legacy_rndu, acceptance number "
                     "drawn only when needed; and the program's THETA / TAU / MIX (bpa_sampler_set_program_moves: sliding window 1 time in 10 and the metropolized "
                     "Gibbs draw otherwise, thetas re-drawn inside the rubber-band and the mixing proposals): the program's iteration move for move, "
                     "equal effective samples per iteration on the same data (tools/ess_compare.py), "
                     "and the step lengths the program's burn-in (finetune = 1) tuned itself to, the program's priors; "
                     "ESS = n / (1 + 2 sum of autocorrelations) (initial positive sequence) of the traces of tau_root and theta_root, one sample per "
                     "iteration; the program on its own simulated 10000-locus set, the device on the bench's synthetic set of the same model and "
                     "size (the program's trace prints 6 decimals)")


def run_config1(eng, iters=12):
    """BASELINE configs[0] (examples/frogs A00 JC69: 5 loci of unphased diploid sequences, 42-60 tips after phasing, from the
    files the reference ships: tests/golden/frogs) as real MCMC on the device — the big-tree sampler (csrc/bigsampler.hpp)"""
    import bpp_amd
    from bpp_amd import seqio, synth
    g = os.path.join(ROOT, "tests", "golden", "frogs")
    recs = seqio.load_dataset(os.path.join(g, "frogs.txt"), os.path.join(g, "frogs.Imap.txt"), ["K", "C", "L", "H"], [1, 1, 1, 1], model="jc69")
    parent = [4, 4, 5, 6, 5, 6, -1]                       # K C L H | KC KCL root: (((K, C), L), H) of the example's control file
    tau0 = [0.0] * 4 + [0.01, 0.02, 0.03]
    thetas = [0.02] * 7
    rng = np.random.default_rng(9)
    data = []
    for r in recs:
        left, right, times, root = synth.msc_start_tree(r["species"], parent, tau0, thetas, rng)
        data.append(dict(seqs=r["seqs"], weights=r.get("weights", np.ones(len(r["seqs"][0]))), left=left, right=right, times=times, root=root,
                         states=4, rate_cats=1, model="jc69", rates=np.ones(1)))
    smp = bpp_amd.Sampler(eng, [seqio.make_locus(eng, r) for r in recs], data, seed=1)
    smp.set_species_tree(parent, tau0, thetas)
    for i, r in enumerate(recs):
        smp.set_tip_species(i, r["species"])
    smp.set_tau_prior(3.0, 100.0)
    smp.set_theta_prior(3.0, 150.0, 0.003)
    smp.set_finetune(0.004, 0.004, 0.002, 0.1)
    smp.initialize()
    lnl0 = smp.summary()["total_lnl"]
    smp.iterate(2)
    eng.synchronize()
    l0 = smp.summary()["launches"]
    t0 = time.perf_counter()
    smp.iterate(iters)
    eng.synchronize()
    dt = time.perf_counter() - t0
    sm = smp.summary()
    out = dict(iterations_per_s=round(iters / dt, 2), ms_per_iteration=round(1e3 * dt / iters, 3), iterations=iters, loci=len(recs),
               tips=[len(r["seqs"]) for r in recs], patterns=[len(r["seqs"][0]) for r in recs], implementation=smp.kind(),
               launches_per_iteration=round((sm["launches"] - l0 - 1) / iters, 1), start_lnl=round(lnl0, 6),
               acceptance=round(sm["accepted"] / max(sm["proposals"], 1), 3),
               reference_program="BASELINE.md: the unmodified program on this example, one thread: ~270 iterations/s on the survey container "
                                 "(690 on this box's host: tools/bench_bpp_hip.py, round 2)",
               note="five loci are no work for a GPU: ~107 us of every ~130 us step is the proposal code of ONE lane per locus (measured "
                    "inside a one-launch-per-call variant that ran at the same 42 it/s: profiles/r4/c1_resident_phases.txt), not the 730 "
                    "launches; a host core does such a step in 1.6 us.  The lock-step batched form pays off from thousands of loci on "
                    "(configs 2-4); here it shows that the device path covers the reference's own example")
    smp.close()
    return out


# ------------------------------------------------------- the unmodified program on the other configs' data (same box) ---
def _phylip_from_data(data, names):
    """bpp_amd.synth loci (patterns + weights) as the reference's sequential PHYLIP + Imap: every pattern written `weight` times"""
    out = []
    for d in data:
        cols = [j for j, w in enumerate(d["weights"]) for _ in range(int(w))]
        out.append(f"\n{len(names)} {len(cols)}\n\n")
        for t, nm in enumerate(names):
            out.append(f"{nm}^{nm.lower()}1        " + "".join(d["seqs"][t][j] for j in cols) + "\n")
        out.append("\n")
    return {"syn.txt": "".join(out), "syn.Imap.txt": "".join(f"{c.lower()}1\t{c}\n" for c in names)}


def program_rate(files, ctl, n1, n2, threads_list, scale_to, nloci, budget_s=60.0):
    """whole MCMC iterations/s of oracle/_ref/bpp (the unmodified reference program, AVX2) on the given input files and
    control file (fields {nsample}, {threads}): differential wall time of an n1- and an n2-iteration run per thread
    count, scaled to `scale_to` loci.  Returns the best thread count's rate and what was tried."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    if not os.path.exists(O.REF_BIN):
        return None
    tried = {}
    t_start = time.time()
    with tempfile.TemporaryDirectory() as d:
        for name, text in files.items():
            open(os.path.join(d, name), "w").write(text)

        def wall(ns, th):
            open(os.path.join(d, "a00.ctl"), "w").write(ctl.format(nsample=ns, threads=(f"threads = {th} 1 1" if th > 1 else "")))
            t0 = time.perf_counter()
            subprocess.run([O.REF_BIN, "--cfile", "a00.ctl"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
            return time.perf_counter() - t0
        for th in threads_list:
            if tried and time.time() - t_start > budget_s:
                break
            # (the short run twice, the faster one counts: a hiccup in it inflates the rate of a short differential)
            t1, t2 = min(wall(n1, th), wall(n1, th)), wall(n2, th)
            tried[th] = round((n2 - n1) / max(t2 - t1, 1e-9) * nloci / scale_to, 3)
    best = max(tried, key=lambda k: tried[k])
    return dict(value=tried[best], unit=f"iterations/s (whole A00 MCMC iterations of the unmodified program, scaled from {nloci} to {scale_to} loci)",
                cores=best, kind="reference", threads_tried={str(k): v for k, v in tried.items()}, host_cpu_quota=cpu_quota(),
                sample=f"oracle/_ref/bpp (AVX2) on {nloci} loci of this config's data, {n2} vs {n1} iterations differential, burnin 0, finetune = 1")


def other_config_cpu_baseline(key, data):
    """`other_configs.<key>.cpu_baseline`: the unmodified program at its best thread count on (a share of) the config's own data"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bpphip as B
    q = int(cpu_quota() or os.cpu_count() or 1)
    threads = sorted({max(2, q), max(2, 2 * q)})
    if key == "c3":
        n = min(len(data), 1000)
        ctl = B.A00_CTL.format(species=B.SPECIES8, phase="0 0 0 0 0 0 0 0", nloci=n, model="gtr", alpha="alphaprior = 1 1 4", taub=300,
                               burnin=0, sampfreq=1, nsample="{nsample}", extra="{threads}")
        return program_rate(_phylip_from_data(data[:n], "ABCDEFGH"), ctl, 20, 400, threads, len(data), n, budget_s=25.0)      # (>= 3 s of MCMC between the two runs: a 30-iteration differential once read 62 it/s for 11)
    if key == "c4":
        n = min(len(data), 128)
        sp6 = "6  A B C D E F\n                  1 1 1 1 1 1\n                  ((((A, B), C), (D, E)), F);"
        ctl = B.A00_CTL.format(species=sp6, phase="0 0 0 0 0 0", nloci=n, model="lg", alpha="alphaprior = 1 1 4", taub=40,
                               burnin=0, sampfreq=1, nsample="{nsample}", extra="{threads}").replace("thetaprior = gamma 2 1000", "thetaprior = gamma 2 100")
        return program_rate(_phylip_from_data(data[:n], "ABCDEF"), ctl, 5, 100, threads, len(data), n, budget_s=25.0)
    if key == "c5":
        g = os.path.join(ROOT, "tests", "golden", "anopheles")
        files = {"loci_realign.txt": open(os.path.join(g, "loci_realign.txt")).read(), "Imap.txt": open(os.path.join(g, "Imap.txt")).read()}
        ctl = B.ANOPHELES_CTL.format(tree=B.ANOPHELES_MSCI_TREE, phiprior="phiprior = 1 1", burnin=0, sampfreq=1, nsample="{nsample}", extra="{threads}")
        return program_rate(files, ctl, 100, 600, sorted({1, min(4, q)}), 100, 100)
    if key == "c1":
        g = os.path.join(ROOT, "tests", "golden", "frogs")
        files = {"frogs.txt": open(os.path.join(g, "frogs.txt")).read(), "frogs.Imap.txt": open(os.path.join(g, "frogs.Imap.txt")).read()}
        ctl = B.FROGS_CTL.format(burnin=0, sampfreq=1, nsample="{nsample}", extra="{threads}")
        return program_rate(files, ctl, 200, 1700, [1], 5, 5)
    return None


def run_config5(eng, iters=40):
    """BASELINE configs[4] (examples/anopheles: 100 loci x 12 sequences, 6 species with two sequences each, JC69,
    cleandata = 1; priors and step lengths of anopheles-bpp-msci.ctl) as real MCMC on the device: the generic sampler
    (12 tips), the MSC on the control file's species tree (R,((C,G),((A,Q),L))) — the introgression model's own species-tree
    proposals are host control of the reference and out of scope"""
    import bpp_amd
    from bpp_amd import seqio, synth
    g = os.path.join(ROOT, "tests", "golden", "anopheles")
    species = ["G", "C", "R", "L", "A", "Q"]
    recs = seqio.load_dataset(os.path.join(g, "loci_realign.txt"), os.path.join(g, "Imap.txt"), species, None, model="jc69", cleandata=True)
    parent = [6, 6, 10, 8, 7, 7, 9, 8, 9, 10, -1]
    tau0 = [0.0] * 6 + [0.004, 0.004, 0.008, 0.012, 0.016]
    thetas = [0.02] * 11
    rng = np.random.default_rng(77)
    data = []
    for r in recs:
        left, right, times, root = synth.msc_start_tree(r["species"], parent, tau0, thetas, rng)
        data.append(dict(seqs=r["seqs"], weights=r["weights"], left=left, right=right, times=times, root=root, states=4, rate_cats=1, model="jc69", rates=np.ones(1)))
    smp = bpp_amd.Sampler(eng, [seqio.make_locus(eng, r) for r in recs], data, seed=1)
    smp.set_species_tree(parent, tau0, thetas)
    for i, r in enumerate(recs):
        smp.set_tip_species(i, r["species"])
    smp.set_tau_prior(2.0, 10.0)
    # BPP's own iteration (see run_sampler): the control file's priors, the program's default step lengths and its burn-in rule
    smp.set_proposal_kernel(1)
    smp.set_program_moves(True, 0.1)
    smp.set_theta_prior(2.0, 100.0, 0.001)
    smp.set_finetune(5.0, 0.001, 0.001, 0.3)
    smp.initialize()
    lnl0 = smp.summary()["total_lnl"]
    ft5 = smp.burnin(400)
    smp.iterate(3)
    eng.synchronize()
    l0 = smp.summary()["launches"]
    t0 = time.perf_counter()
    smp.iterate(iters)
    eng.synchronize()
    dt = time.perf_counter() - t0
    sm = smp.summary()
    out = dict(iterations_per_s=round(iters / dt, 2), ms_per_iteration=round(1e3 * dt / iters, 3), iterations=iters, loci=len(recs),
               tips=12, patterns_mean=round(float(np.mean([len(r["seqs"][0]) for r in recs])), 1), implementation=smp.kind(),
               launches_per_iteration=round((sm["launches"] - l0 - 1) / iters, 1), start_lnl=round(lnl0, 6),
               acceptance=round(sm["accepted"] / max(sm["proposals"], 1), 3), moves_short="program",
               step_lengths_after_burnin={k: float(f"{v:.4g}") for k, v in ft5.items()},
               note="100 loci x 12 tips: 33 per-locus proposals + 11 thetas + 5 taus + mixing per iteration, every step a handful of "
                    "launches over 100 lanes of work — a plumbing / parity configuration (SURVEY 8d), not a throughput one")
    smp.close()
    return out


def run_mixed_set(data, nodd=10, iters=200):
    """config 2's loci with a few loci of another kind among them (GTR+Gamma4, 4 rate categories): ONE sampler — the composite
    (csrc/composite.hpp): the JC69 loci on the persistent kernel's part, the others on the generic part, stepped together —
    next to what such a set cost before round 4 (refused; or, forced, all 10 000 loci on the generic path)"""
    import bpp_amd
    from bpp_amd import synth
    eng = bpp_amd.Engine(0, None)
    odd = synth.make_dataset(nodd, 1000, 4, "gtr", 4, seed=99)
    mixed = list(data[:len(data) - nodd]) + odd
    smp = bpp_amd.Sampler(eng, make_loci(eng, mixed), mixed, seed=3)
    par, tau, theta = synth.species_tree_arrays(4)
    smp.set_species_tree(par, tau, theta)
    smp.set_theta_prior(2.0, 1000.0, 8e-5); smp.set_tau_prior(2.0, 500.0)
    smp.set_finetune(0.004, 0.004, 4e-5, 0.006)
    smp.initialize(); smp.iterate(20); eng.synchronize()
    t0 = time.perf_counter(); smp.iterate(iters); eng.synchronize(); dt = time.perf_counter() - t0
    sm = smp.summary()
    out = dict(loci=f"{len(data) - nodd} config-2 JC69 loci + {nodd} GTR+G4 loci", implementation=smp.kind(), iterations_per_s=round(iters / dt, 1),
               ms_per_iteration=round(1e3 * dt / iters, 4), acceptance=round(sm["accepted"] / max(sm["proposals"], 1), 3), moves="the library's uniform-window moves",
               note="the generic part's per-step launches set the pace (a few loci cost a step's latency, not the other loci's throughput)")
    smp.close(); eng.close()
    return out


def scale_projection(key, cfg, data, args, shares=(2, 4, 8)):
    """What ONE rank of an N-GPU strong-scaling run of this config works on, measured on this one GPU: the loci the reference's
    zig-zag deal (threads.c:265-353, bpp_amd/shard.py) gives rank 0 of N, through the device-resident sampler.  iterations/s of
    that share = the upper bound of the N-GPU rate of the WHOLE data set (before the per-step exchange over xGMI)."""
    import bpp_amd
    from bpp_amd import shard
    out = {}
    for n in shares:
        try:
            mine = shard.partition([len(d["seqs"]) * len(d["weights"]) for d in data], n)[0]
            sub = [data[i] for i in mine]
            e = bpp_amd.Engine(0, None)
            a2 = argparse.Namespace(**vars(args))
            a2.loci = len(sub)
            r = run_sampler(e, cfg, sub, make_loci(e, sub), a2, None, 0, args.projection_iters if cfg["model"] != "jc69" else 40, 3 if cfg["model"] != "jc69" else 5)
            out[str(n)] = dict(loci_on_rank0=len(sub), iterations_per_s=r["iterations_per_s"], ms_per_iteration=r["ms_per_iteration"], implementation_kind=r.get("kind"))
            e.close()
        except Exception as ex:       # noqa: BLE001
            out[str(n)] = dict(error=str(ex)[:200])
    return out


# ------------------------------------------------------------------------------------------------ the workload ---
def make_loci(eng, data):
    import bpp_amd
    loci = []
    for d in data:
        S, R = d["states"], d["rate_cats"]
        tips, sites = len(d["seqs"]), len(d["seqs"][0])
        inner, edges = tips - 1, 2 * tips - 2
        mdl = {"jc69": bpp_amd.MODEL_JC69, "gtr": bpp_amd.MODEL_GTR, "lg": bpp_amd.MODEL_LG}[d["model"]]
        loc = bpp_amd.Locus(eng, bpp_amd.DATA_DNA if S == 4 else bpp_amd.DATA_AA, mdl, tips, 2 * inner, S,
                            sites, 1, 2 * edges, R, 0)
        for i, s in enumerate(d["seqs"]):
            loc.set_tip_states(i, s)
        loc.set_pattern_weights(d["weights"])
        if d["model"] != "jc69":
            loc.set_frequencies(0, d["freqs"])
            loc.set_subst_params(0, d["exch"])
        loc.set_category_rates(d["rates"])
        loci.append(loc)
    return loci


def run_host_control(eng, cfg, data, threads, iters=30, warm=3):
    """north_star's literal architecture: MCMC control on the host in C (a00_driver.c: proposals, MSC density,
    bookkeeping, decisions), every proposal step ONE batched bpa_batch_evaluate on the GPU, per-locus lnL back over PCIe.
    Own loci on the same engine (the driver toggles their buffers)."""
    # the worker threads stay where they are (without it the same run is 25 % slower on this 256-logical-core host);
    # libgomp reads this when it is loaded, i.e. with the host library below
    os.environ.setdefault("OMP_PROC_BIND", "close")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostdrv
    from bpp_amd import synth
    cohorts = not os.environ.get("A00_NO_COHORTS") and len(data) >= 64
    if cohorts:
        # two cohorts of loci on two engines of this GPU (a00_set_cohorts): one cohort's step is proposed and marshalled
        # while the other's launch runs — the same trajectory as one engine (tests/test_gpu_host_driver.py)
        import bpp_amd
        eng2 = bpp_amd.Engine(0)
        split = len(data) // 2
        loci = make_loci(eng, data[:split]) + make_loci(eng2, data[split:])
        g = hostdrv.hip_driver_cohorts([eng, eng2], loci, data, split, seed=1)
    else:
        loci = make_loci(eng, data)
        g = hostdrv.hip_driver(eng, loci, data, seed=1)
    g.set_threads(threads)
    parent, tau, theta = synth.species_tree_arrays(cfg["taxa"])
    g.set_species_tree(parent, tau, theta)
    g.set_tau_prior(3.0, 3.0 / tau[-1])
    g.set_theta_prior(2.0, 2.0 / theta[0], 0.5 * theta[0])
    g.initialize()
    for _ in range(warm):
        g.iterate()
    # five timed blocks: the host side shares its CPUs with whatever else runs on the box, the median says what the path does
    rates = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(iters):
            g.iterate()
        rates.append(iters / (time.perf_counter() - t0))
    rate = float(np.median(rates))
    p, a, st = g.counters()
    g.close()
    if cohorts:
        eng2.close()
    return dict(cohorts=2 if cohorts else 1, iterations_per_s=round(rate, 2), ms_per_iteration=round(1e3 / rate, 3), iterations=iters, blocks=len(rates),
                spread=dict(min=round(min(rates), 2), max=round(max(rates), 2)),
                host_threads=threads, launches_per_iteration=round(st / (5 * iters + warm), 1), acceptance=round(a / max(p, 1), 3),
                note="host MCMC control in C (csrc/host/a00_driver.c, per-locus loops on OpenMP worker threads; the trajectory does "
                     "not depend on their number), one batched launch + one synchronisation + 80 KB D2H per proposal step: the "
                     "PCIe-inclusive rate of the drop-in architecture with the control left on the host"
                     + ("; the loci in two cohorts on two engines of the GPU: a per-locus step of one cohort is proposed and marshalled while "
                        "the other's launch runs (a00_set_cohorts; `launches_per_iteration` counts both cohorts' launches)" if cohorts else ""))


def dominant_kernel(cfg, one_gpu=True):
    if cfg["model"] == "jc69":
        if os.environ.get("BPA_NO_CHAIN"):
            return "step_jc69_v2_kernel<256>"
        # one GPU: the whole iteration is one chain launch; several: the chain breaks at the steps whose sums are exchanged
        return "step_jc69_v2_chain_kernel<256>" if one_gpu else "step_jc69_v2_chain_kernel<256> + step_jc69_v2_kernel<256>"
    if cfg["model"] == "gtr":
        return "step_s4_klane_v2_kernel<256,false>" if os.environ.get("BPA_KLANE_V2") else "step_s4_klane_v3_kernel<256,false>"
    k = os.environ.get("BPA_S20_KERNEL", "wave")
    return {"wave": "partials_lnl_wave20_kernel<20,true,2,1>", "wave2": "partials_lnl_wave20_kernel<20,true,1,2>", "pipe": "partials_lnl_pipe20_kernel<20,true,2>",
            "pipemfma": "partials_lnl_pipemfma20_kernel<false,2>"}.get(k, f"20-state kernel `{k}`")


_PROFILE_US = {}       # (config, kernel) -> mean duration (us) of the full-batch launches in the committed profile's trace pass


def traffic_from_profiles(config, kernel):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes committed under profiles/ (tools/profile_cfg.sh on
    this same command): 2*FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, both in KB"""
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"profile_{config}.json")))
        pm = json.load(open(cands[-1]))["pmc_per_dispatch"]
        want = kernel.split(" + ")[0].replace(" ", "")
        key = [k for k in pm if want.rstrip(">") in k.replace(" ", "") and "FETCH_SIZE" in pm[k]]
        key = (key or [k for k in pm if want.split("<")[0] in k and "FETCH_SIZE" in pm[k]])[0]
        prof = json.load(open(cands[-1]))
        same = (prof.get("moved_same_file") or {}).get(key)
        if same:
            # the profile's own duration of the SAME full-batch launches its counters were averaged over (tools/profile_cfg.sh buckets
            # a kernel's dispatches by grid size): `frac_pmc` divides bytes and time of one file
            _PROFILE_US[(config, kernel)] = same["mean_us"]
        return (round((2 * pm[key]["FETCH_SIZE"]["mean"] + pm[key]["WRITE_SIZE"]["mean"]) * 1024),
                "traffic_from_committed_profile: " + os.path.relpath(cands[-1], ROOT) + f" ({key.split('(')[0]}; separate --pmc passes over the kernel's full-batch "
                "dispatches; FETCH_SIZE doubled per the gfx950 note)")
    except Exception:
        return None, None


def counters_from_profiles(config, kernel):
    """what binds a kernel that HBM does not, from the committed profile's SQ passes (full-batch dispatches): the share of its
    wave-cycles spent waiting, issuing VALU / LDS instructions, LDS bank conflicts per active LDS cycle, scalar per vector
    instruction — ratios of per-dispatch counter means, labelled as the builder-run figures they are"""
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"profile_{config}.json")))
        pm = json.load(open(cands[-1]))["pmc_per_dispatch"]
        want = kernel.split(" + ")[0].replace(" ", "")
        key = [k for k in pm if want.rstrip(">") in k.replace(" ", "") and "SQ_WAVE_CYCLES" in pm[k]]
        key = (key or [k for k in pm if want.split("<")[0] in k and "SQ_WAVE_CYCLES" in pm[k]])
        # (the persistent kernel: its program-moves instance)
        key = ([k for k in key if "true, true" in k] or key)[0]
        c = {n: v["mean"] for n, v in pm[key].items()}
        wc = c["SQ_WAVE_CYCLES"]
        out = dict(wait_any=round(c["SQ_WAIT_ANY"] / wc, 3), valu_active=round(c["SQ_ACTIVE_INST_VALU"] / wc, 3),
                   lds_active=round(c.get("SQ_ACTIVE_INST_LDS", 0.0) / wc, 3))
        if c.get("SQ_LDS_IDX_ACTIVE"):
            out["lds_bank_conflict"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3)
        if c.get("SQ_INSTS_VALU"):
            out["salu_per_valu"] = round(c.get("SQ_INSTS_SALU", 0.0) / c["SQ_INSTS_VALU"], 3)
        out["of"] = "wave-cycles (rocprofv3 SQ passes of " + os.path.relpath(cands[-1], ROOT) + ")"
        return out
    except Exception:       # noqa: BLE001
        return None


def add_honest_fracs(r, codes_ratio=None, flops_per_launch=None, codes_note=None):
    """next to SURVEY 8d's `frac`: `frac_codes` (the same launches priced as the kernels hold the data: tip children as codes,
    forwarded parents not re-read — bpa_plan_work_codes) and `flops_frac` (K1's Np R (4S^2 - S) flops / the kernel's time / the
    FP64 peak); a `frac` above what HBM can stream is flagged as the accounting artefact it is"""
    if not isinstance(r, dict):
        return r
    if codes_ratio:
        r["codes_bytes_per_launch"] = round(r["algorithmic_bytes_per_launch"] * codes_ratio)
        r["frac_codes"] = round(r["frac"] * codes_ratio, 5)
        if codes_note:
            r["frac_codes_note"] = codes_note
    if flops_per_launch and r.get("avg_kernel_us"):
        r["flops_per_launch"] = round(flops_per_launch)
        r["tflops"] = round(flops_per_launch / (r["avg_kernel_us"] * 1e-6) / 1e12, 3)
        r["flops_frac"] = round(r["tflops"] / FP64_PEAK_TFLOPS, 5)
        r["flops_peak"] = FP64_PEAK_TFLOPS
    if r.get("frac", 0) > HBM_ACHIEVABLE_FRAC:
        r["warning"] = (f"frac {r['frac']} > {HBM_ACHIEVABLE_FRAC}: more than HBM can stream; SURVEY 8d's bytes charge a tip child as a one-hot "
                        "CLV the kernel reads as codes - read frac_codes / frac_pmc")
    return r


def profile_head(config):
    """the source hash (tools/src_hash.py) the committed profile of `config` was taken on, and whether it is this tree's"""
    try:
        import glob
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from src_hash import src_hash
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"profile_{config}.json")))
        sha = json.load(open(cands[-1])).get("kernels_sha")
        return f"{sha} ({'= this build' if sha == src_hash() else 'NOT this build: ' + src_hash()})"
    except Exception:       # noqa: BLE001
        return None


def add_frac_pmc(r):
    """`frac_pmc`: HBM bytes actually MOVED per launch (the PMC passes under profiles/) / the launch's duration / the HBM peak —
    next to `frac`, which prices the algorithmic bytes"""
    if isinstance(r, dict) and r.get("traffic") and r.get("avg_kernel_us"):
        r["traffic_from_committed_profile"] = True      # (a builder-run rocprofv3 PMC number inside a driver-run line: profiles/)
        # bytes and time of the SAME launches: the profile's own mean duration of the full-batch dispatches its counters were
        # averaged over, when it has one (round 6); else this run's event time (a persistent launch: its length is the run's)
        us = next((v for (c_, k_), v in _PROFILE_US.items() if k_ == r.get("kernel")), None) if (r.get("launches", 0) or 0) > 1 else None
        r["frac_pmc_time_us"] = us if us else r["avg_kernel_us"]
        r["frac_pmc_basis"] = "committed profile: bytes and duration of its full-batch dispatches" if us else "committed profile's bytes / this run's event time"
        r["moved_GBps"] = round(r["traffic"] / (r["frac_pmc_time_us"] * 1e-6) / 1e9, 2)
        r["frac_pmc"] = round(r["moved_GBps"] / r["peak"], 5)
        kn = r.get("kernel") or ""
        cfg_ = "c3" if "klane" in kn else "c4" if "20" in kn.split("<")[0] else "c2"
        pmc = counters_from_profiles(cfg_, "iter_kernel" if "iter_kernel" in kn else kn)
        if pmc:
            r["pmc"] = pmc
    return r


class Dist:
    """torch.distributed + the stream the engine and the collectives share"""

    def __init__(self, world, rank, local_rank):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.world, self.rank = torch, dist, world, rank
        # BENCH_DIST_BACKEND=gloo + BENCH_FORCE_DEVICE=0 let the N>1 code path be exercised on a one-GPU box (two ranks
        # sharing device 0); the driver's runs use nccl (= RCCL), one GPU per rank
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if "BENCH_FORCE_DEVICE" in os.environ:
            local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
        self.local_rank = local_rank
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend, rank=rank, world_size=world)
        # ONE explicit torch stream for the engine and the collectives, so that an all-reduce is ordered after the
        # kernel that produced the sum (torch's default stream has handle 0, which the engine would replace)
        self.tstream = torch.cuda.Stream()
        torch.cuda.set_stream(self.tstream)
        self.stream = self.tstream.cuda_stream
        self.p2p = None
        self.tape_p2p = False          # --p2p-sums: the tape's exchanges over the p2p mailboxes too (default: RCCL)
        self.sampler_exchange = None

    def max(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_int(self, x):
        t = self.torch.tensor([x], dtype=self.torch.int64, device="cuda")
        self.dist.all_reduce(t)
        return int(t.item())

    def sync(self):
        self.torch.cuda.synchronize()
        self.dist.barrier()
        self.torch.cuda.synchronize()

    def setup_p2p(self, eng, n):
        """the one-shot all-reduce over xGMI peer mappings (bpa_p2p_*), used only when its start-up self-test against
        RCCL passes on EVERY rank"""
        import bpp_amd
        torch, dist = self.torch, self.dist
        ok = 1
        p2p = None
        try:
            p2p = bpp_amd.P2P(eng, self.rank, self.world, 512)
            handles = [None] * self.world
            dist.all_gather_object(handles, p2p.handle)
            p2p.connect(handles)
            for k in range(6):
                x = torch.arange(n, dtype=torch.float64, device="cuda") * (0.5 + self.rank) + k + 1e-3 * self.rank
                y = x.clone()
                torch.cuda.current_stream().synchronize()
                p2p.allreduce(x.data_ptr(), x.numel())
                dist.all_reduce(y)
                torch.cuda.synchronize()
                if p2p.status() != 0 or not torch.allclose(x, y, rtol=1e-12, atol=0):
                    ok = 0
                    break
        except Exception as exc:          # no peer mapping on this system: RCCL it is
            log(f"p2p all-reduce unavailable ({exc})")
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int64, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            p2p = None
        elif p2p is not None:
            p2p.set_timeout_ms(1000)      # inside timed regions a lost flag costs 1 s once, then the section is repeated over RCCL
        self.p2p = p2p
        log("sums between ranks: " + ("one-shot p2p all-reduce (self-test against RCCL passed)" if p2p else "RCCL all-reduce"))


def run_tape(eng, cfg, config_key, data, loci, args, D, steps, warmup):
    """the likelihood hot path alone on a resident proposal tape; returns the section's dict"""
    import bpp_amd
    from bpp_amd.schedule import A00Schedule, TreeState
    nloci = len(data)
    rank = D.rank if D else 0
    world = D.world if D else 1
    t0 = time.time()
    trees = [TreeState(d["left"], d["right"], d["times"], d["root"]) for d in data]
    subst = None
    if cfg["model"] == "gtr" and not args.no_subst_proposals:
        # the per-locus frequency / exchangeability / alpha proposals of a GTR+Gamma analysis
        R_ = cfg["rate_cats"]
        subst = dict(freqs=[d["freqs"] for d in data], exch=[d["exch"] for d in data], alpha=[0.5] * nloci, rate_cats=R_,
                     gamma=lambda a, cats: bpp_amd.compute_gamma_cats(a, a, cats) if cats > 1 else np.ones(1))
    sch = A00Schedule(trees, seed=1 + rank, taus=tuple(t * cfg.get("divergence", 1.0) for t in cfg["taus"]), subst=subst)
    init = sch.initial_step()
    iters = [sch.iteration() for _ in range(args.tape_iters)]
    log(f"{config_key} tape: {args.tape_iters} iterations x {len(iters[0])} batched steps in {time.time() - t0:.1f}s")

    sum_buf = D.torch.zeros(4096, dtype=D.torch.float64, device="cuda") if D else None   # room for the per-workgroup partial sums
    sum_parts = []

    def mkplan(st):
        p = bpp_amd.Plan(eng, [loci[i] for i in st.loci], st.mat_off, st.mat_pmatrix, st.mat_length,
                         st.op_off, st.ops, st.root_clv, st.root_scaler)
        if st.global_decision is not None:
            # the sum an all-loci proposal is decided on (and the ranks all-reduce): written by the step kernel as
            # per-workgroup partial sums where the plan runs on the engine's packing (no launch of its own), else the
            # plain total; --sum-launch forces the total as its own launch
            if args.sum_launch:
                p.enable_sum(sum_buf.data_ptr() if sum_buf is not None else None)
            else:
                sum_parts.append(p.enable_partial_sums(sum_buf.data_ptr() if sum_buf is not None else None,
                                                       sum_buf.numel() if sum_buf is not None else 0))
        return p

    p_init = mkplan(init)
    plans = [[mkplan(st) for st in it] for it in iters]
    sum_view = sum_buf[:1] if sum_buf is not None else None
    if D and sum_parts:
        # every rank must all-reduce the same number of doubles: the largest count over the ranks
        cnt = D.torch.tensor([max(sum_parts)], dtype=D.torch.int64, device="cuda")
        D.dist.all_reduce(cnt, op=D.dist.ReduceOp.MAX)
        sum_view = sum_buf[:int(cnt.item())]
    p2p = D.p2p if (D and D.tape_p2p and sum_view is not None and sum_view.numel() <= 512) else None

    # parameter installs of the tape, resident in HBM: (which, device address) per step, applied through p_init
    # (which holds every locus) right before the step's launch
    staged = [[[(w, eng.stage(v)) for w, v in st.params] for st in it] for it in iters]
    p_init.launch()
    lnl0 = p_init.lnl()
    log(f"{config_key} start-up lnL (sum over this rank's loci) = {lnl0.sum():.6f}")

    work = [[p.work() for p in it] for it in plans]
    it_pattern_updates = np.mean([sum(w["pattern_updates"] for w in it) for it in work])
    it_node_updates = np.mean([sum(w["node_updates"] for w in it) for it in work])
    it_flops = np.mean([sum(w["flops_partials"] for w in it) for it in work])
    steps_per_iter = float(np.mean([len(it) for it in plans]))
    tot_b = sum(w["bytes_partials"] for it in work for w in it)
    codes_ratio_k1k2 = sum(w["bytes_codes"] for it in work for w in it) / max(tot_b, 1.0)   # (handed to the sampler sections of the same config)
    flops_per_byte = sum(w["flops_partials"] for it in work for w in it) / max(tot_b, 1.0)

    # launch segments: consecutive steps up to and including an all-loci step (TAU/MIX), whose summed lnL is then
    # all-reduced across ranks — the per-proposal reduction of threads.c:544-591.  Inside a segment bpa_plans_launch
    # sends the consecutive per-locus steps out as one chain launch.  One GPU: the whole iteration is one host call.
    segments = []
    for sts, pls, stg in zip(iters, plans, staged):
        segs, cur, pre = [], [], []
        for st, p, installs in zip(sts, pls, stg):
            if installs and cur:
                segs.append((pre, bpp_amd.PlanSequence(cur), False))
                cur, pre = [], []
            if installs:
                pre = installs
            cur.append(p)
            if D is not None and st.global_decision is not None:
                segs.append((pre, bpp_amd.PlanSequence(cur), True))
                cur, pre = [], []
        if cur:
            segs.append((pre, bpp_amd.PlanSequence(cur), False))
        segments.append(segs)
    sum_ptr, sum_n = (sum_buf.data_ptr(), sum_view.numel()) if sum_buf is not None else (None, 0)

    def run_iteration(i):
        if args.host_in_loop:
            for p, installs in zip(plans[i % len(plans)], staged[i % len(plans)]):
                for w, dptr in installs:
                    p_init.set_params_device(w, dptr)
                p.launch()
                p.lnl()                      # sync + 8 B/locus D2H, as host MCMC control would need
            return
        for pre, seq, reduce_after in segments[i % len(segments)]:
            for w, dptr in pre:
                p_init.set_params_device(w, dptr)
            if reduce_after and p2p is not None:
                seq.launch_exchange(p2p, sum_ptr, sum_n)
                continue
            seq.launch()
            if reduce_after:
                D.dist.all_reduce(sum_view)

    sync = D.sync if D else eng.synchronize
    while True:
        for i in range(warmup):
            run_iteration(i)
        sync()
        if not args.no_timing_events:
            eng.enable_timing(True, stride=args.event_stride)
        t0 = time.perf_counter()
        for i in range(steps):
            run_iteration(warmup + i)
        enqueue_s = time.perf_counter() - t0          # host time to enqueue the timed region (GPU still running)
        sync()
        elapsed = time.perf_counter() - t0
        log(f"{config_key} tape: host enqueue {1e3 * enqueue_s / steps:.4f} ms/step of {1e3 * elapsed / steps:.4f} ms/step")
        tm = eng.timing() if not args.no_timing_events else None
        eng.enable_timing(False)
        if p2p is None:
            break
        # a p2p exchange that timed out on ANY rank voids the run: measure again over RCCL
        bad = D.torch.tensor([1 if p2p.status() != 0 else 0], dtype=D.torch.int64, device="cuda")
        D.dist.all_reduce(bad, op=D.dist.ReduceOp.MAX)
        if int(bad.item()) == 0:
            break
        log("p2p all-reduce timed out: repeating the measurement over RCCL")
        p2p = D.p2p = None

    allreduce_check = None
    if D is not None:
        elapsed = D.max(elapsed)
        # self-check of the N>1 data path (outside the timed region): the device-side sum of the last all-loci step,
        # all-reduced, must equal the sum over ranks of the per-locus values
        last = [p for st, p in zip(iters[-1], plans[-1]) if st.global_decision is not None][-1]
        last.launch()
        if p2p is not None:
            p2p.allreduce(sum_buf.data_ptr(), sum_view.numel())
        else:
            D.dist.all_reduce(sum_view)
        D.torch.cuda.synchronize()
        got = float(sum_view.sum().item())
        want = D.torch.tensor([float(last.lnl().sum())], dtype=D.torch.float64, device="cuda")
        D.dist.all_reduce(want)
        allreduce_check = "ok" if abs(got - float(want.item())) <= 1e-9 * abs(got) else f"MISMATCH {got} vs {float(want.item())}"

    ms_per_step = 1e3 * elapsed / steps
    total_loci = D.sum_int(nloci) if D else nloci
    roofline = None
    if tm and tm["launches"]:
        # achieved = algorithmic bytes of the kernels the events bracketed / their time: the chain
        # and step kernels (K4 + K1 + K2) for JC69, the K1 + K2 kernel where the P-matrix phase is its own launch
        kern = dominant_kernel(cfg, D is None)
        achieved = tm["bytes"] / (tm["partials_ms"] * 1e-3) / 1e9
        traffic, src = traffic_from_profiles(config_key, kern) if args.loci is None else (None, None)
        roofline = dict(bound="hbm", kernel=kern, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=src,
                        avg_kernel_us=round(1e3 * tm["partials_ms"] / tm["launches"], 3),
                        avg_us_per_proposal_step=round(1e3 * tm["partials_ms"] / max(tm["steps"], 1), 3),
                        algorithmic_bytes_per_launch=round(tm["bytes"] / tm["launches"]),
                        launches=tm["launches"], proposal_steps=tm["steps"], profile_head=profile_head(config_key) if traffic else None,
                        timing=f"hipExtLaunchKernelGGL start/stop events on the engine stream, every {args.event_stride}-th launch of the timed region")
        add_honest_fracs(roofline, tm["bytes_codes"] / max(tm["bytes"], 1.0), flops_per_byte * tm["bytes"] / tm["launches"])
    out = dict(
        workload=cfg["name"] + f"; {nloci} loci on this rank, {sum(len(d['weights']) for d in data) / nloci:.2f} patterns/locus, "
                 f"{steps_per_iter:.0f} batched proposal steps/iteration ({it_node_updates / nloci:.1f} node updates + "
                 f"{steps_per_iter:.0f} lnL evals per locus)",
        steps=steps, warmup=warmup, ms_per_step=round(ms_per_step, 5),
        iterations_per_s=round(steps / elapsed, 3),
        iterations_per_s_10k_loci=round((total_loci / 10000.0) * steps / elapsed, 3),
        loci_total=total_loci,
        site_lnl_updates_per_s=round(it_pattern_updates * (total_loci / nloci) * steps / elapsed),
        node_updates_per_iteration=round(float(it_node_updates)),
        gflops_partials=round(it_flops * (total_loci / nloci) * steps / elapsed / 1e9, 2),
        roofline=add_frac_pmc(roofline), allreduce_check=allreduce_check, codes_ratio=round(codes_ratio_k1k2, 5),
        note="pre-recorded proposal tape, accept/reject by a seeded coin: the likelihood path alone (no MCMC decision)")
    for it in plans:
        for p in it:
            p.close()
    p_init.close()
    return out, (init, iters)


# step lengths the unmodified program's burn-in (finetune = 1) tunes itself to on a 10 000-locus set of this shape (the
# `statistical_efficiency` section measures them again on the box: `step_lengths`); the all-loci ones shrink with sqrt(loci)
PROGRAM_FT = dict(gage=18.5, gspr=0.0019, theta=3e-5, tau=1.9e-5, mix=0.0059)


def run_sampler(eng, cfg, data, loci, args, D, first_locus, steps, warmup, moves="program", codes_ratio=None):
    """BASELINE's metric proper: whole A00 MCMC iterations/s with every decision on the device.
    moves = "program" (JC69 loci on the persistent kernel): BPP's own iteration — its generator and Bactrian-Laplace windows
    (bpa_sampler_set_proposal_kernel), theta by the metropolized Gibbs draw 9 times in 10, thetas re-drawn inside the
    rubber-band and the mixing step (bpa_sampler_set_program_moves: stree.c:3957, 5840; prop_mixing.c:272);
    "uniform": our 64-bit streams, uniform windows, a sliding-window theta, no re-draws (rounds 1-3's headline)."""
    import bpp_amd
    from bpp_amd import synth
    nloci = len(data)
    world = D.world if D else 1
    smp = bpp_amd.Sampler(eng, loci, data, seed=1)
    if D is not None:
        # loci sharded; one small sum all-reduce per THETA (all populations together) / TAU / MIX step on the engine's stream
        smp_sum = D.torch.zeros(16, dtype=D.torch.float64, device=f"cuda:{D.local_rank}")      # BPA_SAMPLER_SUMS

        def smp_allreduce(ptr, count, stream):
            if D.p2p is not None:
                D.p2p.allreduce(ptr, count)
            else:
                D.dist.all_reduce(smp_sum[:count])
            return True
        # RCCL backend: the exchange as NATIVE code (libbpp_amd_rccl.so: ncclAllReduce on the engine's stream behind the
        # library's callback type, its own communicator) — no Python runs inside bpa_sampler_iterate; the Python callback
        # (torch.distributed / the p2p exchange) otherwise
        native = None
        use_p2p = D.p2p is not None and cfg["model"] == "jc69" and not os.environ.get("BENCH_PY_ALLREDUCE")
        if use_p2p:
            pass
        elif D.p2p is None and os.environ.get("BENCH_DIST_BACKEND", "nccl") == "nccl" and not os.environ.get("BENCH_PY_ALLREDUCE"):
            try:
                uid = D.torch.zeros(128, dtype=D.torch.uint8, device=f"cuda:{D.local_rank}")
                if D.rank == 0:
                    uid.copy_(D.torch.tensor(list(bpp_amd.RcclExchange.unique_id()), dtype=D.torch.uint8))
                D.dist.broadcast(uid, 0)
                native = bpp_amd.RcclExchange(bytes(uid.cpu().tolist()), D.world, D.rank, D.local_rank)
            except Exception as ex:       # noqa: BLE001
                log(f"native RCCL exchange unavailable ({str(ex)[:120]}): torch.distributed instead")
                native = None
        if use_p2p:
            smp.set_p2p(D.p2p, first_locus)
        elif native is not None:
            smp.set_allreduce_native(native, None, first_locus)
        else:
            smp.set_allreduce(smp_allreduce, smp_sum.data_ptr(), first_locus)
        D.sampler_exchange = ("inside the persistent kernel, through the ranks' peer-mapped mailboxes over xGMI (bpa_sampler_set_p2p; self-tested against RCCL at start-up)" if use_p2p else
                              "native RCCL (libbpp_amd_rccl.so: ncclAllReduce on the engine stream, no Python in the loop)" if native is not None else
                              "p2p one-shot exchange kernel per step" if D.p2p is not None else "torch.distributed")
    sp_parent, sp_tau, sp_theta = synth.species_tree_arrays(cfg["taxa"])
    div = cfg.get("divergence", 1.0)                  # (config 4's set is simulated at 3 x the default divergence: the chain starts where the data were made)
    sp_tau, sp_theta = [t * div for t in sp_tau], [t * div for t in sp_theta]
    smp.set_species_tree(sp_parent, sp_tau, sp_theta)
    smp.set_tau_prior(3.0, 3.0 / sp_tau[-1])
    generic = cfg["model"] != "jc69"
    # the program's moves run inside the persistent kernel: one rank, or several exchanging through the in-kernel mailboxes
    program = (not generic) and moves == "program" and (D is None or (D.p2p is not None and not os.environ.get("BENCH_PY_ALLREDUCE")))
    # ... on the generic sampler the host of every rank decides from the sums over all ranks' loci (through the all-reduce callback)
    generic_program = generic and moves == "program"
    if generic_program:
        # BPP's own iteration on the generic sampler: its generator / Bactrian-Laplace windows / acceptance rule in the per-locus
        # kernels, THETA / TAU / MIX with their theta re-draws decided on the host from the loci's device sums (gs_prog_*); the
        # program's default step lengths (bpp.c:530-549), tuned below by its burn-in rule
        smp.set_proposal_kernel(1)
        smp.set_program_moves(True, 0.1)
        smp.set_theta_prior(2.0, 2.0 / sp_theta[0], 0.001)
        smp.set_finetune(5.0, 0.001, 0.001, 0.3)
    elif generic:
        smp.set_theta_prior(2.0, 2.0 / sp_theta[0], 0.5 * sp_theta[0])
    elif program:
        scale = math.sqrt(10000.0 / max(D.sum_int(len(data)) if D else len(data), 1))
        smp.set_proposal_kernel(1)
        smp.set_program_moves(True, 0.1)                                 # (bpp.c:650: the sliding window 1 time in 10)
        smp.set_theta_prior(2.0, 2.0 / sp_theta[0], PROGRAM_FT["theta"] * scale)
        smp.set_finetune(PROGRAM_FT["gage"], PROGRAM_FT["gspr"], PROGRAM_FT["tau"] * scale, PROGRAM_FT["mix"] * scale)
    else:
        # step lengths of the all-loci moves at which the chain moves (acceptance of THETA / TAU / MIX around 0.3): they shrink
        # with the square root of the number of loci (all ranks' loci: one decision for the whole data set).  The work of an
        # iteration does not depend on them — every proposal is evaluated in full before its decision.
        scale = 1.0 / math.sqrt(max(D.sum_int(len(data)) if D else len(data), 1))
        smp.set_theta_prior(2.0, 2.0 / sp_theta[0], 0.008 * scale)
        smp.set_finetune(0.004, 0.004, 0.004 * scale, 0.6 * scale)
    gtr = cfg["model"] == "gtr"
    if gtr:
        # the per-locus substitution-parameter moves of a GTR + Gamma analysis (3 frequencies, 5 exchangeabilities, alpha)
        for i, d in enumerate(data):
            smp.set_subst_model(i, d["freqs"], d["exch"], 0.5)
        smp.set_subst_moves(0.2, 0.3, 0.5, 1.0, 1.0)
    smp.initialize()
    kind = smp.kind()
    sync = D.sync if D else eng.synchronize
    burnin_ft = None
    if generic_program:
        burnin_ft = smp.burnin(400)
    if program and kind == "persistent":       # (N > 1: every rank runs it; the per-locus moves' counts are pooled through the mailboxes)
        # the program's burn-in (finetune = 1): 800 iterations, the step lengths reset from the acceptance proportions after every
        # quarter and at the end (bpa_sampler_burnin: reset_finetune, method.c:1508-1516, 5364) — outside the timed region
        burnin_ft = smp.burnin(800)
    # A step = `ips` MCMC iterations, chosen so that the timed region lasts >= 0.25 s whatever --steps is (an iteration
    # of config 2 takes ~0.1 ms: 20 of them would be a 2 ms region); the rate does not depend on it.  Every rank takes
    # the same count (the ranks enter the same collectives).
    smp.iterate(max(warmup, 1))
    sync()
    probe = 2 if generic else 20
    t0 = time.perf_counter()
    smp.iterate(probe)
    sync()
    est = (time.perf_counter() - t0) / probe
    if D is not None:
        est = D.max(est)
    ips = max(1, int(np.ceil(0.3 / max(est * steps, 1e-9))))      # (0.3: the probe runs a little slower than the long launch)
    niter = steps * ips
    while True:
        smp.iterate(warmup)
        sync()
        w0 = smp.work()
        l0 = smp.summary()["launches"]
        smp.enable_timing(0 if args.no_timing_events else (1 if kind == "persistent" else args.event_stride))
        t0 = time.perf_counter()
        smp.iterate(niter)
        enq = time.perf_counter() - t0            # host time to enqueue the timed region (the GPU is still running)
        sync()
        dt = time.perf_counter() - t0
        log(f"sampler ({kind}): {steps} steps x {ips} iterations in {dt:.3f} s; host enqueue {1e3 * enq / niter:.4f} ms/iteration of {1e3 * dt / niter:.4f} ms/iteration")
        if D is not None and D.p2p is not None:
            # a p2p exchange that timed out on ANY rank voids the run (checked before anything reads the sampler's state)
            bad = D.torch.tensor([1 if D.p2p.status() != 0 else 0], dtype=D.torch.int64, device="cuda")
            D.dist.all_reduce(bad, op=D.dist.ReduceOp.MAX)
            if int(bad.item()) != 0:
                log("p2p exchange timed out during the sampler section: repeating it over RCCL")
                D.p2p = None
                if use_p2p:
                    # the exchange lived inside the kernel: a fresh sampler on the callback path
                    try:
                        smp.close()
                    except Exception:       # noqa: BLE001
                        pass
                    return run_sampler(eng, cfg, data, loci, args, D, first_locus, steps, warmup)
                continue
        tm = smp.timing()
        smp.enable_timing(0)
        l1 = smp.summary()["launches"]
        w1 = smp.work()
        break
    if D is not None:
        dt = D.max(dt)
    total_loci = D.sum_int(nloci) if D else nloci
    sm = smp.summary()
    npop_inner = cfg["taxa"] - 1
    roofline = None
    if generic and (tm["sweep_launches"] + tm["allloci_launches"]):
        # the engine's step kernel, launched once per proposal step by the sampler: all timed launches together
        nl = max(w1["sweeps"] - w0["sweeps"], 1)
        bytes_per_launch = (w1["bytes"] - w0["bytes"]) / nl
        us = 1e3 * (tm["sweep_ms"] + tm["allloci_ms"]) / (tm["sweep_launches"] + tm["allloci_launches"])
        achieved = bytes_per_launch / (us * 1e-6) / 1e9
        kern = dominant_kernel(cfg)
        traffic, src = traffic_from_profiles("c3" if gtr else "c4", kern) if args.loci is None else (None, None)
        if traffic and smp.streams() >= 2:
            # the profile's dispatches are whole-batch launches (tape + BPA_GS_NOSPLIT=1 sampler); the launches timed here
            # cover half the loci: no PMC figure for a launch of this size
            traffic, src = None, f"{src}: {traffic} B per launch over ALL loci; the launches timed here are half-batches"
        roofline = dict(bound="hbm", kernel=kern, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=src, avg_kernel_us=round(us, 3),
                        algorithmic_bytes_per_launch=round(bytes_per_launch), launches=tm["sweep_launches"] + tm["allloci_launches"],
                        node_updates_per_launch=round((w1["node_updates"] - w0["node_updates"]) / nl),
                        timing=f"hipExtLaunchKernelGGL start/stop events on the engine stream, every {args.event_stride}-th launch of the step kernel in the timed region",
                        note="the generic device-resident sampler (csrc/gsampler.hpp): every proposal step = one launch of a per-locus "
                             "proposal kernel (trees in HBM) + the engine's step kernel over the records it wrote; algorithmic bytes = K1 + K2 "
                             "of the node updates the proposals actually asked for (device counters)"
                             + (f"; the per-locus steps run as {smp.streams()} part-batch launches on as many streams that overlap in time "
                                "(csrc/gsampler_host.hpp gs_fork): a launch here covers a part of the loci and its duration includes the "
                                "time it shares the chip with the other parts' launches — the kernel alone: likelihood_only" if smp.streams() >= 2 else ""),
                        streams=smp.streams())
    elif kind == "persistent" and tm["sweep_launches"]:
        # every launch of the timed region carries events; a launch = up to 4096 whole iterations of all loci
        nl = tm["sweep_launches"]
        bytes_per_launch = (w1["bytes"] - w0["bytes"]) / nl
        us = 1e3 * tm["sweep_ms"] / nl
        achieved = bytes_per_launch / (us * 1e-6) / 1e9
        kname = f"smp2::iter_kernel<{4 if cfg['taxa'] <= 4 else 8}>"
        traffic, src = traffic_from_profiles("c2", "iter_kernel") if args.loci is None else (None, None)
        roofline = dict(bound="hbm", kernel=kname, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=src,
                        avg_kernel_us=round(us, 3), algorithmic_bytes_per_launch=round(bytes_per_launch), launches=nl,
                        iterations_per_launch=round(niter / nl, 1), kernel_us_per_iteration=round(us * nl / niter, 3),
                        algorithmic_bytes_per_iteration=round((w1["bytes"] - w0["bytes"]) / niter),
                        node_updates_per_iteration=round((w1["node_updates"] - w0["node_updates"]) / niter),
                        timing="hipExtLaunchKernelGGL start/stop events on the engine stream around EVERY launch of the timed region",
                        note=f"one launch = whole MCMC iterations of every locus ({3 * cfg['taxa'] - 3} per-locus proposals, the theta step, "
                             f"{cfg['taxa'] - 1} tau steps and the mixing step each, proposal control and decisions included); algorithmic "
                             "bytes = K1 + K2 + K4 (SURVEY 8d) of the node updates the proposals actually ran, per-locus AND all-loci steps "
                             "(device counters).  The loci's state stays in LDS for the whole launch: HBM sees one load and one store of it "
                             "per LAUNCH (`traffic`), so the fraction says how fast the likelihood work is done, not how busy the HBM is — "
                             "the kernel is bound by the latency of one wave's dependent instruction stream (DESIGN 6)")
    elif tm["sweep_launches"]:
        sweeps = max(w1["sweeps"] - w0["sweeps"], 1)
        bytes_per_sweep = (w1["bytes"] - w0["bytes"]) / sweeps
        us = 1e3 * tm["sweep_ms"] / tm["sweep_launches"]
        achieved = bytes_per_sweep / (us * 1e-6) / 1e9
        traffic, src = traffic_from_profiles("c2", "sweep_kernel") if args.loci is None else (None, None)
        nt_ = 4 if cfg['taxa'] <= 4 else 8
        roofline = dict(bound="hbm", kernel=(f"smp2::iter_kernel<{nt_}> (the per-locus sweep of one iteration per launch; the all-loci steps: smp::sweep_kernel<{nt_}>)"
                                             if kind == "hybrid" else f"smp::sweep_kernel<{nt_}>"), achieved=round(achieved, 2),
                        peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=src,
                        avg_kernel_us=round(us, 3), algorithmic_bytes_per_launch=round(bytes_per_sweep),
                        launches=tm["sweep_launches"],
                        node_updates_per_launch=round((w1["node_updates"] - w0["node_updates"]) / sweeps),
                        allloci_step_avg_us=(round(1e3 * tm["allloci_ms"] / tm["allloci_launches"], 3) if tm["allloci_launches"] else None),
                        timing=f"hipExtLaunchKernelGGL start/stop events on the engine stream, every {args.event_stride}-th sweep / all-loci launch of the timed region",
                        note=f"one launch = the {3 * cfg['taxa'] - 3} per-locus proposals of an iteration for every locus, proposal control included; "
                             "algorithmic bytes = K1 + K2 + K4 of the node updates the proposals actually ran (device counters); the working "
                             "set lives in LDS for the whole launch: latency-bound by the leader lanes' serial proposal code, not by HBM")
    if roofline is not None:
        # K1 flops of the timed launches: pattern updates x R x (4 S^2 - S) (device counters); `codes_ratio`: the same config's
        # tape (exact per plan: bpa_plan_work_codes) — the samplers count node updates, not which children were tips
        S_ = 20 if cfg["model"] == "lg" else 4
        fl = (w1["pattern_updates"] - w0["pattern_updates"]) * cfg["rate_cats"] * (4 * S_ * S_ - S_)
        nl_ = max(round((w1["bytes"] - w0["bytes"]) / max(roofline["algorithmic_bytes_per_launch"], 1)), 1)
        add_honest_fracs(roofline, codes_ratio, fl / nl_, "codes / algorithmic byte ratio of this config's tape (likelihood_only.codes_ratio)" if codes_ratio else None)
        if roofline.get("traffic"):
            roofline["profile_head"] = profile_head("c2" if cfg["model"] == "jc69" else "c3" if gtr else "c4")
    out = dict(iterations_per_s=round(niter / dt, 3), iterations_per_s_10k_loci=round(niter / dt * total_loci / 10000.0, 3),
               ms_per_iteration=round(1e3 * dt / niter, 5), ms_per_step=round(1e3 * dt / steps, 4), iterations_per_step=ips,
               timed_region_s=round(dt, 4), steps=steps, warmup=warmup, n_gpus=world, loci_total=total_loci, kind=kind,
               proposals_per_locus_iteration=3 * cfg["taxa"] - 3 + (9 if gtr else 0),
               launches_per_iteration=round(max(l1 - l0 - (0 if kind == "persistent" else 1), 0) / niter, 4),      # (-1: the settle launch of the first summary)
               acceptance=round(sm["accepted"] / max(sm["proposals"], 1), 3),
               # pattern (site) log-likelihood updates of THIS run: node updates x the locus's patterns, device counters over the timed region
               site_lnl_updates_per_s=round((w1["pattern_updates"] - w0["pattern_updates"]) * (total_loci / nloci) / dt),
               moves=("the program's (BPP v4.8.7 defaults): legacy_rndu + Bactrian-Laplace windows, theta by the metropolized Gibbs draw 9 times in 10 "
                      "(stree.c:3957), thetas re-drawn inside the rubber band (stree.c:5840) and the mixing step (prop_mixing.c:272); step lengths "
                      "from the program's burn-in rule run on the device (bpa_sampler_burnin, 800 iterations)" if program else
                      "uniform windows on the library's 64-bit streams, sliding-window theta, no theta re-draws in TAU / MIX" if not generic else
                      "the program's (BPP v4.8.7 defaults) on the generic sampler: legacy_rndu + Bactrian-Laplace windows in the per-locus kernels, "
                      "THETA (Gibbs 9 in 10) / TAU / MIX with their theta re-draws decided on the host from the loci's device sums; step lengths from "
                      "the program's burn-in rule (bpa_sampler_burnin, 400 iterations)" if generic_program else
                      "uniform windows on the library's 64-bit streams (generic sampler)"),
               moves_short=("program" if (program or generic_program) else "uniform"),
               theta_gibbs_draws=(dict(zip(("proposed", "accepted"), smp.gibbs_counters())) if program else None),
               step_lengths_after_burnin=({k: float(f"{v:.4g}") for k, v in burnin_ft.items()} if burnin_ft else None),
               theta_gibbs_draws_generic=(dict(zip(("proposed", "accepted"), smp.gibbs_counters())) if generic_program else None),
               taus_after=[float(x) for x in smp.taus()[cfg["taxa"]:]],
               thetas_after=[float(x) for x in smp.thetas()[cfg["taxa"]:]],
               roofline=add_frac_pmc(roofline),
               implementation=("generic path (csrc/gsampler.hpp): proposals on the device as records for the engine's step kernels; "
                               "tree moves + 3 frequency, 5 exchangeability and 1 alpha move per locus" if gtr else
                               "generic path (csrc/gsampler.hpp): proposals on the device as the records of the tiled 20-state kernels "
                               "(pmatrix_wg2_kernel, partials_lnl_pipe20_kernel, lnl_reduce_wave_kernel); tree moves only (the empirical "
                               "amino-acid models have no free parameters)" if generic else
                               "persistent iteration kernel (csrc/sweep2.hpp): all iterations of a call in one launch, the loci's state in "
                               "LDS, a group of lanes per locus, all-loci decisions from device-scope fixed-point accumulators" if kind == "persistent" else
                               "several ranks: the per-locus sweep of an iteration = one launch of the persistent kernel (csrc/sweep2.hpp), the "
                               "all-loci steps one launch each (csrc/sampler.hpp) with the sums all-reduced in between" if kind == "hybrid" else
                               "LDS sweep kernel (csrc/sampler.hpp): all per-locus proposals of an iteration in one launch, one launch per all-loci step"),
               note="the A00 sampler (species tree fixed) resident on the device: population-aware GAGE+GSPR per locus, a THETA "
                    "step per population, a rubber-band TAU step per divergence and one MIX step per iteration, "
                    "Metropolis-Hastings on priors x MSC density x likelihood (density bit-equal to gtree_logprob); reproduces "
                    "the unmodified program's posterior (tests/test_a00_posterior.py) and the C host driver's trajectory on "
                    "the reference (tests/test_gpu_sampler.py, tests/test_gpu_host_driver.py)")
    if D is not None and D.p2p is not None and D.p2p.status() != 0:
        out = dict(error="a p2p exchange timed out during the sampler section: its numbers are void (rerun without --p2p-sums)")
    smp.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--loci", type=int, default=None, help="loci of the data set (default: the config's)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1: weak = every rank owns its own data set of the config's size; strong = ONE data set of the "
                         "config's size, loci dealt to the ranks by the reference's zig-zag (threads.c:265-353)")
    ap.add_argument("--full-record", default=None,
                    help="where the full record goes (default: bench_full.json next to this file and under gpurun_out/); the stdout line is the compact one")
    ap.add_argument("--tape-iters", type=int, default=4, help="distinct A00 iterations in the resident tape")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true", help="c2: skip the device-resident sampler (`value` is then the tape's)")
    ap.add_argument("--no-tape", action="store_true", help="c2: skip the likelihood-only tape section")
    ap.add_argument("--no-host-control", action="store_true", help="c2: skip the host-driven section (a00_driver.c on the GPU back-end)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the c3 / c4 tape sections of the default run")
    ap.add_argument("--efficiency", action="store_true",
                    help="run the ESS/s section (the program's chain with its burn-in next to the device's: ~50 s more) and the long thread sweep")
    ap.add_argument("--no-efficiency", action="store_true",
                    help="skip the ESS/s section (one more 1 900-iteration run of the reference program with its burn-in, BPP's move kernel on the device)")
    ap.add_argument("--no-bpp-program", action="store_true",
                    help="skip timing the unmodified reference program (thread sweep) on the host cores")
    ap.add_argument("--no-scale-projection", action="store_true", help="skip the one-GPU measurements of the per-rank shares of N = 2, 4, 8")
    ap.add_argument("--projection-iters", type=int, default=20, help="iterations timed per share of the scale projection (c3 / c4)")
    ap.add_argument("--no-uniform-kernel", action="store_true", help="c2: skip the companion run with the library's uniform-window moves")
    ap.add_argument("--no-timing-events", action="store_true")
    ap.add_argument("--p2p-sums", action="store_true",
                    help="N > 1: exchange the sums with the one-shot p2p all-reduce over xGMI peer mappings (self-tested "
                         "against RCCL at start-up) instead of RCCL (torch.distributed), the default")
    ap.add_argument("--c4-divergence", type=float, default=None,
                    help="config 4: divergence factor of the synthetic amino-acid set (default 3: ~195 patterns per locus; 1: SURVEY 8d's literal theta 0.02 / tau_root 0.05, 105 patterns)")
    ap.add_argument("--p2p", action="store_true",
                    help="(the default at N > 1 since round 5; accepted for old command lines) exchange the all-loci sums INSIDE the "
                         "persistent kernel through peer-mapped mailboxes over xGMI: the program's moves then run at N > 1 too.  The "
                         "mailboxes are self-tested against RCCL on every rank at start-up, every wait is bounded, and a rank that "
                         "times out sends the whole section back over RCCL (--no-p2p's path)")
    ap.add_argument("--no-p2p", action="store_true",
                    help="N > 1: no peer-mapped mailboxes at all — the sampler then runs its all-loci steps one launch each with a "
                         "native RCCL all-reduce in between (the persistent kernel only for the per-locus sweeps; north_star's wording)")
    ap.add_argument("--sum-launch", action="store_true",
                    help="produce the total of an all-loci step with a launch of its own (default: per-workgroup partial sums written by the step kernel)")
    ap.add_argument("--event-stride", type=int, default=7,
                    help="attach the kernel start/stop events to every n-th launch of the timed region")
    ap.add_argument("--no-subst-proposals", action="store_true",
                    help="GTR configs: leave the per-locus frequency / exchangeability / alpha proposals out of the tape")
    ap.add_argument("--host-in-loop", action="store_true",
                    help="tape: copy the per-locus lnL of every step back to the host before launching the next")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        sys.exit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if args.c4_divergence is not None:
        CONFIGS["c4"]["divergence"] = args.c4_divergence
    cfg = CONFIGS[args.config]
    nloci_cfg = args.loci or cfg["loci"]
    # N > 1: `value` is the weak-scaling figure for c2 (BASELINE configs[1] is "on 1 MI355X": every rank gets that workload)
    # and the strong-scaling one for c3 / c4 (configs[2]: "10 000 loci ... sharded across 8 MI355X"); the OTHER mode's sampler
    # rate is measured in the same run and reported as value_strong / value_weak — never one silently for the other
    scaling_given = args.scaling is not None
    if args.scaling is None:
        args.scaling = "weak" if args.config == "c2" else "strong"

    import bpp_amd
    from bpp_amd import synth, shard

    # BENCH_FORCE_DIST=1: the N>1 code path (launch segments + an all-reduce per all-loci step) in a one-rank group
    D = Dist(world, rank, local_rank) if (world > 1 or os.environ.get("BENCH_FORCE_DIST")) else None
    if D is not None:
        local_rank = D.local_rank
    eng = bpp_amd.Engine(local_rank, D.stream if D else None)

    # ---- synthetic input, resident in HBM
    def make_data(mode):
        if mode == "strong" and world > 1:
            # ONE data set; the reference's zig-zag deal by work = tips x patterns (threads.c:265-353)
            full = synth.make_dataset(nloci_cfg, cfg["sites"], cfg["taxa"], cfg["model"], cfg["rate_cats"], seed=12345, divergence=cfg.get("divergence", 1.0))
            mine = shard.partition([len(d["seqs"]) * len(d["weights"]) for d in full], world)[rank]
            return [full[i] for i in mine], int(1 << 20) * rank          # distinct per-locus random streams on every rank
        return (synth.make_dataset(nloci_cfg, cfg["sites"], cfg["taxa"], cfg["model"], cfg["rate_cats"], seed=12345 + 1000 * rank, divergence=cfg.get("divergence", 1.0)),
                rank * nloci_cfg)
    t0 = time.time()
    data, first_locus = make_data(args.scaling)
    nloci = len(data)
    npat = sum(len(d["weights"]) for d in data)
    log(f"dataset: {nloci} loci on rank 0, {npat} patterns ({npat / nloci:.2f}/locus) in {time.time() - t0:.1f}s ({args.scaling})")
    loci = make_loci(eng, data)
    if D is not None and not args.no_p2p and not os.environ.get("BENCH_NO_P2P"):
        # the mailboxes of the one-shot exchange over xGMI peer mappings (self-tested against RCCL on every rank): the
        # device-resident sampler exchanges its sums through them INSIDE its persistent kernel; the tape only with --p2p-sums
        D.setup_p2p(eng, 256)
        D.tape_p2p = bool(args.p2p_sums) and D.p2p is not None

    tape_sec = sampler_sec = None
    tape_steps = None
    if not (args.config in ("c2", "c3") and args.no_tape):
        log("section: likelihood-only tape")
        tape_sec, tape_steps = run_tape(eng, cfg, args.config, data, loci, args, D, args.steps, args.warmup)
    if args.config in ("c2", "c3", "c4") and not args.no_sampler:
        log("section: device-resident sampler (headline)")
        sampler_sec = run_sampler(eng, cfg, data, loci, args, D, first_locus, args.steps, args.warmup, codes_ratio=(tape_sec or {}).get("codes_ratio"))
        if "error" in sampler_sec:
            log(sampler_sec["error"])
        elif args.config == "c2" and D is None and not args.no_uniform_kernel:
            # rounds 1-3's headline iteration (half the program's effective samples per iteration), for comparison
            u = run_sampler(eng, cfg, data, make_loci(eng, data), args, None, first_locus, max(args.steps // 4, 5), args.warmup, moves="uniform", codes_ratio=(tape_sec or {}).get("codes_ratio"))
            sampler_sec["device_uniform_kernel"] = {k: u[k] for k in ("iterations_per_s", "ms_per_iteration", "acceptance", "moves", "roofline") if k in u}

    other_mode = None
    if D is not None and world > 1 and sampler_sec is not None and "error" not in sampler_sec and not scaling_given:
        om = "strong" if args.scaling == "weak" else "weak"
        try:
            d_o, fl_o = make_data(om)
            # (the mailboxes belong to the first engine: with them the other mode's loci live there too)
            e_o = eng if D.p2p is not None else bpp_amd.Engine(local_rank, D.stream)
            s_o = run_sampler(e_o, cfg, d_o, make_loci(e_o, d_o), args, D, fl_o, max(args.steps // 2, 5), args.warmup)
            if e_o is not eng:
                e_o.close()
            other_mode = dict(scaling=om, iterations_per_s=s_o.get("iterations_per_s"), iterations_per_s_10k_loci=s_o.get("iterations_per_s_10k_loci"),
                              loci_total=s_o.get("loci_total"), ms_per_iteration=s_o.get("ms_per_iteration"), kind=s_o.get("kind"), error=s_o.get("error"))
        except Exception as ex:       # noqa: BLE001
            other_mode = dict(scaling=om, error=str(ex)[:300])

    cpu = None
    if rank == 0 and not args.no_cpu_baseline and tape_steps is not None:
        init, iters = tape_steps
        n_cpu_iter = min(2, len(iters))
        log("section: CPU tape replay")
        cb = cpu_baseline(data, [init], [s for it in iters[:n_cpu_iter] for s in it], n_cpu_iter)
        scale = nloci_cfg if args.config != "c2" else 10000.0
        ac = cb["all_cores"]
        cpu = dict(value=round(1.0 / (cb["sec_per_locus_iter"] * scale), 3),
                   unit=f"iterations/s ({int(scale)}-locus A00 iterations, likelihood hot path only: the tape of `likelihood_only`)",
                   cores=cb["cores"], kind=cb["kind"],
                   sample=f"{cb['sampled_loci']} loci x {n_cpu_iter} tape iterations x {cb['repeats']} repeats "
                          f"({cb['seconds']:.1f}s incl. the all-cores leg), same tape as the GPU, AVX2 back-end",
                   all_cores=dict(value=round(1.0 / (ac["sec_per_locus_iter"] * scale), 3), cores=ac["workers"],
                                  host_logical_cores=ac["host_logical_cores"], host_cpu_quota=cpu_quota(),
                                  workers_tried={k: round(1.0 / (v * scale), 3) for k, v in ac["tried"].items()},
                                  sample=f"{ac['sampled_loci']} loci x {n_cpu_iter} tape iterations x {ac['repeats']} repeats, "
                                         f"one locus per worker thread at a time (loci are independent: threads.c:87-200)"))

    bpp_prog = None
    chain = None
    # the ESS/s section (a 2 900-iteration run of the program with its burn-in + the device's chain: ~50 s) is opt-in since round 6
    # (--efficiency): the default run must finish in about two minutes; the line carries the committed long runs' ESS-per-iteration
    # ratios (profiles/r*/ess_*.json) either way
    want_eff = args.efficiency and not args.no_efficiency
    if rank == 0 and world == 1 and args.config == "c2" and not args.no_cpu_baseline and not args.no_bpp_program:
        try:
            ncores = os.cpu_count() or 1
            # one thread, what the CPU quota grants (the box shows 256 logical cores and grants 16: more threads than that are
            # only throttled) and twice that — round 2's sweep over 8 ... 128 found the best there every time
            q = int(cpu_quota() or ncores)
            sweep = sorted({1, max(2, min(q, ncores)), max(2, min(2 * q, ncores))})
            log("section: the unmodified program on the host (thread sweep)")
            r = (bpp_program_baseline(nloci_cfg, cfg["sites"], sweep, reps=2, budget_s=150.0, chain_samples=2500, long_short=(100, 900)) if want_eff
                 else bpp_program_baseline(nloci_cfg, cfg["sites"], sweep))
            chain = r.pop("chain", None) if r else None
            if r:
                best = max((k for k in r if k > 1), key=lambda k: r[k]["median"], default=1)
                bpp_prog = dict(unit="whole MCMC iterations/s of the unmodified reference program (10k loci, A00 JC69), incl. its MCMC control",
                                threads={str(k): v for k, v in r.items()}, best_threads=best, best_median=r[best]["median"],
                                host_logical_cores=ncores, host_cpu_quota=cpu_quota(), kind="reference",
                                sample="bpp --simulate data (seed 12345), differential wall time of a 100- and a 700-iteration run per thread count "
                                       "(1 thread: 10 vs 70): ~45 s of CPU work (--efficiency: 900-iteration runs, two measurements each, as in rounds 3-5)")
        except Exception as ex:       # noqa: BLE001
            bpp_prog = dict(error=str(ex)[:200])

    # ---- statistical efficiency: BPP's own move kernel and tuned step lengths on the device, ESS/s next to the program's
    efficiency = None
    if bpp_prog and "error" not in bpp_prog and chain and "error" not in chain and chain.get("finetune"):
        try:
            log("section: statistical efficiency")
            efficiency = run_efficiency(eng, cfg, data, None, chain, bpp_prog["best_median"])
        except Exception as ex:       # noqa: BLE001
            efficiency = dict(error=str(ex)[:300])
    elif chain and "error" in chain:
        efficiency = dict(error=chain["error"])

    # ---- the other single-GPU configurations of BASELINE.json on the same box (tape = likelihood path)
    others = None
    if args.config == "c2" and world == 1 and D is None and not args.no_other_configs:
        others = {}
        for key, k_steps, k_warm in (("c3", 12, 2), ("c4", 6, 1)):
            try:
                oc = CONFIGS[key]
                t0 = time.time()
                e2 = bpp_amd.Engine(local_rank, None)
                log(f"section: other config {key}")
                d2 = synth.make_dataset(oc["loci"], oc["sites"], oc["taxa"], oc["model"], oc["rate_cats"], seed=12345, divergence=oc.get("divergence", 1.0))
                l2 = make_loci(e2, d2)
                a2 = argparse.Namespace(**vars(args))
                a2.tape_iters = 2
                a2.loci = None
                sec, _ = run_tape(e2, oc, key, d2, l2, a2, None, k_steps, k_warm)
                sec["unit"] = f"iterations/s (one iteration = the A00 proposal schedule over this config's {oc['loci']} loci)"
                sec["device_resident_sampler"] = run_sampler(e2, oc, d2, l2, a2, None, 0, k_steps, k_warm, codes_ratio=sec.get("codes_ratio"))
                e2.close()
                if not args.no_cpu_baseline:
                    sec["cpu_baseline"] = other_config_cpu_baseline(key, d2)
                    if sec["cpu_baseline"]:
                        sec["ratio"] = dict(sampler_over_cpu_baseline=round(sec["device_resident_sampler"]["iterations_per_s"] / sec["cpu_baseline"]["value"], 1),
                                            tape_over_cpu_baseline=round(sec["iterations_per_s"] / sec["cpu_baseline"]["value"], 1))
                if not args.no_scale_projection:
                    sec["scale_projection"] = scale_projection(key, oc, d2, a2)
                sec["seconds"] = round(time.time() - t0, 1)
                others[key] = sec
            except Exception as ex:       # noqa: BLE001
                others[key] = dict(error=str(ex)[:300])

    if others is not None:
        try:
            t0 = time.time()
            e1 = bpp_amd.Engine(local_rank, None)
            log("section: config 1")
            others["c1"] = dict(device_resident_sampler=run_config1(e1), seconds=None)
            e1.close()
            if not args.no_cpu_baseline:
                others["c1"]["cpu_baseline"] = other_config_cpu_baseline("c1", None)
                if others["c1"]["cpu_baseline"]:
                    others["c1"]["ratio"] = dict(sampler_over_cpu_baseline=round(others["c1"]["device_resident_sampler"]["iterations_per_s"] / others["c1"]["cpu_baseline"]["value"], 3))
            others["c1"]["seconds"] = round(time.time() - t0, 1)
        except Exception as ex:       # noqa: BLE001
            others["c1"] = dict(error=str(ex)[:300])
        try:
            t0 = time.time()
            e5 = bpp_amd.Engine(local_rank, None)
            log("section: config 5")
            others["c5"] = dict(device_resident_sampler=run_config5(e5), seconds=None)
            e5.close()
            if not args.no_cpu_baseline:
                others["c5"]["cpu_baseline"] = other_config_cpu_baseline("c5", None)
                if others["c5"]["cpu_baseline"]:
                    others["c5"]["ratio"] = dict(sampler_over_cpu_baseline=round(others["c5"]["device_resident_sampler"]["iterations_per_s"] / others["c5"]["cpu_baseline"]["value"], 3))
            others["c5"]["seconds"] = round(time.time() - t0, 1)
        except Exception as ex:       # noqa: BLE001
            others["c5"] = dict(error=str(ex)[:300])

        if args.config == "c2" and args.loci is None:
            try:
                log("section: mixed set")
                others["mixed_set"] = run_mixed_set(data)
            except Exception as ex:       # noqa: BLE001
                others["mixed_set"] = dict(error=str(ex)[:300])

    # ---- MCMC control on the host in C (last: libgomp pins the calling thread under OMP_PROC_BIND, and threads or
    # processes started afterwards would inherit that one-CPU mask — the CPU baselines above must not)
    host_sec = None
    if rank == 0 and world == 1 and D is None and args.config == "c2" and not args.no_host_control:
        mask = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
        try:
            q = cpu_quota()
            log("section: host control in C")
            host_sec = run_host_control(eng, cfg, data, max(1, min(16, int(q) if q else (os.cpu_count() or 1))))
        except Exception as ex:       # noqa: BLE001
            host_sec = dict(error=str(ex)[:300])
        if mask is not None:
            os.sched_setaffinity(0, mask)

    # ---- what one rank of an N-GPU strong-scaling run of this config would work on, measured here (N = 1 only)
    projection = None
    if rank == 0 and world == 1 and D is None and args.config in ("c2", "c3", "c4") and not args.no_scale_projection and not args.no_sampler:
        try:
            log("section: scale projection")
            projection = scale_projection(args.config, cfg, data, args)
        except Exception as ex:       # noqa: BLE001
            projection = dict(error=str(ex)[:300])

    if rank == 0:
        headline_sampler = sampler_sec is not None and "error" not in sampler_sec
        parallelism = "1 GPU"
        if D is not None:
            parallelism = (f"loci sharded over {world} GPU(s) ({args.scaling}), " +
                           (("the sums of the THETA / TAU / MIX steps exchanged " + (D.sampler_exchange or "over RCCL")) if headline_sampler else
                            ("one-shot p2p all-reduce over xGMI (RCCL-checked at start-up)" if D.tape_p2p else "RCCL all-reduce") +
                            " of the sums the THETA / TAU / MIX steps are decided on"))
        if headline_sampler:
            value = sampler_sec["iterations_per_s_10k_loci"] if (args.scaling == "weak" and args.config in ("c2", "c3")) else sampler_sec["iterations_per_s"]
            ms_per_step = sampler_sec["ms_per_step"]
            roofline = sampler_sec.pop("roofline")
            metric = BASELINE_METRIC
        elif tape_sec is not None:
            value = (tape_sec["iterations_per_s_10k_loci"] if (args.config == "c2" and args.scaling == "weak") else tape_sec["iterations_per_s"])
            ms_per_step = tape_sec["ms_per_step"]
            roofline = tape_sec["roofline"]
            metric = BASELINE_METRIC + " [likelihood hot path only: proposal tape]"
        else:                                    # (--no-sampler --no-tape: only the side sections were asked for)
            value, ms_per_step, roofline = None, None, None
            metric = "no headline section was run (--no-sampler --no-tape)"
        loci_unit = 10000 if args.config in ("c2", "c3") else nloci_cfg
        # ---- `cpu_baseline` is like-for-like with `value`: whole MCMC iterations/s of the unmodified reference program at
        # its best thread count on this box's host cores; the tape replay through the reference's locus API (likelihood
        # path only, comparable with `likelihood_only`) keeps its own key
        cpu_like = cpu
        vs_baseline = None
        ratios = None
        if headline_sampler and bpp_prog and "error" not in bpp_prog:
            b = bpp_prog["threads"][str(bpp_prog["best_threads"])]
            cpu_like = dict(value=b["median"], unit="iterations/s (whole A00 MCMC iterations over 10000 loci, MCMC control included: like `value`)",
                            cores=bpp_prog["best_threads"], kind="reference",
                            sample=f"the unmodified program (oracle/_ref/bpp, AVX2, threads = {bpp_prog['best_threads']} 1 1) on its own simulated "
                                   f"10000-locus data set, {b['iterations']} iterations differential, {b['runs']} measurements",
                            spread=dict(min=b["min"], max=b["max"]), one_thread=bpp_prog["threads"].get("1", {}).get("median"),
                            host_logical_cores=bpp_prog["host_logical_cores"], host_cpu_quota=bpp_prog["host_cpu_quota"],
                            note="the box grants this container `host_cpu_quota` CPUs (cgroup cpu.max) of its logical cores: the thread "
                                 "count is the best inside that quota, and the ratio below holds against THAT many cores")
        if headline_sampler and args.config == "c2" and args.scaling == "weak":
            # the reference program measured in the survey container (other hardware: 8-vCPU Xeon
            # 2.1 GHz), threads = 8, this metric on this configuration
            vs_baseline = round(value / 25.9, 2)
        if cpu_like and cpu_like.get("value"):
            ratios = dict(value_over_cpu_baseline=round(value / cpu_like["value"], 1))
            if cpu and tape_sec and cpu is not cpu_like:
                ratios["likelihood_only_over_tape_replay_all_cores"] = round(
                    (tape_sec["iterations_per_s_10k_loci"] if args.config == "c2" else tape_sec["iterations_per_s"]) / cpu["all_cores"]["value"], 1)
                ratios["likelihood_only_over_tape_replay_one_core"] = round(
                    (tape_sec["iterations_per_s_10k_loci"] if args.config == "c2" else tape_sec["iterations_per_s"]) / cpu["value"], 1)
        out = {
            "metric": metric,
            "value": value,
            "unit": (f"iterations/s (one iteration = one A00 MCMC iteration over {loci_unit} loci" +
                     ("; weak scaling: N data sets of that size, value = N x per-rank rate)" if (D is not None and args.scaling == "weak") else ")")),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "iterations_per_step": sampler_sec["iterations_per_step"] if headline_sampler else 1,
            "ms_per_iteration": sampler_sec["ms_per_iteration"] if headline_sampler else ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "value_weak": (None if D is None else (value if args.scaling == "weak" else
                                                   (other_mode or {}).get("iterations_per_s_10k_loci" if args.config in ("c2", "c3") else "iterations_per_s"))),
            "value_strong": (None if D is None else (value if args.scaling == "strong" else (other_mode or {}).get("iterations_per_s"))),
            "scaling_other_mode": other_mode,
            # BASELINE.json's `published` is empty (BASELINE.md says so): no published number for this metric -> null.  What rounds 2-5
            # reported here — value / 25.9, the survey's OWN measurement of the program with threads = 8 on its 8-vCPU container (other
            # hardware) — keeps a key of its own; the same-box figure is cpu_baseline
            "vs_baseline": None,
            "vs_survey_measurement": ({"ratio": vs_baseline, "ref": "BASELINE.md section 2: 25.9 iterations/s = the unmodified program, threads = 8, on the survey "
                                       "container's 8-vCPU Xeon 2.1 GHz (other hardware; not a published number)"} if vs_baseline else None),
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["name"] + f"; {nloci} loci on rank 0, {npat / nloci:.2f} patterns/locus" +
                       (f"; {3 * cfg['taxa'] - 3} gene-tree proposals per locus + a theta step per population + {cfg['taxa'] - 1} tau + 1 mixing step per iteration; moves: " + sampler_sec.get("moves", "?") if headline_sampler else ""),
                       "workload_short": cfg["name"] + f"; {nloci} loci on rank 0, {npat / nloci:.1f} patterns/locus",
                       "parallelism": parallelism,
                       "moves": (sampler_sec.get("moves_short") if headline_sampler else "tape (accept/reject by a seeded coin)"),
                       "decisions": ("every proposal, density, likelihood and accept/reject on the device" if headline_sampler else None)},
            # the second half of the metric comes from the SAME run as `value` (the sampler's device counters); the tape's own
            # figure stays under likelihood_only
            "site_lnl_updates_per_s": (sampler_sec.get("site_lnl_updates_per_s") if headline_sampler else tape_sec["site_lnl_updates_per_s"] if tape_sec else None),
            "roofline": roofline,
            "cpu_baseline": cpu_like,
            "speedups": ratios,
            "cpu_tape_replay": cpu if cpu is not cpu_like else None,
            "reference_program_on_host": bpp_prog,
            "statistical_efficiency": efficiency,
            "ess_per_s": (None if not efficiency or "error" in efficiency else dict(
                device=dict(tau_root=efficiency["device"]["tau_root"]["ess_per_s"], theta_root=efficiency["device"]["theta_root"]["ess_per_s"]),
                reference_program=dict(tau_root=efficiency["reference_program"]["tau_root"]["ess_per_s"], theta_root=efficiency["reference_program"]["theta_root"]["ess_per_s"]),
                ratio=efficiency["ratio_ess_per_s"])),
            "scale_projection": projection,
            "device_resident_sampler": sampler_sec,
            "host_control_in_c": host_sec,
            "likelihood_only": tape_sec,
            "other_configs": others,
            "allreduce_check": tape_sec["allreduce_check"] if tape_sec else None,
            "allreduce": (None if D is None else dict(
                sampler=D.sampler_exchange,
                tape=("p2p one-shot exchange kernel over xGMI peer mappings (bpa_p2p_*), self-tested against RCCL at start-up" if D.tape_p2p and D.p2p is not None
                      else "RCCL (torch.distributed)" if os.environ.get("BENCH_DIST_BACKEND", "nccl") == "nccl" else "gloo (test switch)"))),
        }
    if D is not None and D.p2p is not None:
        D.p2p.close()
    eng.close()
    if D is not None:
        D.dist.destroy_process_group()
        # RCCL writes its version banner to the C stdout buffer, which a pipe only flushes at exit: push it out now so
        # that the JSON line is the LAST line on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        write_full_record(out, args.full_record)
        print(json.dumps(compact_line(out), separators=(",", ":")), flush=True)


if __name__ == "__main__":
    main()
