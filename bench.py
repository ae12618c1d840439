#!/usr/bin/env python3
"""bench.py — throughput of the likelihood hot path on the configuration BASELINE.json's
metric is quoted on: A00, synthetic 10 000 loci x 1 000 sites, 4 taxa, JC69, 1 rate
category (configs[1]) — per GPU (weak scaling: every rank owns its own 10 000 loci).

A "step" is one A00 MCMC iteration's worth of hot-path work for all loci of the rank:
the batched proposal steps of bpp_amd/schedule.py (3 GAGE + 6 GSPR + 3 TAU + 1 MIX for
4 taxa; each = P-matrices -> root-path partials -> root lnL for every locus in one fused
launch sequence; TAU/MIX steps also produce the all-loci lnL sum that is all-reduced
across ranks).  All descriptors and loci are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3] [--loci L]

N>1 is launched by the driver with torch.distributed.run (one rank per GPU, RCCL).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s peak

CONFIGS = {
    "c2": dict(taxa=4, model="jc69", rate_cats=1, sites=1000, loci=10000, taus=(0.001, 0.002, 0.003),
               name="C2: A00, 10000 loci x 1000 sites, 4 taxa, JC69, 1 rate cat"),
    "c3": dict(taxa=8, model="gtr", rate_cats=4, sites=1000, loci=10000, taus=(0.0011, 0.0025, 0.005),
               name="C3: A00, 10000 loci x 1000 sites, 8 taxa, GTR+G4"),
    "c4": dict(taxa=6, model="lg", rate_cats=4, sites=500, loci=2000, taus=(0.01, 0.015, 0.02, 0.035, 0.05),
               name="C4: A00, 2000 loci x 500 aa sites, 6 taxa, LG+G4"),
}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baseline(data, steps_init, steps_iter, n_iter, budget_s=12.0, sample=192):
    """The same tape on the host CPU, one core, on a bounded sample of the loci:
    through the REAL reference's update API when oracle/_ref travelled
    (kind "reference"), else through the oracle's C loop (kind "port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    import tape
    sample = min(sample, len(data))
    use_ref = O.have_ref()
    per_locus = []
    t_start = time.time()
    # calibrate repeats on the first locus, then spend ~budget_s overall
    reps = 200
    done = 0
    for li in range(sample):
        sub_full = tape.locus_subtape(steps_init + steps_iter, li)
        sub_init = tape.locus_subtape(steps_init, li)
        if use_ref:
            rl = tape.ref_locus_for(data[li])
            _, t_full = tape.ref_replay(rl, tape.ref_tape_arrays(sub_full), repeats=reps)
            _, t_init = tape.ref_replay(rl, tape.ref_tape_arrays(sub_init), repeats=reps)
            rl.free()
        else:
            _, t_full = tape.oracle_tape_run(data[li], sub_full, repeats=reps)
            _, t_init = tape.oracle_tape_run(data[li], sub_init, repeats=reps)
        per_locus.append(max(t_full - t_init, 1e-12) / (reps * n_iter))
        done += 1
        if li == 0:
            est = (t_full + t_init)
            reps = int(max(20, min(20000, reps * budget_s / max(est * sample, 1e-9))))
        if time.time() - t_start > 2.5 * budget_s:
            break
    sec_per_locus_iter = float(np.mean(per_locus))
    return dict(sec_per_locus_iter=sec_per_locus_iter, kind="reference" if use_ref else "port",
                cores=1, sampled_loci=done, repeats=reps, seconds=time.time() - t_start)


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


def bpp_program_baseline(nloci, sites, threads_list, n1=20, n2=120):
    """The unmodified reference PROGRAM (oracle/_ref/bpp, built in place from /root/reference) on this
    box's host cores: data from its own simulator (SURVEY.md App. B control files), A00 JC69, whole
    MCMC iterations/s from the differential wall time of an n1- and an n2-iteration run, per thread
    count.  This includes BPP's MCMC control (MSC prior, proposals), which this repo does not build."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as O
    if not os.path.exists(O.REF_BIN):
        return None
    out = {}
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "sim.ctl"), "w").write(SIM_CTL.format(nloci=nloci, sites=sites))
        subprocess.run([O.REF_BIN, "--simulate", "sim.ctl"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=300)
        for th in threads_list:
            tl = f"threads = {th} 1 1" if th > 1 else ""
            ts = []
            for ns in (n1, n2):
                open(os.path.join(d, "a00.ctl"), "w").write(A00_CTL.format(nloci=nloci, nsample=ns, threads=tl))
                t0 = time.perf_counter()
                subprocess.run([O.REF_BIN, "--cfile", "a00.ctl"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=900)
                ts.append(time.perf_counter() - t0)
            out[th] = round((n2 - n1) / max(ts[1] - ts[0], 1e-9) * nloci / 10000.0, 2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--loci", type=int, default=None, help="loci per GPU (default: the config's)")
    ap.add_argument("--tape-iters", type=int, default=4, help="distinct A00 iterations in the resident tape")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true", help="skip the device-resident sampler section")
    ap.add_argument("--no-bpp-program", action="store_true",
                    help="skip timing the unmodified reference program (1 thread and many threads) on the host cores")
    ap.add_argument("--no-timing-events", action="store_true")
    ap.add_argument("--rccl-sums", action="store_true",
                    help="N > 1: all-reduce the sums with RCCL (torch.distributed) instead of the one-shot p2p exchange")
    ap.add_argument("--sum-launch", action="store_true",
                    help="produce the total of an all-loci step with a launch of its own (default: per-workgroup partial sums written by the step kernel)")
    ap.add_argument("--event-stride", type=int, default=7,
                    help="attach the kernel start/stop events to every n-th launch of the timed region "
                         "(an event pair costs ~4 us of stream time per launch; 7 is co-prime with the 13 steps "
                         "of an iteration, so every step type is sampled)")
    ap.add_argument("--no-subst-proposals", action="store_true",
                    help="GTR configs: leave the per-locus frequency / exchangeability / alpha proposals out of the tape")
    ap.add_argument("--host-in-loop", action="store_true",
                    help="copy the per-locus lnL of every step back to the host before launching the next "
                         "(what a host-resident accept/reject needs; PCIe-inclusive rate, reported in DESIGN.md)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    cfg = CONFIGS[args.config]
    nloci = args.loci or cfg["loci"]

    import bpp_amd
    from bpp_amd import synth
    from bpp_amd.schedule import A00Schedule, TreeState

    dist = None
    sum_buf = None
    stream = None
    # BENCH_FORCE_DIST=1: the N>1 code path (launch segments + an all-reduce per all-loci step) in a one-rank group —
    # what the collectives' enqueue costs on the host, measurable on a one-GPU box
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):
        import torch
        import torch.distributed as dist_
        dist = dist_
        # BENCH_DIST_BACKEND=gloo + BENCH_FORCE_DEVICE=0 let the N>1 code path be exercised on a
        # one-GPU box (two ranks sharing device 0); the driver's runs use nccl (= RCCL), one GPU per rank
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if "BENCH_FORCE_DEVICE" in os.environ:
            local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend, rank=rank, world_size=world)
        # the engine and the collectives share ONE explicit torch stream, so that an all-reduce is
        # ordered after the kernel that produced the sum (torch's default stream has handle 0, which
        # the engine would replace by a private stream)
        tstream = torch.cuda.Stream()
        torch.cuda.set_stream(tstream)
        stream = tstream.cuda_stream
        sum_buf = torch.zeros(4096, dtype=torch.float64, device="cuda")     # room for the per-workgroup partial sums
        sum_view = sum_buf[:1]

    eng = bpp_amd.Engine(local_rank, stream)

    # ---- synthetic input (this rank's shard: its own nloci loci), resident in HBM
    t0 = time.time()
    data = synth.make_dataset(nloci, cfg["sites"], cfg["taxa"], cfg["model"], cfg["rate_cats"],
                              seed=12345 + 1000 * rank)
    npat = sum(len(d["weights"]) for d in data)
    log(f"dataset: {nloci} loci, {npat} patterns ({npat / nloci:.2f}/locus) in {time.time() - t0:.1f}s")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
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

    # ---- the proposal tape (host MCMC control stand-in), then resident plans
    t0 = time.time()
    trees = [TreeState(d["left"], d["right"], d["times"], d["root"]) for d in data]
    subst = None
    if cfg["model"] == "gtr" and not args.no_subst_proposals:
        # the per-locus frequency / exchangeability / alpha proposals of a GTR+Gamma analysis
        R_ = cfg["rate_cats"]
        subst = dict(freqs=[d["freqs"] for d in data], exch=[d["exch"] for d in data], alpha=[0.5] * nloci, rate_cats=R_,
                     gamma=lambda a, cats: bpp_amd.compute_gamma_cats(a, a, cats) if cats > 1 else np.ones(1))
    sch = A00Schedule(trees, seed=1 + rank, taus=cfg["taus"], subst=subst)
    init = sch.initial_step()
    iters = [sch.iteration() for _ in range(args.tape_iters)]
    log(f"tape: {args.tape_iters} iterations x {len(iters[0])} batched steps in {time.time() - t0:.1f}s")

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
                n = p.enable_partial_sums(sum_buf.data_ptr() if sum_buf is not None else None,
                                          sum_buf.numel() if sum_buf is not None else 0)
                sum_parts.append(n)
        return p

    sum_parts = []
    p_init = mkplan(init)
    plans = [[mkplan(st) for st in it] for it in iters]
    if sum_buf is not None and sum_parts:
        # every rank must all-reduce the same number of doubles: the largest count over the ranks (a rank's own
        # workgroup count depends on its loci; the entries past it stay zero)
        import torch
        cnt = torch.tensor([max(sum_parts)], dtype=torch.int64, device="cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
        sum_view = sum_buf[:int(cnt.item())]

    # ---- how the sums travel between the ranks: the one-shot all-reduce over xGMI peer mappings (bpa_p2p_*: one hop,
    # one small kernel) when its start-up self-test against RCCL passes on EVERY rank, else RCCL (torch.distributed)
    p2p = None
    if dist is not None and not args.rccl_sums and sum_view.numel() <= 512:
        import torch
        ok = 1
        try:
            p2p = bpp_amd.P2P(eng, rank, world, 512)
            handles = [None] * world
            dist.all_gather_object(handles, p2p.handle)
            p2p.connect(handles)
            for k in range(6):
                x = torch.arange(sum_view.numel(), dtype=torch.float64, device="cuda") * (0.5 + rank) + k + 1e-3 * rank
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
        log("sums between ranks: " + ("one-shot p2p all-reduce (self-test against RCCL passed)" if p2p else "RCCL all-reduce"))
    # parameter installs of the tape, resident in HBM: (which, device address) per step, applied through p_init
    # (which holds every locus) right before the step's launch
    staged = [[[(w, eng.stage(v)) for w, v in st.params] for st in it] for it in iters]
    p_init.launch()
    lnl0 = p_init.lnl()
    log(f"start-up lnL (sum over loci) = {lnl0.sum():.6f}")

    # work per tape iteration (algorithmic)
    work = [[p.work() for p in it] for it in plans]
    it_pattern_updates = np.mean([sum(w["pattern_updates"] for w in it) for it in work])
    it_node_updates = np.mean([sum(w["node_updates"] for w in it) for it in work])
    it_bytes_partials = np.mean([sum(w["bytes_partials"] for w in it) for it in work])
    it_bytes_pmatrix = np.mean([sum(w["bytes_pmatrix"] for w in it) for it in work])
    it_flops = np.mean([sum(w["flops_partials"] for w in it) for it in work])
    launches_per_iter = np.mean([len(it) for it in plans])

    # launch segments: consecutive steps up to and including an all-loci step (TAU/MIX), whose
    # summed lnL is then all-reduced across ranks — the per-proposal reduction of
    # threads.c:544-591, over xGMI.  One GPU: the whole iteration is one host call.
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
            if dist is not None and st.global_decision is not None:
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
                dist.all_reduce(sum_view)

    def sync():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        else:
            eng.synchronize()

    while True:
        for i in range(args.warmup):
            run_iteration(i)
        sync()
        if not args.no_timing_events:
            eng.enable_timing(True, stride=args.event_stride)
        t0 = time.perf_counter()
        for i in range(args.steps):
            run_iteration(args.warmup + i)
        enqueue_s = time.perf_counter() - t0          # host time to enqueue the timed region (GPU still running)
        sync()
        elapsed = time.perf_counter() - t0
        log(f"host enqueue {1e3 * enqueue_s / args.steps:.4f} ms/step of {1e3 * elapsed / args.steps:.4f} ms/step")
        tm = eng.timing() if not args.no_timing_events else None
        eng.enable_timing(False)
        if p2p is None:
            break
        # a p2p exchange that timed out on ANY rank voids the run: measure again over RCCL
        import torch
        bad = torch.tensor([1 if p2p.status() != 0 else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) == 0:
            break
        log("p2p all-reduce timed out: repeating the measurement over RCCL")
        p2p = None

    allreduce_check = None
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # self-check of the N>1 data path (outside the timed region): the device-side sum of the last
        # all-loci step, all-reduced, must equal the sum over ranks of the per-locus values
        last = [p for st, p in zip(iters[-1], plans[-1]) if st.global_decision is not None][-1]
        last.launch()
        if p2p is not None:
            p2p.allreduce(sum_buf.data_ptr(), sum_view.numel())
        else:
            dist.all_reduce(sum_view)
        torch.cuda.synchronize()
        got = float(sum_view.sum().item())
        want = torch.tensor([float(last.lnl().sum())], dtype=torch.float64, device="cuda")
        dist.all_reduce(want)
        allreduce_check = "ok" if abs(got - float(want.item())) <= 1e-9 * abs(got) else f"MISMATCH {got} vs {float(want.item())}"

    ms_per_step = 1e3 * elapsed / args.steps
    total_loci = nloci * world
    iters_per_s_10k = (total_loci / 10000.0) * args.steps / elapsed
    site_lnl_updates_per_s = it_pattern_updates * world * args.steps / elapsed

    roofline = None
    if tm and tm["launches"]:
        kernel_ms = tm["partials_ms"] / tm["launches"]
        # the fused step kernel does K4 (P-matrix writes) + K1 + K2: algorithmic bytes of all three
        # which kernel the engine's events bracket: the single fused step kernel (JC69: K4 + K1 + K2), or — where the
        # P-matrix phase runs as its own launch (GTR+G: eigen code, 20 states) — the K1 + K2 kernel alone
        klane = cfg["model"] == "gtr" and not os.environ.get("BPA_NO_KLANE")
        fused = cfg["model"] != "lg" and not klane
        bytes_per_launch = (it_bytes_partials + (it_bytes_pmatrix if fused else 0)) / launches_per_iter
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        roofline = dict(bound="hbm", kernel=(("step_jc69_kernel<256>" if os.environ.get("BPA_JC69_V1") else "step_jc69_v2_kernel<256>") if cfg["model"] == "jc69" else ("step_s4_klane_kernel<256,false>" if os.environ.get("BPA_KLANE_V1") else "step_s4_klane_v2_kernel<256,false>") if klane else "step_s4_fused_kernel<64,4>" if cfg["model"] != "lg" else "partials_lnl_tiledk_kernel<20,3>"), achieved=round(achieved, 2),
                        peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5),
                        traffic=None, avg_kernel_us=round(1e3 * kernel_ms, 3),
                        algorithmic_bytes_per_launch=round(bytes_per_launch),
                        launches=tm["launches"],
                        timing=f"hipExtLaunchKernelGGL start/stop events on the engine stream, every {args.event_stride}-th launch of the timed region",
                        note=("52k lanes per launch: latency/launch bound (2.3 us empty-grid floor), cache-resident working set, not HBM bound (SURVEY §7)"
                              if args.config == "c2" else None))

    # HBM traffic per launch from the rocprofv3 PMC passes committed under profiles/ (collected by
    # tools/profile_cfg.sh on this same command): 2*FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md
    # §HBM) + WRITE_SIZE, both reported in KB
    if roofline is not None and args.loci is None:
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"profile_{args.config}.json")))
            pm = json.load(open(cands[-1]))["pmc_per_dispatch"]
            want = roofline["kernel"].replace(" ", "")
            key = [k for k in pm if want in k.replace(" ", "") and "FETCH_SIZE" in pm[k]]
            key = (key or [k for k in pm if want.split("<")[0] in k and "FETCH_SIZE" in pm[k]])[0]
            roofline["traffic"] = round((2 * pm[key]["FETCH_SIZE"]["mean"] + pm[key]["WRITE_SIZE"]["mean"]) * 1024)
            roofline["traffic_source"] = os.path.relpath(cands[-1], ROOT) + " (separate --pmc passes; FETCH_SIZE doubled per the gfx950 note)"
        except Exception:
            pass

    # ---- extra (not `value`): the same loci under device-resident proposal control — a real sampler
    # (proposals, accept/reject, rollback on the device; bpa_sampler_t)
    sampler = None
    if args.config == "c2" and not args.no_sampler:
        smp = bpp_amd.Sampler(eng, loci, data, seed=1)
        if dist is not None:
            # loci sharded; one small sum all-reduce per THETA (all populations together) / TAU / MIX step (RCCL on the engine's stream)
            smp_sum = torch.zeros(16, dtype=torch.float64, device=f"cuda:{local_rank}")      # BPA_SAMPLER_SUMS

            def smp_allreduce(ptr, count, stream):
                if p2p is not None:
                    p2p.allreduce(ptr, count)
                else:
                    dist.all_reduce(smp_sum[:count])
                return True
            smp.set_allreduce(smp_allreduce, smp_sum.data_ptr(), rank * nloci)
        sp_parent, sp_tau, sp_theta = synth.species_tree_arrays(cfg["taxa"])
        smp_taus = sp_tau[cfg["taxa"]:]
        smp.set_species_tree(sp_parent, sp_tau, sp_theta)
        smp.set_tau_prior(3.0, 3.0 / sp_tau[-1])
        smp.set_theta_prior(2.0, 2.0 / sp_theta[0], 0.5 * sp_theta[0])
        smp.initialize()
        smp.iterate(args.warmup)
        eng.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        smp.iterate(args.steps)
        eng.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        sm = smp.summary()
        sampler = dict(iterations_per_s=round(args.steps / dt * nloci * world / 10000.0, 1), ms_per_iteration=round(1e3 * dt / args.steps, 4),
                       n_gpus=world,
                       launches_per_iteration=(2 if dist is None else 3) + (2 if dist is None else 3) * (len(smp_taus) + 1),   # sweep, THETA, (TAU.. + MIX) x (step + sum/decide) proposals_per_locus_iteration=3 * cfg["taxa"] - 3,
                       acceptance=round(sm["accepted"] / max(sm["proposals"], 1), 3),
                       taus_after=[float(x) for x in smp.taus()[cfg["taxa"]:]],
                       thetas_after=[float(x) for x in smp.thetas()[cfg["taxa"]:]],
                       note="the A00 sampler (species tree fixed) resident on the device: population-aware GAGE+GSPR per "
                            "locus, a THETA step per population, a rubber-band TAU step per divergence and one MIX step "
                            "per iteration, Metropolis-Hastings on priors x MSC density x likelihood (density bit-equal "
                            "to gtree_logprob); reproduces the unmodified program's posterior "
                            "(tests/test_a00_posterior.py) and the C host driver's trajectory on the reference "
                            "(tests/test_gpu_sampler.py, tests/test_gpu_host_driver.py)")
        if p2p is not None and p2p.status() != 0:
            sampler = dict(error="a p2p exchange timed out during the sampler section: its numbers are void (rerun with --rccl-sums)")
        smp.close()

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        n_cpu_iter = min(2, len(iters))
        cb = cpu_baseline(data, [init], [s for it in iters[:n_cpu_iter] for s in it], n_cpu_iter)
        v = 1.0 / (cb["sec_per_locus_iter"] * 10000.0)
        cpu = dict(value=round(v, 3), unit="iterations/s (10k-locus A00 iterations, hot path only)",
                   cores=cb["cores"], kind=cb["kind"],
                   sample=f"{cb['sampled_loci']} loci x {n_cpu_iter} tape iterations x {cb['repeats']} repeats "
                          f"({cb['seconds']:.1f}s), same tape as the GPU, AVX2 back-end, 1 thread")

    bpp_prog = None
    if rank == 0 and world == 1 and args.config == "c2" and not args.no_cpu_baseline and not args.no_bpp_program:
        try:
            ncores = os.cpu_count() or 1
            many = max(2, min(64, ncores // 2))
            r = bpp_program_baseline(nloci, cfg["sites"], [1, many])
            if r:
                bpp_prog = dict(unit="whole MCMC iterations/s of the unmodified reference program (10k loci, A00 JC69), "
                                     "incl. its MCMC control", threads={str(k): v for k, v in r.items()},
                                host_logical_cores=ncores, kind="reference",
                                sample="bpp --simulate data (seed 12345), differential wall time of 20- vs 120-iteration runs")
        except Exception as ex:       # noqa: BLE001
            bpp_prog = dict(error=str(ex)[:200])

    if rank == 0:
        out = {
            "metric": "MCMC iterations/sec (A00) + site-lnL updates/sec, 10k loci per GPU (likelihood hot path)",
            "value": round(iters_per_s_10k, 3),
            "unit": "iterations/s (one iteration = A00 proposal schedule over 10 000 loci)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["name"] + f"; {nloci} loci/GPU, {npat / nloci:.2f} patterns/locus, "
                       f"{launches_per_iter:.0f} batched proposal steps/iteration "
                       f"({it_node_updates / nloci:.1f} node updates + {launches_per_iter:.0f} lnL evals per locus)",
                       "parallelism": (f"loci sharded over {world} GPU(s), " + ("one-shot p2p all-reduce over xGMI (RCCL-checked at start-up)" if p2p is not None else "RCCL all-reduce") + " of the lnL sum per TAU/MIX step" if world > 1 or dist is not None else "1 GPU")},
            "site_lnl_updates_per_s": round(site_lnl_updates_per_s),
            "node_updates_per_iteration": round(float(it_node_updates)),
            "gflops_partials": round(it_flops * args.steps / elapsed / 1e9, 2),
            "kernel_tflops": (round(it_flops / launches_per_iter / (tm["partials_ms"] / tm["launches"] * 1e-3) / 1e12, 3)
                              if tm and tm["launches"] else None),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "reference_program_on_host": bpp_prog,
            "device_resident_sampler": sampler,
            "allreduce_check": allreduce_check,
            "allreduce": (None if dist is None else "p2p one-shot over xGMI peer mappings (bpa_p2p_*), self-tested against RCCL at start-up" if p2p is not None else "RCCL (torch.distributed)"),
        }
    for it in plans:
        for p in it:
            p.close()
    p_init.close()
    if p2p is not None:
        p2p.close()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()
        # RCCL writes its version banner to the C stdout buffer, which a pipe only flushes at exit: push it out now so
        # that the JSON line is the LAST line on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
