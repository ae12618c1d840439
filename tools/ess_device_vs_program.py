"""Effective samples per ITERATION: the device sampler with the program's moves AND the program's burn-in rule
(bpa_sampler_burnin: step lengths reset from the acceptance proportions, method.c:1122-1153, 5364) against the unmodified
program (oracle/_ref/bpp, finetune = 1, the same burn-in) on the SAME data set, both from the program's default step lengths
(bpp.c:530-549).  ESS by Geyer's initial positive sequence on `batches` consecutive batches of each trace: mean and standard error.

    python tools/ess_device_vs_program.py [nloci] [samples] [burnin] [threads] [batches]      (GPU box; writes JSON to stdout)
"""
import json, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bpp_amd
from bpp_amd import synth
import tape
from bench import ess

nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
burnin = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 8
batches = int(sys.argv[5]) if len(sys.argv) > 5 else 10
data = synth.make_dataset(nloci, 1000, 4, "jc69", 1, seed=12345)


def batch_ess(x):
    x = np.asarray(x, dtype=float)
    per = [ess(b) / len(b) for b in np.array_split(x, batches)]
    return dict(ess_per_iteration=float(np.mean(per)), se=float(np.std(per, ddof=1) / np.sqrt(len(per))), whole_trace=ess(x) / len(x),
                mean=float(x.mean()), sd=float(x.std()))


# ---- the program
td = tempfile.mkdtemp(prefix="essdp")
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
    f"burnin = {burnin}\nsampfreq = 1\nnsample = {nsample}\n" + (f"threads = {threads} 1 1\n" if threads > 1 else ""))
t0 = time.time()
r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "bpp"), "--cfile", "a00.ctl"], cwd=td, capture_output=True, text=True)
prog_s = time.time() - t0
ft = re.findall(r"finetune = 1 Gage:(\S+) Gspr:(\S+) th1:(\S+) th2:(\S+) tau:(\S+) mix:([0-9.eE+-]+)", r.stdout)
prog_ft = dict(zip(("gage", "gspr", "th1", "th2", "tau", "mix"), map(float, ft[-1]))) if ft else None
rows = [ln.split("\t") for ln in open(os.path.join(td, "out.mcmc.txt"))]
head, body = [h.strip() for h in rows[0]], np.array([[float(x) for x in r_] for r_ in rows[1:]])
col = {h: i for i, h in enumerate(head)}
prog = {"tau_root": batch_ess(body[:, col["tau:5ABCD"]] if "tau:5ABCD" in col else body[:, [i for h, i in col.items() if h.startswith("tau:5")][0]]),
        "theta_root": batch_ess(body[:, [i for h, i in col.items() if h.startswith("theta:5")][0]])}

# ---- the device: the program's defaults, its burn-in rule, its moves
eng = bpp_amd.Engine(0)
smp = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=11)
parent, tau, theta = synth.species_tree_arrays(4)
smp.set_species_tree(parent, tau, theta)
smp.set_tau_prior(2.0, 500.0)
smp.set_proposal_kernel(1)
smp.set_program_moves(True, 0.1)
smp.set_theta_prior(2.0, 1000.0, 0.001)          # opt_finetune_theta[0] (bpp.c:549)
smp.set_finetune(5.0, 0.001, 0.001, 0.3)          # Gage, Gspr, tau, mix (bpp.c:530-546)
smp.initialize()
dev_ft = smp.burnin(burnin)
tr_tau, tr_theta = [], []
t0 = time.time()
for _ in range(nsample):
    smp.iterate(1)
    tr_tau.append(smp.taus()[-1]); tr_theta.append(smp.thetas()[-1])
dev_s = time.time() - t0
pj, _ = smp.adapt_finetune()                       # (acceptance proportions of the sampling phase; the step lengths it returns are not used)
# (the program prints 6 decimals: the device's traces are rounded the same way before the ESS)
dev = {"tau_root": batch_ess(np.round(tr_tau, 6)), "theta_root": batch_ess(np.round(tr_theta, 6))}
out = dict(nloci=nloci, samples=nsample, burnin=burnin, batches=batches, program=prog, device=dev,
           program_finetune=prog_ft, device_finetune=dev_ft, device_pjump_sampling=pj, program_seconds=round(prog_s, 1), device_seconds=round(dev_s, 1))
for k in ("tau_root", "theta_root"):
    a, b = dev[k], prog[k]
    ratio = a["ess_per_iteration"] / b["ess_per_iteration"]
    out[k + "_ratio"] = dict(value=round(ratio, 3), se=round(ratio * float(np.hypot(a["se"] / a["ess_per_iteration"], b["se"] / b["ess_per_iteration"])), 3))
print(json.dumps(out, indent=1))
