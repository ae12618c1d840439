"""Statistical efficiency per ITERATION, the C host driver with the program's moves against the unmodified program
(oracle/_ref/bpp) on the SAME synthetic data set (CPU only): ESS of tau / theta traces, step lengths from the program's
burn-in.   python tools/ess_compare.py [nloci] [sites] [samples]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bpp_amd import synth
import hostdrv
from bench import ess

nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 500
nsample = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
variant = sys.argv[4] if len(sys.argv) > 4 else "program"
data = synth.make_dataset(nloci, sites, 4, "jc69", 1, seed=5)
names = "ABCD"
td = tempfile.mkdtemp(prefix="essx")
with open(os.path.join(td, "seqs.txt"), "w") as f:
    for d in data:
        seqs = ["".join(ch * int(w) for ch, w in zip(s, d["weights"])) for s in d["seqs"]]
        f.write(f"4 {len(seqs[0])}\n")
        for nm, s in zip(names, seqs):
            f.write(f"s^{nm.lower()}  {s}\n")
        f.write("\n")
open(os.path.join(td, "imap.txt"), "w").write("a A\nb B\nc C\nd D\n")
open(os.path.join(td, "a00.ctl"), "w").write(
    "seed = 1\nseqfile = seqs.txt\nImapfile = imap.txt\njobname = out\nspeciesdelimitation = 0\n"
    "speciestree = 0\nspecies&tree = 4  A B C D\n                  1 1 1 1\n                 (((A, B), C), D);\nusedata = 1\n"
    f"nloci = {nloci}\ncleandata = 0\nthetaprior = gamma 2 1000\ntauprior = gamma 2 500\nfinetune = 1\nprint = 1 0 0 0\n"
    f"burnin = 2000\nsampfreq = 1\nnsample = {nsample}\nthreads = 8 1 1\n")
r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "bpp"), "--cfile", "a00.ctl"], cwd=td, capture_output=True, text=True)
out = r.stdout
ft = re.findall(r"finetune = 1 Gage:(\S+) Gspr:(\S+) th1:(\S+) th2:(\S+) tau:(\S+) mix:([0-9.eE+-]+)", out)
print("program finetune:", ft[-1] if ft else None)
for ln in out.split("\n"):
    if "Gage" in ln or "pjump" in ln.lower() or re.match(r"\s*100%", ln):
        print("   ", ln[:200])
rows = [ln.split("\t") for ln in open(os.path.join(td, "out.mcmc.txt"))]
head, body = [h.strip() for h in rows[0]], np.array([[float(x) for x in r_] for r_ in rows[1:]])
print("columns", head)
for c in range(1, body.shape[1]):
    print(f"  program {head[c]:14s} mean {body[:, c].mean():.6g} sd {body[:, c].std():.3g} ESS/iter {ess(body[:, c]) / len(body):.4f}")
ftv = dict(zip(("gage", "gspr", "th1", "th2", "tau", "mix"), map(float, ft[-1])))
drv = hostdrv.reference_driver(data, seed=6)
drv.set_proposal_kernel(1)
if variant == "program":
    drv.set_program_moves(True, 0.1)
parent, tau, theta = synth.species_tree_arrays(4)
drv.set_species_tree(parent, tau, theta)
drv.set_tau_prior(2.0, 500.0)
drv.set_theta_prior(2.0, 1000.0, ftv["th2"])
drv.set_finetune(ftv["gage"], ftv["gspr"], ftv["tau"], ftv["mix"])
drv.set_threads(8)
drv.initialize()
os.environ.pop("A00_DECLOG", None)
for _ in range(1000):
    drv.iterate()
S = []
for _ in range(nsample):
    drv.iterate()
    S.append(drv.thetas()[4:] + drv.taus()[4:] + [drv.total_lnl()])
S = np.array(S)
lab = ["theta_AB", "theta_ABC", "theta_root", "tau_AB", "tau_ABC", "tau_root", "lnL"]
for c in range(S.shape[1]):
    print(f"  driver  {lab[c]:14s} mean {S[:, c].mean():.6g} sd {S[:, c].std():.3g} ESS/iter {ess(S[:, c]) / len(S):.4f}")
print("driver counters", drv.counters(), "gibbs", drv.gibbs_counters())
