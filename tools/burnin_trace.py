"""the burn-in rule round by round on a config-shaped set with the program's moves: python tools/burnin_trace.py [c3|c4] [loci]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, bpp_amd, bench
from bpp_amd import synth
key = sys.argv[1] if len(sys.argv) > 1 else "c4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
oc = bench.CONFIGS[key]
eng = bpp_amd.Engine(0)
data = synth.make_dataset(n, oc["sites"], oc["taxa"], oc["model"], oc["rate_cats"], seed=12345, divergence=oc.get("divergence", 1.0))
smp = bpp_amd.Sampler(eng, bench.make_loci(eng, data), data, seed=1)
parent, tau, theta = synth.species_tree_arrays(oc["taxa"])
print("start taus", tau[oc["taxa"]:], "thetas", theta[oc["taxa"]:])
smp.set_proposal_kernel(1); smp.set_program_moves(True, 0.1)
smp.set_species_tree(parent, tau, theta); smp.set_tau_prior(3.0, 3.0/tau[-1]); smp.set_theta_prior(2.0, 2.0/theta[0], 0.001)
smp.set_finetune(5.0, 0.001, 0.001, 0.3)
smp.initialize()
for r in range(6):
    ft = smp.burnin(400)
    g = smp.gibbs_counters()
    print(r, {k: float(f"{v:.3g}") for k, v in ft.items()}, "gibbs", g, "taus", [float(f"{x:.4g}") for x in smp.taus()[oc["taxa"]:]], "thetas", [float(f"{x:.4g}") for x in smp.thetas()[oc["taxa"]:]], "lnL", round(smp.summary()["total_lnl"], 2))
smp.iterate(400)
print("pjump", smp.adapt_finetune()[0])
