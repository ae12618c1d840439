import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, bpp_amd
from bpp_amd import synth
import hostdrv, tape
taxa, model, R, nloci = 8, "gtr", 4, 60
eng = bpp_amd.Engine(0)
data = synth.make_dataset(nloci, 300, taxa, model, R, seed=41)
host = hostdrv.hip_driver(eng, tape.make_engine_loci(eng, data), data, seed=43)
dev = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=43)
parent, tau0, thetas = synth.species_tree_arrays(taxa)
for drv in (host, dev):
    drv.set_proposal_kernel(1); drv.set_program_moves(True, 0.3)
    drv.set_species_tree(parent, tau0, thetas); drv.set_tau_prior(3.0, 3.0 / tau0[-1]); drv.set_theta_prior(2.0, 1000.0, 0.0004)
    drv.set_finetune(2.0, 0.002, 0.0004, 0.05)
host.initialize(); dev.initialize()
for it in range(4):
    host.iterate(); dev.iterate(1)
    print(it, 'taus eq', np.array_equal(dev.taus(), host.taus()))
    print('  dev ', [f"{x:.10g}" for x in dev.thetas()[taxa:]])
    print('  host', [f"{x:.10g}" for x in host.thetas()[taxa:]])
    print('  counters', dev.summary()['proposals'], dev.summary()['accepted'], host.counters(), dev.gibbs_counters(), host.gibbs_counters())
    mx=0; mlp=0; ml=0
    for i in range(nloci):
        a,b=dev.tree(i),host.tree(i)
        assert list(a["left"])==list(b["left"]) and list(a["pop"])==list(b["pop"]), i
        ta,tb=np.array(a["time"]),np.array(b["time"]); m=tb>0
        mx=max(mx, float(np.max(np.abs(ta[m]-tb[m])/tb[m])))
        mlp=max(mlp, abs(a["logpr"]-b["logpr"])/abs(b["logpr"])); ml=max(ml, abs(a["lnl"]-b["lnl"])/abs(b["lnl"]))
    print('  max rel diff: times', mx, 'logpr', mlp, 'lnl', ml)
    if it == 2:
        cnt=0
        for i in range(nloci):
            a,b=dev.tree(i),host.tree(i)
            ta,tb=np.array(a["time"]),np.array(b["time"])
            for k in range(len(tb)):
                if tb[k]>0 and abs(ta[k]-tb[k])/tb[k] > 1e-12 and cnt<12:
                    cnt+=1; print('   locus',i,'node',k,'pop',a["pop"][k],'t',ta[k],tb[k],'rel',abs(ta[k]-tb[k])/tb[k])
        print('   taus', dev.taus()[taxa:])
