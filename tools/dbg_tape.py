import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import bpp_amd
from bpp_amd import synth
import tape, oraclelib as O
eng=bpp_amd.Engine(0)
for taxa,model,R,scaling,nloci in [(4,"jc69",1,False,30),(8,"gtr",4,False,10)]:
    data=synth.make_dataset(nloci,400,taxa,model,R,seed=21)
    loci=tape.make_engine_loci(eng,data,scaling)
    sch=tape.make_schedule(data,seed=4,scaling=scaling,taus=(0.001,0.002,0.003) if taxa==4 else (0.0011,0.0025,0.005))
    steps=[sch.initial_step()]
    for _ in range(3): steps+=sch.iteration()
    got=[]
    for st in steps:
        p=tape.plan_for_step(eng,loci,st); p.launch(); got.append(p.lnl()); p.close()
    for li in range(nloci):
        sub=tape.locus_subtape(steps,li)
        want=tape.oracle_replay(data[li],sub,scaling)
        mine=np.array([got[s["step"]][s["task"]] for s in sub])
        bad=[(s["step"],s["kind"],float(m-w)) for s,m,w in zip(sub,mine,want) if abs(m-w)>1e-13*abs(w)]
        if bad: print(taxa,li,"bad:",bad[:6], "nsteps",len(steps))
print("done")
