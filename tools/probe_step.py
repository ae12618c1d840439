"""per-phase wall-clock stamps inside step_jc69_kernel on config 2 (profiling aid)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bpp_amd
from bpp_amd import synth
import tape
eng = bpp_amd.Engine(0)
data = synth.make_dataset(10000, 1000, 4, "jc69", 1, seed=12345)
loci = tape.make_engine_loci(eng, data)
sch = tape.make_schedule(data, seed=1)
init = sch.initial_step()
steps = sch.iteration() + sch.iteration()
p0 = tape.plan_for_step(eng, loci, init); p0.launch(); p0.lnl()
plans = [tape.plan_for_step(eng, loci, st) for st in steps]
for p in plans: p.launch()
eng.synchronize()
L = bpp_amd.lib()
L.bpa_plan_probe.restype = C.c_int
L.bpa_plan_probe.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
names = ["entry", "lvl1 lane_rec", "lvl2 record", "lvl3 inputs", "updates done", "root+log", "sum", "tail mats"]
for st, p in list(zip(steps, plans))[:13]:
    out = (C.c_double * 9)()
    # keep the GPU busy right before, as in the bench
    for q in plans[:5]: q.launch()
    assert L.bpa_plan_probe(p.h, out)
    print(f"{st.kind:5s} ops={len(st.ops):6d} mats={len(st.mat_pmatrix):6d} | " + " ".join(f"{names[i].split()[0]}={out[i]:.2f}" for i in range(8)) + f" | span={out[8]:.2f} us")
