"""where a workgroup of the multi-category 4-state step kernel spends its life: per-workgroup wall-clock stamps (bpa_plan_probe)
of one per-locus step of a config-3 tape, mean over workgroups.  usage: python tools/probe_klane.py [loci]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bpp_amd
from bpp_amd import synth
from bpp_amd.schedule import A00Schedule, TreeState
import bench
nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = bench.CONFIGS["c3"]
eng = bpp_amd.Engine(0)
data = synth.make_dataset(nloci, cfg["sites"], cfg["taxa"], cfg["model"], cfg["rate_cats"], seed=12345)
loci = bench.make_loci(eng, data)
trees = [TreeState(d["left"], d["right"], d["times"], d["root"]) for d in data]
sch = A00Schedule(trees, seed=1, taus=cfg["taus"], subst=None)
init = sch.initial_step(); it = sch.iteration()
def mk(st):
    return bpp_amd.Plan(eng, [loci[i] for i in st.loci], st.mat_off, st.mat_pmatrix, st.mat_length, st.op_off, st.ops, st.root_clv, st.root_scaler)
p0 = mk(init); p0.launch(); p0.lnl()
L = bpp_amd.lib()
L.bpa_plan_probe.argtypes = [C.c_void_p, C.POINTER(C.c_double)]; L.bpa_plan_probe.restype = C.c_int
names = ["entry", "lane table in", "slot + records in LDS", "matrices + children in", "updates issued", "K2 + barrier", "site terms + barrier", "end"]
for st in it[:6]:
    p = mk(st)
    for _ in range(3): p.launch()
    eng.synchronize()
    out = (C.c_double * 24)()
    assert L.bpa_plan_probe(p.h, out), bpp_amd.api._err()
    w = p.work()
    print(f"step: {w['node_updates']} node updates, {out[17]:.0f} workgroups, span {out[8]:.1f} us")
    print("   since kernel start: " + " | ".join(f"{out[i]:.1f}" for i in range(8)))
    print("   since own start:    " + " | ".join(f"{names[i]} {out[9+i]:.1f}" for i in range(8)))
    p.close()
