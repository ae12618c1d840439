import os, sys, time
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bpp_amd
from bpp_amd import synth
import tape
eng = bpp_amd.Engine(0)
data = synth.make_dataset(10000, 1000, 4, "jc69", 1, seed=12345)
loci = tape.make_engine_loci(eng, data)
sch = tape.make_schedule(data, seed=1)
init = sch.initial_step()
steps = sch.iteration()
p0 = tape.plan_for_step(eng, loci, init); p0.launch(); p0.lnl()
for st in steps[:3]:
    p = tape.plan_for_step(eng, loci, st); p.launch(); p.lnl(); p.close()
t_c = t_l = t_g = t_x = 0
for st in steps:
    t0 = time.perf_counter(); p = tape.plan_for_step(eng, loci, st); t1 = time.perf_counter()
    p.launch(); t2 = time.perf_counter(); p.lnl(); t3 = time.perf_counter(); p.close(); t4 = time.perf_counter()
    t_c += t1 - t0; t_l += t2 - t1; t_g += t3 - t2; t_x += t4 - t3
n = len(steps)
print(f"per step: create {1e3*t_c/n:.3f} ms, launch {1e3*t_l/n:.3f}, get_lnl {1e3*t_g/n:.3f}, destroy {1e3*t_x/n:.3f}")
