"""config 2: does cache residency matter?  Average kernel time of the 13 step plans of an iteration cycled in order
(as the bench does) against each plan launched alone, back to back (its records and CLVs stay where they were)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bpp_amd
from bpp_amd import synth
import tape
eng = bpp_amd.Engine(0)
data = synth.make_dataset(10000, 1000, 4, "jc69", 1, seed=12345)
loci = tape.make_engine_loci(eng, data)
sch = tape.make_schedule(data, seed=1)
init = sch.initial_step()
iters = [sch.iteration() for _ in range(4)]
p0 = tape.plan_for_step(eng, loci, init); p0.launch(); p0.lnl()
plans = [[tape.plan_for_step(eng, loci, st) for st in it] for it in iters]


def timed(seq, reps):
    for p in seq: p.launch()
    eng.synchronize()
    eng.enable_timing(True, stride=1)
    for _ in range(reps):
        for p in seq: p.launch()
    eng.synchronize()
    tm = eng.timing(); eng.enable_timing(False)
    return 1e3 * tm["partials_ms"] / tm["launches"]


print(f"4 iterations x 13 plans cycled: {timed([p for it in plans for p in it], 10):.2f} us per launch")
print(f"1 iteration x 13 plans cycled:  {timed(plans[0], 40):.2f} us per launch")
for k in (0, 3, 9, 12):
    print(f"plan {k} ({iters[0][k].kind}) alone:        {timed([plans[0][k]], 300):.2f} us per launch")
