"""where the children of a step's node updates come from, on update lists shaped like configs 3 / 4 (A00 proposal tape of
bpp_amd/schedule.py, CPU only): tip codes | the previous update's parent (forwarded in registers) | a parent of an EARLIER update of
the same step (distance 2, 3, ... updates back: what step_s4_klane_v3_kernel keeps in the lane's LDS words since round 6, and what
used to be read back from HBM behind the lane's own store) | a buffer no update of the step writes (read ahead).
usage: python tools/opstat.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bpp_amd import synth                               # noqa: E402
from bpp_amd.schedule import A00Schedule, TreeState     # noqa: E402
import bench                                            # noqa: E402

for key in ("c3", "c4"):
    cfg = bench.CONFIGS[key]
    data = synth.make_dataset(300, 60, cfg["taxa"], "jc69", 1, seed=5)
    trees = [TreeState(d["left"], d["right"], d["times"], d["root"]) for d in data]
    sch = A00Schedule(trees, seed=1, taus=cfg["taus"], subst=None)
    sch.initial_step()
    tips = cfg["taxa"]
    src = dict(tip=0, old=0)
    dist, nops_hist, nsteps, with_reread = {}, {}, 0, 0
    for _ in range(3):
        for st in sch.iteration():
            ops, off = st.ops, st.op_off
            for t in range(len(off) - 1):
                o0, o1 = off[t], off[t + 1]
                if o1 == o0:
                    continue
                nsteps += 1
                nops_hist[o1 - o0] = nops_hist.get(o1 - o0, 0) + 1
                wrote, rr = {}, False
                for o in range(o0, o1):
                    op = ops[o]
                    for c in (int(op["left_clv"]), int(op["right_clv"])):
                        if c < tips:
                            src["tip"] += 1
                        elif c in wrote:
                            d = o - wrote[c]
                            dist[d] = dist.get(d, 0) + 1
                            rr = rr or d > 1
                        else:
                            src["old"] += 1
                    wrote[int(op["parent_clv"])] = o
                with_reread += rr
    print(f"{key}: {nsteps} locus-steps, updates per step {dict(sorted(nops_hist.items()))}")
    print(f"   children: tips {src['tip']}, not written in the step {src['old']}, written d updates earlier {dict(sorted(dist.items()))}"
          f" (d = 1: forwarded); locus-steps with a child at d > 1: {with_reread}")
