"""two ranks (gloo, sharing the GPU): hybrid path and one-launch-per-step path side by side in each process"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, torch.distributed as dist
import bpp_amd
from bpp_amd import synth
import tape
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
data = synth.make_dataset(160, 300, 4, "jc69", 1, seed=3)
per = len(data)//world; first = rank*per; mine = data[first:first+per]
eng = bpp_amd.Engine(0)
def make(v1):
    os.environ.pop("BPA_SMP_V1", None)
    if v1: os.environ["BPA_SMP_V1"] = "1"
    s = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, mine), mine, seed=7)
    t = torch.zeros(16, dtype=torch.float64, device="cuda")
    def ar(ptr, count, stream):
        eng.synchronize(); dist.all_reduce(t[:count]); torch.cuda.synchronize(); return True
    s._t = t
    s.set_allreduce(ar, t.data_ptr(), first)
    par, tau, theta = synth.species_tree_arrays(4)
    s.set_species_tree(par, tau, theta)
    s.set_tau_prior(3.0, 1000.0); s.set_theta_prior(2.0, 1000.0, 0.001); s.set_finetune(0.003, 0.005, 0.0008, 0.2)
    s.initialize()
    return s
hyb, old = make(False), make(True)
for it in range(6):
    hyb.iterate(1); old.iterate(1)
    a, b = hyb.summary(), old.summary()
    nd = 0; firstd = None
    for i in range(per):
        x, y = hyb.tree(i), old.tree(i)
        d = [k for k in ("left", "right", "parent", "clv", "pmat", "pop", "time", "lnl", "logpr") if (list(x[k]) if hasattr(x[k], "__len__") else x[k]) != (list(y[k]) if hasattr(y[k], "__len__") else y[k])]
        if d:
            nd += 1
            if firstd is None: firstd = (i, d, {k: (x[k], y[k]) for k in d})
    print(f"rank {rank} it {it}: accepted {a['accepted']} {b['accepted']} proposals {a['proposals']} {b['proposals']} taus equal {hyb.taus() == old.taus()} thetas equal {hyb.thetas() == old.thetas()} loci differing {nd}", firstd if firstd else "", flush=True)
    if nd: break
dist.barrier(); dist.destroy_process_group()
