"""worker of tests/test_gpu_p2p.py: one rank of a two-process group sharing GPU 0 (gloo as the side channel)"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bpp_amd


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    eng = bpp_amd.Engine(0)
    p = bpp_amd.P2P(eng, rank, world, 300)
    handles = [None] * world
    dist.all_gather_object(handles, p.handle)
    p.connect(handles)
    rng = np.random.default_rng(100 + rank)
    worst, rounds = 0.0, 300
    for it in range(rounds):
        n = [1, 210, 7, 300][it % 4]
        mine = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 6)
        x = torch.tensor(mine, dtype=torch.float64, device="cuda")
        eng.synchronize(); torch.cuda.synchronize()
        p.allreduce(x.data_ptr(), n)
        assert p.status() == 0
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        want = np.zeros(n)
        for v in allv:                       # rank order: the same additions as the kernel
            want = want + v
        got = x.cpu().numpy()
        worst = max(worst, float(np.max(np.abs(got - want))))
        assert np.array_equal(got, want), (it, n)
    # back to back without host synchronisation in between: the double-buffered mailboxes must hold
    x = torch.full((210,), float(rank + 1), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(50):
        p.allreduce(x.data_ptr(), 210)
    assert p.status() == 0
    tot = float(sum(range(1, world + 1)))
    expect = float(rank + 1)
    for _ in range(50):
        expect = tot if _ == 0 else expect * world
    got = float(x[0].item())
    ok = got == tot * world ** 49
    # a rank that exchanges alone must come back with an error, not hang
    import time
    timed_out = None
    dist.barrier()
    if rank == 0:
        t0 = time.time()
        p.allreduce(x.data_ptr(), 4)
        timed_out = p.status() == 1 and 1.0 < time.time() - t0 < 30.0
    dist.barrier()
    print(json.dumps(dict(rank=rank, rounds=rounds, worst=worst, chained_ok=bool(ok), timed_out=timed_out)), flush=True)
    p.close(); eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
