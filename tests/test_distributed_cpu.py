"""N>1 path on CPU: loci sharded over 2 and over 8 ranks (gloo; 8 = the world size BASELINE's configs name), per-rank sums of per-locus
log-likelihoods (computed here with the oracle standing in for the GPU engine),
all-reduced — must equal the single-process total whatever the partition."""
import os
import sys
import numpy as np
import pytest
import torch.multiprocessing as mp

from bpp_amd import shard


def test_partition_covers_all_loci_once():
    rng = np.random.default_rng(0)
    work = rng.integers(1, 500, 103)
    for nr in (1, 2, 3, 8):
        for zz in (False, True):
            parts = shard.partition(work, nr, zz)
            allidx = np.concatenate(parts)
            assert sorted(allidx) == list(range(103))
        parts = shard.partition(work, nr, True)
        loads = [work[p].sum() for p in parts]
        assert max(loads) - min(loads) <= work.max() * 2        # zig-zag balances
    assert [len(p) for p in shard.partition(np.ones(10), 4, False)] == [2, 3, 2, 3]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import torch.distributed as dist
    from bpp_amd import synth, shard as sh
    import oraclelib as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = synth.make_dataset(24, 200, 4, "jc69", 1, seed=99)          # same data on every rank
    work = [4 * len(d["weights"]) for d in data]
    mine = sh.partition(work, world, True)[rank]
    lnl = [O.OracleLocus(4, 1, data[i]["seqs"], data[i]["weights"]).full_lnl(
        data[i]["left"], data[i]["right"], data[i]["times"], data[i]["root"]) for i in mine]
    # packed per-proposal vector as threads.c:544-559: {logl_diff, logpr_diff, count_above, count_below}
    packed = sh.allreduce_sum([sum(lnl), 0.5 * len(mine), len(mine), 1.0], dist)
    q.put((rank, [float(x) for x in packed], [int(i) for i in mine]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_allreduce_of_locus_sums(world):
    import oraclelib as O
    from bpp_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    port += 13*world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = synth.make_dataset(24, 200, 4, "jc69", 1, seed=99)
    total = sum(O.OracleLocus(4, 1, d["seqs"], d["weights"]).full_lnl(d["left"], d["right"], d["times"], d["root"])
                for d in data)
    assert sorted(i for _, _, m in res for i in m) == list(range(24))
    for _, packed, _ in res:
        assert abs(packed[0] - total) <= 1e-12 * abs(total)
        assert packed[1] == 12.0 and packed[2] == 24.0 and packed[3] == float(world)        # (8 ranks: the zig-zag's eight parts of 24 loci)
