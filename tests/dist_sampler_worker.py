"""one rank of the sharded device-resident sampler (tests/test_gpu_dist_sampler.py): its contiguous share of the
loci, the all-loci steps' sum all-reduced through torch.distributed (gloo here so that two ranks can share the
one GPU of the test box; bench.py uses nccl = RCCL)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    rank, world, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
    import torch
    import torch.distributed as dist
    import bpp_amd
    from bpp_amd import synth
    import tape
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    gtr = bool(os.environ.get("DIST_GTR"))           # the generic sampler (8 taxa, GTR + Gamma4) with its parameter moves
    mixed_kinds = os.environ.get("DIST_COMPOSITE")   # loci of several kinds: "both" ranks' shares mixed, or only rank 0's ("one")
    if mixed_kinds:
        # 8 taxa: JC69 loci of the persistent kernel's size with a GTR + Gamma4 locus after every eighth (in the first half only
        # for "one": rank 1's share is then of one kind, a plain sampler next to rank 0's composite)
        fit = synth.make_dataset(136 if mixed_kinds == "one" else 128, 300, 8, "jc69", 1, seed=5)
        gtrs = synth.make_dataset(16, 300, 8, "gtr", 4, seed=7)
        data = []
        for i, d in enumerate(fit):
            data.append(d)
            if i % 8 == 0 and i // 8 < (8 if mixed_kinds == "one" else 16):
                data.append(gtrs[i // 8])
        assert len(data) == 144
        gtr = True                                   # (the 8-taxon species tree below)
    else:
        data = synth.make_dataset(48, 300, 8, "gtr", 4, seed=3) if gtr else synth.make_dataset(160, 300, 4, "jc69", 1, seed=3)
    per = len(data) // world
    first = rank * per
    mine = data[first:first + per]
    eng = bpp_amd.Engine(0)
    loci = tape.make_engine_loci(eng, mine)
    smp = bpp_amd.Sampler(eng, loci, mine, seed=7)
    if os.environ.get("DIST_PROGRAM"):
        # BPP's kernel with the program's THETA / TAU / MIX: two sums per theta in the THETA exchange, five values per TAU
        smp.set_proposal_kernel(1)
        smp.set_program_moves(True, 0.3)
    t = torch.zeros(16, dtype=torch.float64, device="cuda")        # BPA_SAMPLER_SUMS

    def allreduce(ptr, count, stream):
        eng.synchronize()                       # gloo is host-driven: the device sums must be complete
        dist.all_reduce(t[:count])
        torch.cuda.synchronize()
        return True
    p2p = None
    if world > 1 and os.environ.get("DIST_P2P"):
        # the sums exchanged INSIDE the persistent kernel over peer-mapped mailboxes (the handles travel over gloo)
        p2p = bpp_amd.P2P(eng, rank, world, 64)
        handles = [None] * world
        dist.all_gather_object(handles, p2p.handle)
        p2p.connect(handles)
        p2p.set_timeout_ms(20000)                # (two processes time-share the one test GPU)
        smp.set_p2p(p2p, first)
    elif world > 1:
        smp.set_allreduce(allreduce, t.data_ptr(), first)
    if os.environ.get("DIST_MIXED"):
        # three species ((0,1),2); the gene tips A,B,C,D sit in species 0,0,1,2 in the first half of the loci and in
        # 0,1,1,2 in the second: which tip species can hold a coalescence differs between the two ranks' shares
        parent, tau, theta = [3, 3, 4, 4, -1], [0, 0, 0, 0.001, 0.003], [0.002]*5
        smp.set_species_tree(parent, tau, theta)
        for i in range(per):
            smp.set_tip_species(i, [0, 0, 1, 2] if first + i < len(data)//2 else [0, 1, 1, 2])
    else:
        parent, tau, theta = synth.species_tree_arrays(8 if gtr else 4)
        smp.set_species_tree(parent, tau, theta)
    if gtr and not mixed_kinds:
        for i, d in enumerate(mine):
            smp.set_subst_model(i, d["freqs"], d["exch"], 0.5)
        smp.set_subst_moves(0.3, 0.4, 0.8, 1.0, 1.0)
    smp.set_tau_prior(3.0, 1000.0)
    smp.set_theta_prior(2.0, 1000.0, 0.001)
    smp.set_finetune(0.003, 0.005, 0.0008, 0.2)
    smp.initialize()
    ft = None
    if os.environ.get("DIST_BURNIN"):
        ft = smp.burnin(200)                     # the program's step-length rule at the end of 200 iterations (every rank calls it)
    smp.iterate(4 if gtr else 12)
    res = dict(ft=ft, rank=rank, first=first, kind=smp.kind(), taus=smp.taus(), thetas=smp.thetas(), summary=smp.summary(),
               times=[[float(x) for x in smp.tree(i)["time"]] for i in range(per)],
               lnl=[smp.tree(i)["lnl"] for i in range(per)])
    with open(f"{out}.{rank}.json", "w") as f:
        json.dump(res, f)
    smp.close()
    if p2p is not None:
        dist.barrier()
        p2p.close()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
