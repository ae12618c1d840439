"""the generic sampler on a small set, one iteration: BPA_GS_DIFF=1 runs every GAGE / GSPR step by both proposal kernels from the
same state and reports what differs; BPA_GS_SYNC=1 names every launch; GS2_LOCI / GS2_TAXA / GS2_MODEL size the set"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["BPA_SMP_GENERIC"] = "1"
import bpp_amd, tape
from bpp_amd import synth
eng = bpp_amd.Engine(0)
N = int(os.environ.get("GS2_LOCI", "300")); TAXA = int(os.environ.get("GS2_TAXA", "4")); MODEL = os.environ.get("GS2_MODEL", "jc69")
data = synth.make_dataset(N, 300, TAXA, MODEL, 4 if MODEL == "gtr" else 1, seed=19)
dev = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=29)
parent, tau0, thetas = synth.species_tree_arrays(TAXA)
dev.set_species_tree(parent, tau0, thetas)
dev.set_tau_prior(3.0, 3.0 / tau0[-1]); dev.set_theta_prior(2.0, 1000.0, 0.001); dev.set_finetune(0.003, 0.005, 0.0008, 0.2)
print("init", file=sys.stderr, flush=True)
dev.initialize()
print("initialized", dev.kind(), file=sys.stderr, flush=True)
dev.iterate(1)
print("iterated", dev.summary(), file=sys.stderr, flush=True)
