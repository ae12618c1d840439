"""bpp_amd — MI355X-native (HIP, gfx950) per-locus partial-likelihood engine with
the update API of BPP's locus_t / gnode_t hot path.  See DESIGN.md.

The product is libbpp_amd.so (C ABI: include/bpp_amd.h); this package is the
ctypes plumbing around it and the in-tree build driver."""
import os as _os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless set) and streams that share a queue run one
# after the other.  A generic sampler of a large 4-state set runs three part-batches on three streams; with a
# second engine alive in the process (bench.py's other-config sections) the fifth stream landed on a busy queue and config 3
# ran at 188 instead of 235 it/s (NOTES.md 12).  The runtime reads the variable when it starts, so it is set here, before the
# library — and with it the HIP runtime — is loaded; a value the user set is kept.  (The library's first entry points do the same
# for hosts that are not Python: engine.hip, hw_queues_default)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .api import (BpaError, Engine, Locus, Plan, PlanSequence, Sampler, P2P, RcclExchange, GNode, GTree, Op, OP_DTYPE, lib,
                  locus_update_matrices, locus_update_partials, locus_root_loglikelihood,
                  locus_update_all_matrices, locus_update_all_partials,
                  compute_gamma_cats, compress_site_patterns, map_nt, map_aa,
                  DATA_DNA, DATA_AA, MODEL_JC69, MODEL_GTR, MODEL_LG, SCALE_BUFFER_NONE,
                  ATTRIB_ARCH_HIP)

__all__ = ["BpaError", "Engine", "Locus", "Plan", "RcclExchange", "GNode", "GTree", "Op", "OP_DTYPE", "lib",
           "locus_update_matrices", "locus_update_partials", "locus_root_loglikelihood",
           "locus_update_all_matrices", "locus_update_all_partials", "PlanSequence", "Sampler",
           "compute_gamma_cats", "compress_site_patterns", "map_nt", "map_aa"]
