"""bpp_amd — MI355X-native (HIP, gfx950) per-locus partial-likelihood engine with
the update API of BPP's locus_t / gnode_t hot path.  See DESIGN.md.

The product is libbpp_amd.so (C ABI: include/bpp_amd.h); this package is the
ctypes plumbing around it and the in-tree build driver."""
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
