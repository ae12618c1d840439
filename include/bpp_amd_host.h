/*
 * bpp_amd_host.h — host-side MCMC control in C over the likelihood boundary.
 *
 * BASELINE.json north_star: "host MCMC control flow stays in C and calls HIP through a thin
 * C-ABI shim".  This is that host side, reduced to what drives the likelihood path: a
 * lock-step gene-tree sampler that, like BPP's A00 iteration (method.c:5490-5602), sweeps
 * every locus with gene-node age proposals (GAGE, gtree.c:4585), subtree prune/regraft
 * proposals (GSPR, gtree.c:6531) and an all-loci mixing step (prop_mixing.c:52), keeps the
 * gnode_t fields the reference keeps (left/right/parent/time/clv_index/scaler_index/
 * pmatrix_index), toggles the double buffers before every evaluation exactly as the
 * reference does (SWAP_CLV_INDEX / SWAP_PMAT_INDEX / SWAP_SCALER_INDEX, locus.c:24-26) and
 * toggles back on rejection.  "Step j of every locus" is handed to a likelihood back-end as
 * one batch.  Two back-ends exist: libbpp_amd.so (this repo's product, a00_backend_hip) and,
 * for tests only, the real reference's locus API (oracle/ref_shim.c: ref_backend_eval) —
 * the same driver, the same seeds, the same trajectory on both is the drop-in check.
 *
 * The acceptance rule is Metropolis on the likelihood ratio (times the proposal's Hastings
 * factor for the root-age and mixing multipliers); BPP's MSC prior density is out of scope
 * (SURVEY.md §8f rank 1).
 */
#ifndef BPP_AMD_HOST_H
#define BPP_AMD_HOST_H

#include "bpp_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Random numbers: every locus owns a stream (so that loci can be advanced independently, on the
   host or on the device, with identical results) and the all-loci mixing step a global one.
   64-bit LCG (Knuth MMIX constants); a00_rndu gives a uniform in (0,1).  A GAGE proposal always
   consumes 2 numbers of its locus's stream (proposal, acceptance), a GSPR proposal 3.           */
typedef unsigned long long a00_rng_t;
static inline a00_rng_t a00_rng_seed(unsigned long seed, unsigned stream)
{
  a00_rng_t z = 0x9E3779B97F4A7C15ULL*(a00_rng_t)(stream + 1) ^ (a00_rng_t)seed*0xD1B54A32D192ED03ULL;
  z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 29;
  return z | 1ULL;
}
static inline double a00_rndu(a00_rng_t * r)
{
  *r = *r*6364136223846793005ULL + 1442695040888963407ULL;
  return (double)((*r >> 11) + 0.5)*(1.0/9007199254740992.0);
}
#define A00_GLOBAL_STREAM 0xFFFFFFFFu

/* gene tree of one locus: tips 0..tips-1, inner nodes after; the root node object stays
   the root (gtree.c:6129-6175), so pmatrix indices never collide */
typedef struct a00_tree
{
  int      tips, n, root;
  int *    left, * right, * parent;      /* [n], -1 = none          */
  double * time;                         /* [n] node ages           */
  int *    clv, * pmat, * scaler;        /* [n] current buffer indices (gnode_t fields) */
  double   rate_mui;                     /* gtree_t.rate_mui        */
  double   lnl;                          /* current log-likelihood  */
} a00_tree_t;

/* one proposal step for a set of loci, in node terms */
typedef struct a00_step
{
  unsigned             nloci;
  const unsigned *     locus;            /* [nloci] index of the locus            */
  a00_tree_t * const * tree;             /* [nloci] its tree, proposal installed  */
  const unsigned *     br_off;           /* [nloci+1] */
  const int *          branches;         /* child node ids whose P-matrix changes */
  const unsigned *     nd_off;           /* [nloci+1] */
  const int *          nodes;            /* inner node ids to recompute, children first */
} a00_step_t;

typedef int (*a00_eval_fn)(void * ctx, const a00_step_t * step, double * lnl /* [nloci] */);

typedef struct a00_driver a00_driver_t;

a00_driver_t * a00_create(unsigned nloci, a00_eval_fn eval, void * ctx, unsigned long seed);
void           a00_destroy(a00_driver_t *);
/* install the start tree of locus i (arrays are copied); scaling != 0 gives inner nodes scalers */
int            a00_set_tree(a00_driver_t *, unsigned i, int tips, const int * left, const int * right,
                            const double * times, int root, int scaling);
const a00_tree_t * a00_tree(const a00_driver_t *, unsigned i);
/* species-tree divergence times (ascending) for the TAU step (stree.c:5512 propose_tau): one all-loci
   proposal per tau per iteration; the gene-node ages between the neighbouring taus are rescaled
   ("rubber band", stree.c:4338-4779), the touched branches/root paths re-evaluated
   (gtree_return_partials, gtree.c:145-175) and ONE decision taken from the summed difference
   (threads.c:544-559).  Without taus the iteration has no TAU steps.  n <= 8.                    */
int            a00_set_taus(a00_driver_t *, const double * taus, unsigned n);
unsigned       a00_get_taus(const a00_driver_t *, double * taus);
/* the rubber-band map shared by host and device: new age of a gene node of age t when tau -> tnew,
   with lo / hi the neighbouring taus (hi < 0: none above) */
static inline double a00_rubber_band(double t, double lo, double tau, double tnew, double hi)
{
  if (t > lo && t <= tau) return lo + (t - lo)*(tnew - lo)/(tau - lo);
  if (t > tau && (hi < 0 || t < hi)) return hi < 0 ? tnew + (t - tau) : hi - (hi - t)*(hi - tnew)/(hi - tau);
  return t;
}
static inline double a00_tau_proposal(double u, double lo, double tau, double hi)
{
  return lo + (0.05 + 0.9*u)*((hi < 0 ? 2*tau - lo : hi) - lo);
}
/* start-up evaluation: all matrices, all partials, lnL (method.c:4285-4297) */
int            a00_initialize(a00_driver_t *);
/* one iteration: GAGE over inner nodes, GSPR over non-root nodes, one TAU step per tau, one MIX step */
int            a00_iterate(a00_driver_t *);
double         a00_total_lnl(const a00_driver_t *);
void           a00_counters(const a00_driver_t *, unsigned long * proposals, unsigned long * accepted,
                            unsigned long * steps);

/* likelihood back-end on libbpp_amd.so: loci[i] must have been created with the buffer
   counts of method.c:4110-4146 (2*inner CLVs, 2*edges P-matrices, 2*inner scalers)       */
typedef struct a00_hip_ctx { bpa_engine_t * engine; bpa_locus_t ** loci; } a00_hip_ctx_t;
int a00_backend_hip(void * ctx /* a00_hip_ctx_t* */, const a00_step_t * step, double * lnl);

#ifdef __cplusplus
}
#endif
#endif
