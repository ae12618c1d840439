/*
 * a00_driver.c — host-side MCMC control in plain C over the likelihood boundary
 * (include/bpp_amd_host.h): gene trees of all loci under the multispecies coalescent on a fixed
 * species tree, the per-locus loops of BPP's proposals hoisted into lock-step batches ("step j of
 * every locus").  Each move states the reference routine it follows.  Two proposal kernels
 * (a00_set_proposal_kernel): uniform windows on our own 64-bit streams (default; what the device-resident
 * sampler runs too), or BPP's own — legacy_rndu streams and the Bactrian-Laplace window of
 * legacy_rnd_symmetrical (random.c:104-122, 192-238), restated in bpp_amd_host.h.  BPP's trajectory itself
 * is out of reach of ANY batched driver: gtree_propose_ages_serial / gtree_propose_spr_serial
 * (gtree.c:5836 ff.) run locus by locus on ONE generator per thread, so the n-th number of the stream
 * reaches "proposal j of locus i" in locus-major order, while "step j of every locus" — the batch the GPU
 * needs — consumes in proposal-major order; streams here are per locus for that reason.
 */
#define _POSIX_C_SOURCE 199309L     /* clock_gettime (the A00_PROF step profile) */
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "bpp_amd_host.h"

/* section timers of the launch loop: an experimental build's (-DBPA_EXPERIMENTAL), like the library's A/B switches (csrc/device_types.hpp) */
#ifdef BPA_EXPERIMENTAL
#define A00_EXP_SWITCH(name_) getenv(name_)
#else
#define A00_EXP_SWITCH(name_) ((const char *)0)
#endif

#define MAXN 512                          /* nodes per gene tree handled on the stack */

struct a00_driver
{
  unsigned nloci;
  a00_tree_t * trees;
  a00_eval_fn eval;
  void * ctx;
  a00_rng_t * rng;                      /* one stream per locus */
  a00_rng_t grng;                       /* global stream (mixing step) */
  int kernel;                           /* A00_KERNEL_UNIFORM / A00_KERNEL_BPP */
  int pre_valid; double pre_window;     /* the program's moves: the first TAU's window, drawn before the THETA step's numbers (a00_iterate) */
  int program_moves;                    /* A00_KERNEL_BPP: THETA / TAU / MIX as the program runs them (a00_set_program_moves) */
  double theta_slide_prob;              /*   share of sliding-window THETA proposals, the rest Gibbs draws */
  long long run_k[A00_MAXPOP];          /*   k_p and T_p of the current gene trees, from the THETA step's sums on: an accepted */
  double run_T[A00_MAXPOP];             /*   TAU puts the new sums of its three populations in, an accepted MIX multiplies by c */
  int run_ok;
  unsigned int * zrng, gz;              /* A00_KERNEL_BPP: legacy_rndu states, per locus and global */
  /* step scratch */
  unsigned * s_locus; a00_tree_t ** s_tree; unsigned * s_br_off, * s_nd_off;
  int * s_br, * s_nd; size_t cap_br, cap_nd;
  double * s_lnl, * s_hast;
  /* per-locus undo snapshot (whole small tree) */
  int ** u_left, ** u_right, ** u_parent, ** u_clv, ** u_pmat, ** u_scaler; double ** u_time; int * u_root;
  unsigned long proposals, accepted, steps;
  unsigned long gibbs_proposals, gibbs_accepted;     /* of those: THETA Gibbs draws */
  /* species tree (stree->nodes order: tips, then inner populations, children before parents) */
  int S, npop, sp_parent[A00_MAXPOP], sp_left[A00_MAXPOP], sp_right[A00_MAXPOP];
  double tau[A00_MAXPOP], theta[A00_MAXPOP];
  unsigned anc[A00_MAXPOP];             /* bit q: q is p or an ancestor of p (stree->pptable) */
  double ft_gage, ft_gspr, ft_tau, ft_mix;
  double tau_alpha, tau_beta;           /* gamma prior on the root tau (0,0: flat) */
  double theta_alpha, theta_beta, ft_theta;   /* gamma prior on every theta (alpha 0: thetas fixed, no THETA steps) */
  int has_theta[A00_MAXPOP];
  double l2t[A00_MAXPOP], l2t_of[A00_MAXPOP];   /* log(2/theta_p) and the theta it belongs to (tree_logpr_stats) */
  double * s_logpr;                     /* proposed MSC density per slot */
  double * p_logpr, * p_delta; int * p_slot;     /* all-loci steps: per locus */
  int ** u_pop;
  /* substitution-model parameters per locus (a00_set_subst_model): freqs[4] | qrates[6] | alpha, and their moves */
  double * sm; int * sm_ncat; a00_param_fn setpar;
  double ft_freqs, ft_qrates, ft_alpha, alpha_a, alpha_b;
  double * sm_old;                      /* proposed component's old values per slot: [2] for freqs / qrates, alpha */
  /* per-locus staging of a step (a00_set_threads): the loci of a step are proposed in parallel — every draw of a
     per-locus proposal comes from that locus's own stream, so the trajectory does not depend on the thread count —
     each into its own row; `compact` then lines the rows up as the step's slots in locus order */
  int threads, w_cap;                   /* w_cap: ints per locus row (the largest tree's node count) */
  int * w_br, * w_nd, * w_nb, * w_nn;   /* [nloci][w_cap] changed branches / nodes to recompute, and their counts (-1: no proposal) */
  double * w_hast, * w_logpr;
  double * w_diff;                      /* [nloci][A00_MAXPOP] THETA: term differences per locus, summed in locus order afterwards */
  /* two cohorts of loci (a00_set_cohorts): [0, co_split) evaluated through co_ctx[0], the rest through co_ctx[1] — two
     engines, one batch in flight on each.  A per-locus step of cohort B is proposed and marshalled while cohort A's
     launch runs and the other way round; a `view` is a cohort's share of the slot arrays (its loci, its slots from 0). */
  int ncohort; unsigned co_split;
  a00_submit_fn co_submit; a00_wait_fn co_wait; void * co_ctx[2];
  unsigned c_lo, c_hi;                  /* the loci the per-locus loops and `compact` cover: the current view's */
  int cur_view;
  struct a00_view { int * s_br, * s_nd; size_t cap_br, cap_nd; unsigned n; int inflight; } view[3];   /* 0: all loci; 1, 2: the cohorts */
  unsigned * b_locus; a00_tree_t ** b_tree; unsigned * b_br_off, * b_nd_off; double * b_lnl, * b_hast, * b_logpr;   /* the slot arrays as allocated */
  unsigned * t_br_off, * t_nd_off;      /* an all-loci step's second half with its offsets from 0 */
};


static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9*ts.tv_nsec; }

/* the reference's buffer toggles (locus.c:24-26) */
static void swap_clv(a00_tree_t * t, int i)
{
  const int inner = t->tips - 1;
  t->clv[i] = t->tips + (t->clv[i] - t->tips + inner) % (2*inner);
  if (t->scaler[i] != BPA_SCALE_BUFFER_NONE) t->scaler[i] = (t->scaler[i] + inner) % (2*inner);
}
static void swap_pmat(a00_tree_t * t, int i)
{
  const int edges = 2*t->tips - 2;
  t->pmat[i] = (t->pmat[i] + edges) % (2*edges);
}

/* ---- the two proposal kernels (bpp_amd_host.h); stream i >= 0: locus i, i < 0: the global one */
static double draw_u(a00_driver_t * d, long i)
{
  if (d->kernel == A00_KERNEL_BPP) return a00_bpp_rndu(i < 0 ? &d->gz : d->zrng + i);
  return a00_rndu(i < 0 ? &d->grng : d->rng + i);
}
/* a sliding-window step in units of the finetune: uniform on (-1/2, 1/2), or BPP's Bactrian-Laplace variate */
static double draw_window(a00_driver_t * d, long i)
{
  if (d->kernel == A00_KERNEL_BPP) return a00_bpp_rnd_symmetrical(i < 0 ? &d->gz : d->zrng + i);
  return a00_rndu(i < 0 ? &d->grng : d->rng + i) - 0.5;
}
/* a proposal that cannot be made: the uniform kernel keeps the streams in step with the device sampler (a fixed
   number of draws per proposal); BPP draws nothing */
static void skip_u(a00_driver_t * d, long i) { if (d->kernel != A00_KERNEL_BPP) (void)draw_u(d, i); }
/* Metropolis-Hastings acceptance; u0: a uniform drawn in advance (uniform kernel, all-loci steps), < 0 = draw now */
static int accept(a00_driver_t * d, long i, double lnacc, double u0)
{
  if (d->kernel == A00_KERNEL_BPP) return lnacc >= -1e-10 || a00_bpp_rndu(i < 0 ? &d->gz : d->zrng + i) < exp(lnacc);
  if (u0 < 0) u0 = draw_u(d, i);
  return lnacc >= 0 || u0 < exp(lnacc);
}

/* A00_DECLOG=1: every all-loci decision on stderr (what the device samplers are compared with when a trajectory splits) */
static void declog(const char * what, int k, double lnacc, double u, int acc)
{
  static int on = -1;
  if (on < 0) on = getenv("A00_DECLOG") != NULL;
  if (on) fprintf(stderr, "[a00] %s %d lnacc %.17g u %.17g -> %d\n", what, k, lnacc, u, acc);
}

void a00_set_proposal_kernel(a00_driver_t * d, int kind) { d->kernel = kind == A00_KERNEL_BPP ? A00_KERNEL_BPP : A00_KERNEL_UNIFORM; }

void a00_set_program_moves(a00_driver_t * d, int on, double slide_prob)
{ d->program_moves = on != 0; d->theta_slide_prob = slide_prob < 0 ? 0 : slide_prob > 1 ? 1 : slide_prob; }
void a00_gibbs_counters(const a00_driver_t * d, unsigned long * proposals, unsigned long * accepted)
{ if (proposals) *proposals = d->gibbs_proposals; if (accepted) *accepted = d->gibbs_accepted; }

void a00_bpp_kernel_sequence(unsigned int seed, int what, int n, double * out)
{
  int k;
  for (k = 0; k < n; ++k) out[k] = what == 1 ? a00_bpp_rnd_symmetrical(&seed) : what == 2 ? a00_bpp_rndnormal(&seed) : a00_bpp_rndu(&seed);
}
void a00_bpp_gamma_sequence(unsigned int seed, double shape, int n, double * out)
{
  int k;
  for (k = 0; k < n; ++k) out[k] = a00_bpp_rndgamma(&seed, shape);
}
void a00_theta_conditional(double a, double b, long k, double T, double * a1b1) { a00_theta_conditional_invgamma(a, b, k, T, a1b1, a1b1 + 1); }
void a00_theta_conditional_fast(double a, double b, long k, double T, double * a1b1) { a00_theta_conditional_invgamma_fast(a, b, k, T, a1b1, a1b1 + 1); }

static void snapshot(a00_driver_t * d, unsigned i)
{
  a00_tree_t * t = d->trees + i;
  memcpy(d->u_time[i], t->time, (size_t)t->n*(sizeof(double) + 7*sizeof(int)));     /* the whole block (a00_set_tree) */
  d->u_root[i] = t->root;
}
static void restore(a00_driver_t * d, unsigned i)
{
  a00_tree_t * t = d->trees + i;
  memcpy(t->time, d->u_time[i], (size_t)t->n*(sizeof(double) + 7*sizeof(int)));
  t->root = d->u_root[i];
}

a00_driver_t * a00_create(unsigned nloci, a00_eval_fn eval, void * ctx, unsigned long seed)
{
  a00_driver_t * d = (a00_driver_t *)calloc(1, sizeof(*d));
  unsigned q;
  d->nloci = nloci; d->eval = eval; d->ctx = ctx;
  d->rng = (a00_rng_t *)calloc(nloci, sizeof(a00_rng_t));
  for (q = 0; q < nloci; ++q) d->rng[q] = a00_rng_seed(seed, q);
  d->grng = a00_rng_seed(seed, A00_GLOBAL_STREAM);
  /* the legacy_rndu states of A00_KERNEL_BPP: 32 bits of the same seeding function */
  d->zrng = (unsigned int *)calloc(nloci, sizeof(unsigned int));
  for (q = 0; q < nloci; ++q) d->zrng[q] = (unsigned int)(d->rng[q] >> 16);
  d->gz = (unsigned int)(d->grng >> 16);
  d->trees = (a00_tree_t *)calloc(nloci, sizeof(a00_tree_t));
  d->s_locus = (unsigned *)calloc(nloci, sizeof(unsigned));
  d->s_tree = (a00_tree_t **)calloc(nloci, sizeof(a00_tree_t *));
  d->s_br_off = (unsigned *)calloc(nloci + 2, sizeof(unsigned));
  d->s_nd_off = (unsigned *)calloc(nloci + 2, sizeof(unsigned));
  d->t_br_off = (unsigned *)calloc(nloci + 2, sizeof(unsigned));
  d->t_nd_off = (unsigned *)calloc(nloci + 2, sizeof(unsigned));
  d->s_lnl = (double *)calloc(nloci, sizeof(double));
  d->s_hast = (double *)calloc(nloci, sizeof(double));
  d->s_logpr = (double *)calloc(nloci, sizeof(double));
  d->p_logpr = (double *)calloc(nloci, sizeof(double)); d->p_delta = (double *)calloc(nloci, sizeof(double));
  d->p_slot = (int *)calloc(nloci, sizeof(int)); d->u_pop = (int **)calloc(nloci, sizeof(int *));
  d->b_locus = d->s_locus; d->b_tree = d->s_tree; d->b_br_off = d->s_br_off; d->b_nd_off = d->s_nd_off;
  d->b_lnl = d->s_lnl; d->b_hast = d->s_hast; d->b_logpr = d->s_logpr;
  d->c_lo = 0; d->c_hi = nloci;
  d->ft_gage = 0.004; d->ft_gspr = 0.004; d->ft_tau = 0.001; d->ft_mix = 0.3;
  d->threads = 1;
  d->theta_slide_prob = 0.1;
  { const char * ev = getenv("A00_THREADS"); if (ev && atoi(ev) > 0) a00_set_threads(d, atoi(ev)); }
  d->w_nb = (int *)calloc(nloci, sizeof(int)); d->w_nn = (int *)calloc(nloci, sizeof(int));
  d->w_hast = (double *)calloc(nloci, sizeof(double)); d->w_logpr = (double *)calloc(nloci, sizeof(double));
  d->w_diff = (double *)calloc((size_t)nloci*A00_MAXPOP, sizeof(double));
  d->u_left = (int **)calloc(nloci, sizeof(int *)); d->u_right = (int **)calloc(nloci, sizeof(int *));
  d->u_parent = (int **)calloc(nloci, sizeof(int *)); d->u_clv = (int **)calloc(nloci, sizeof(int *));
  d->u_pmat = (int **)calloc(nloci, sizeof(int *)); d->u_scaler = (int **)calloc(nloci, sizeof(int *));
  d->u_time = (double **)calloc(nloci, sizeof(double *)); d->u_root = (int *)calloc(nloci, sizeof(int));
  return d;
}

void a00_destroy(a00_driver_t * d)
{
  unsigned i;
  if (!d) return;
  for (i = 0; i < d->nloci; ++i)
  {
    a00_tree_t * t = d->trees + i;
    free(t->time);                     /* (the locus's block starts with the ages) */
    free(d->u_time[i]);
  }
  d->view[d->cur_view].s_br = d->s_br; d->view[d->cur_view].s_nd = d->s_nd;
  { int v; for (v = 0; v < 3; ++v) { free(d->view[v].s_br); free(d->view[v].s_nd); } }
  free(d->rng); free(d->zrng); free(d->sm); free(d->sm_ncat); free(d->sm_old); free(d->trees); free(d->b_locus); free(d->b_tree); free(d->b_br_off); free(d->b_nd_off);
  free(d->t_br_off); free(d->t_nd_off);
  free(d->b_logpr); free(d->p_logpr); free(d->p_delta); free(d->p_slot); free(d->u_pop);
  free(d->w_br); free(d->w_nd); free(d->w_nb); free(d->w_nn); free(d->w_hast); free(d->w_logpr); free(d->w_diff);
  free(d->b_lnl); free(d->b_hast); free(d->u_left); free(d->u_right); free(d->u_parent);
  free(d->u_clv); free(d->u_pmat); free(d->u_scaler); free(d->u_time); free(d->u_root); free(d);
}

int a00_set_tree(a00_driver_t * d, unsigned i, int tips, const int * left, const int * right,
                 const double * times, int root, int scaling)
{
  a00_tree_t * t = d->trees + i;
  const int n = 2*tips - 1; int k;
  if (i >= d->nloci || n > MAXN || tips < 2) return 0;
  t->tips = tips; t->n = n; t->root = root; t->rate_mui = 1.0; t->lnl = 0;
  /* one block per locus — time | left | right | parent | clv | pmat | scaler | pop — and one of the same shape for the
     roll-back copy: a snapshot is ONE memcpy of 36 n bytes (was eight of separately allocated arrays) */
  {
    const size_t bytes = (size_t)n*(sizeof(double) + 7*sizeof(int));
    char * blk, * ublk;
    free(t->time); free(d->u_time[i]);                    /* (a tree set again) */
    t->time = NULL; d->u_time[i] = NULL;
    blk = (char *)malloc(bytes); ublk = (char *)malloc(bytes);
    int * ip = (int *)(blk + (size_t)n*sizeof(double)), * up = (int *)(ublk + (size_t)n*sizeof(double));
    if (!blk || !ublk) { free(blk); free(ublk); return 0; }
    t->time = (double *)blk; t->left = ip; t->right = ip + n; t->parent = ip + 2*n; t->clv = ip + 3*n; t->pmat = ip + 4*n;
    t->scaler = ip + 5*n; t->pop = ip + 6*n;
    d->u_time[i] = (double *)ublk; d->u_left[i] = up; d->u_right[i] = up + n; d->u_parent[i] = up + 2*n; d->u_clv[i] = up + 3*n;
    d->u_pmat[i] = up + 4*n; d->u_scaler[i] = up + 5*n; d->u_pop[i] = up + 6*n;
    memcpy(t->left, left, (size_t)n*sizeof(int)); memcpy(t->right, right, (size_t)n*sizeof(int)); memcpy(t->time, times, (size_t)n*sizeof(double));
  }
  for (k = 0; k < n; ++k) t->pop[k] = k < tips ? k : -1;
  for (k = 0; k < n; ++k) t->parent[k] = -1;
  for (k = 0; k < n; ++k)
  {
    if (left[k] >= 0) { t->parent[left[k]] = k; t->parent[right[k]] = k; }
    t->clv[k] = k; t->pmat[k] = k;                               /* gtree.c:2433-2439, 2398 */
    t->scaler[k] = (scaling && k >= tips) ? k - tips : BPA_SCALE_BUFFER_NONE;
  }
  return 1;
}

const a00_tree_t * a00_tree(const a00_driver_t * d, unsigned i) { return d->trees + i; }

/* ---- step assembly */
static void reserve(a00_driver_t * d, size_t br, size_t nd)
{
  if (br > d->cap_br) { d->cap_br = 2*br + 1024; d->s_br = (int *)realloc(d->s_br, d->cap_br*sizeof(int)); }
  if (nd > d->cap_nd) { d->cap_nd = 2*nd + 1024; d->s_nd = (int *)realloc(d->s_nd, d->cap_nd*sizeof(int)); }
}

static int marshal_threads = 1;      /* a00_backend_hip's marshalling loop (its context carries no driver): the last a00_set_threads */
void a00_set_threads(a00_driver_t * d, int threads) { d->threads = threads > 0 ? threads : 1; marshal_threads = d->threads; }

/* rows of the per-locus staging: sized by the largest tree, (re)allocated when a larger one arrives */
static int staging_ready(a00_driver_t * d)
{
  unsigned i; int cap = 0;
  for (i = 0; i < d->nloci; ++i) if (d->trees[i].n > cap) cap = d->trees[i].n;
  if (cap <= d->w_cap && d->w_br) return 1;
  free(d->w_br); free(d->w_nd);
  d->w_cap = cap;
  d->w_br = (int *)malloc((size_t)d->nloci*(size_t)cap*sizeof(int));
  d->w_nd = (int *)malloc((size_t)d->nloci*(size_t)cap*sizeof(int));
  return d->w_br && d->w_nd;
}

/* install a proposal on locus i: toggle the buffers of the changed branches and of the nodes
   to recompute (sorted children-first = by age) and leave both lists in the locus's staging row.
   Touches nothing but locus i: safe from any thread. */
static void install_local(a00_driver_t * d, unsigned i, const int * branches, int nb, int * nodes, int nn,
                          double hast, double logpr)
{
  a00_tree_t * t = d->trees + i; int a, b;
  /* unique + sort nodes by age (a parent is always older than its children) */
  for (a = 0; a < nn; ++a) for (b = a + 1; b < nn; ++b) if (nodes[b] == nodes[a]) { nodes[b] = nodes[--nn]; --b; }
  for (a = 1; a < nn; ++a) { int v = nodes[a]; for (b = a; b > 0 && t->time[nodes[b-1]] > t->time[v]; --b) nodes[b] = nodes[b-1]; nodes[b] = v; }
  for (a = 0; a < nb; ++a) swap_pmat(t, branches[a]);
  for (a = 0; a < nn; ++a) swap_clv(t, nodes[a]);
  memcpy(d->w_br + (size_t)i*(size_t)d->w_cap, branches, (size_t)nb*sizeof(int));
  memcpy(d->w_nd + (size_t)i*(size_t)d->w_cap, nodes, (size_t)nn*sizeof(int));
  d->w_nb[i] = nb; d->w_nn[i] = nn; d->w_hast[i] = hast; d->w_logpr[i] = logpr;
}

/* the slot arrays of view v (0: all loci; 1, 2: the cohorts): a cohort's slots start at its first locus's place in the
   arrays (its offsets one further, past the other cohort's end mark), its branch / node lists are its own */
static void set_view(a00_driver_t * d, int v)
{
  struct a00_view * o = d->view + d->cur_view, * w = d->view + v;
  const unsigned lo = v == 2 ? d->co_split : 0u;
  o->s_br = d->s_br; o->s_nd = d->s_nd; o->cap_br = d->cap_br; o->cap_nd = d->cap_nd;
  d->s_br = w->s_br; d->s_nd = w->s_nd; d->cap_br = w->cap_br; d->cap_nd = w->cap_nd;
  d->cur_view = v;
  d->c_lo = lo; d->c_hi = v == 1 ? d->co_split : d->nloci;
  d->s_locus = d->b_locus + lo; d->s_tree = d->b_tree + lo; d->s_lnl = d->b_lnl + lo; d->s_hast = d->b_hast + lo; d->s_logpr = d->b_logpr + lo;
  d->s_br_off = d->b_br_off + lo + (v == 2); d->s_nd_off = d->b_nd_off + lo + (v == 2);
}

/* the staged rows as the step's slots, in locus order; p_slot[i] = the slot of locus i or -1; returns the slot count */
static unsigned compact(a00_driver_t * d)
{
  unsigned i, n = 0; long li;
  /* slots and offsets: a running count in locus order (cheap, serial) ... */
  d->s_br_off[0] = d->s_nd_off[0] = 0;
  for (i = d->c_lo; i < d->c_hi; ++i)
  {
    d->p_slot[i] = -1;
    if (d->w_nb[i] < 0) continue;
    d->s_br_off[n+1] = d->s_br_off[n] + (unsigned)d->w_nb[i];
    d->s_nd_off[n+1] = d->s_nd_off[n] + (unsigned)d->w_nn[i];
    d->p_slot[i] = (int)n++;
  }
  reserve(d, d->s_br_off[n], d->s_nd_off[n]);
  /* ... then every row to its place */
#pragma omp parallel for schedule(static) num_threads(d->threads) if (d->threads > 1)
  for (li = (long)d->c_lo; li < (long)d->c_hi; ++li)
  {
    const int sl = d->p_slot[li];
    if (sl < 0) continue;
    memcpy(d->s_br + d->s_br_off[sl], d->w_br + (size_t)li*(size_t)d->w_cap, (size_t)d->w_nb[li]*sizeof(int));
    memcpy(d->s_nd + d->s_nd_off[sl], d->w_nd + (size_t)li*(size_t)d->w_cap, (size_t)d->w_nn[li]*sizeof(int));
    d->s_locus[sl] = (unsigned)li; d->s_tree[sl] = d->trees + li; d->s_hast[sl] = d->w_hast[li]; d->s_logpr[sl] = d->w_logpr[li];
  }
  return n;
}

static void view_step(const a00_driver_t * d, unsigned n, a00_step_t * s)
{
  s->nloci = n; s->locus = d->s_locus; s->tree = d->s_tree; s->br_off = d->s_br_off; s->branches = d->s_br;
  s->nd_off = d->s_nd_off; s->nodes = d->s_nd;
}

static int step_eval(a00_driver_t * d, unsigned n)
{
  a00_step_t s;
  if (!n) return 1;
  view_step(d, n, &s);
  d->steps++;
  if (d->ncohort == 2 && d->cur_view == 0)
  {
    /* an all-loci step (slots in locus order): each cohort's slots to its engine, both in flight together */
    a00_step_t s1; unsigned n0 = 0, j; int r0 = 2, r1 = 2;
    while (n0 < n && d->s_locus[n0] < d->co_split) ++n0;
    for (j = n0; j <= n; ++j) { d->t_br_off[j - n0] = d->s_br_off[j] - d->s_br_off[n0]; d->t_nd_off[j - n0] = d->s_nd_off[j] - d->s_nd_off[n0]; }
    s1.nloci = n - n0; s1.locus = d->s_locus + n0; s1.tree = d->s_tree + n0; s1.br_off = d->t_br_off; s1.branches = d->s_br + d->s_br_off[n0];
    s1.nd_off = d->t_nd_off; s1.nodes = d->s_nd + d->s_nd_off[n0];
    s.nloci = n0;
    if (n0 && !(r0 = d->co_submit(d->co_ctx[0], &s, d->s_lnl))) return 0;
    if (n - n0 && !(r1 = d->co_submit(d->co_ctx[1], &s1, d->s_lnl + n0))) { if (r0 == 1) (void)d->co_wait(d->co_ctx[0], d->s_lnl, n0); return 0; }
    if (r0 == 1 && !d->co_wait(d->co_ctx[0], d->s_lnl, n0)) { if (r1 == 1) (void)d->co_wait(d->co_ctx[1], d->s_lnl + n0, n - n0); return 0; }
    if (r1 == 1 && !d->co_wait(d->co_ctx[1], d->s_lnl + n0, n - n0)) return 0;
    return 1;
  }
  return d->eval(d->ctx, &s, d->s_lnl);
}

static int path_to_root(const a00_tree_t * t, int v, int * out)
{
  int k = 0;
  for (; v >= 0; v = t->parent[v]) out[k++] = v;
  return k;
}

/* ------------------------------------------------------------------ species tree and MSC --- */
int a00_set_species_tree(a00_driver_t * d, int species, const int * parent, const double * tau, const double * theta)
{
  int p, q, np = 2*species - 1;
  if (species < 1 || np > A00_MAXPOP) return 0;
  for (p = 0; p < np; ++p)
  {
    if (p == np - 1 ? parent[p] != -1 : (parent[p] <= p || parent[p] >= np || parent[p] < species)) return 0;
    if (!(theta[p] > 0) || (p < species ? tau[p] != 0 : !(tau[p] > 0))) return 0;
    if (parent[p] >= 0 && !(tau[parent[p]] > tau[p])) return 0;
  }
  d->S = species; d->npop = np;
  for (p = 0; p < np; ++p) { d->sp_parent[p] = parent[p]; d->sp_left[p] = d->sp_right[p] = -1; d->tau[p] = tau[p]; d->theta[p] = theta[p]; }
  for (p = 0; p < np - 1; ++p) { q = parent[p]; if (d->sp_left[q] < 0) d->sp_left[q] = p; else if (d->sp_right[q] < 0) d->sp_right[q] = p; else return 0; }
  for (p = species; p < np; ++p) if (d->sp_right[p] < 0) return 0;
  for (p = 0; p < np; ++p) { d->anc[p] = 0; for (q = p; q >= 0; q = parent[q]) d->anc[p] |= 1u << q; }
  return 1;
}

int a00_set_tip_species(a00_driver_t * d, unsigned i, const int * species)
{
  a00_tree_t * t = d->trees + i; int k;
  if (i >= d->nloci || !t->pop) return 0;
  for (k = 0; k < t->tips; ++k) { if (species[k] < 0 || species[k] >= d->S) return 0; t->pop[k] = species[k]; }
  return 1;
}

void a00_set_tau_prior(a00_driver_t * d, double alpha, double beta) { d->tau_alpha = alpha; d->tau_beta = beta; }
void a00_set_theta_prior(a00_driver_t * d, double alpha, double beta, double finetune)
{ d->theta_alpha = alpha; d->theta_beta = beta; d->ft_theta = finetune; }
unsigned a00_get_thetas(const a00_driver_t * d, double * theta)
{
  int p;
  for (p = 0; p < d->npop; ++p) theta[p] = d->theta[p];
  return (unsigned)d->npop;
}

/* log prior ratio of the taus when the root tau goes old -> new, the others keeping their place in
   (0, root): gamma(alpha, beta) on the root, uniform Dirichlet below it (propose_tau, stree.c:5655-5657:
   (alpha - 1 - candidate_count + 1) log(new/old) - beta (new - old), candidate_count = inner populations) */
static double root_tau_prior_ratio(const a00_driver_t * d, double oldt, double newt)
{
  if (!(d->tau_alpha > 0)) return 0;
  return (d->tau_alpha - 1 - (d->S - 1) + 1)*log(newt/oldt) - d->tau_beta*(newt - oldt);
}

void a00_set_finetune(a00_driver_t * d, double gage, double gspr, double tau, double mix)
{ d->ft_gage = gage; d->ft_gspr = gspr; d->ft_tau = tau; d->ft_mix = mix; }

unsigned a00_get_taus(const a00_driver_t * d, double * tau)
{
  int p;
  for (p = 0; p < d->npop; ++p) tau[p] = d->tau[p];
  return (unsigned)d->npop;
}

/* youngest population that is an ancestor (or self) of both */
static int lca_pop(const a00_driver_t * d, int p, int q)
{
  while (!((d->anc[q] >> p) & 1u)) p = d->sp_parent[p];
  return p;
}
/* the ancestor (or self) of population p that holds time t (gtree.c:4790-4797) */
static int climb(const a00_driver_t * d, int p, double t)
{
  while (d->sp_parent[p] >= 0 && d->tau[d->sp_parent[p]] <= t) p = d->sp_parent[p];
  return p;
}

/* gtree_logprob (gtree.c:3957): the sum over populations, in stree->nodes order, of
   gtree_update_logprob_contrib; NAN if the tree does not fit the species tree */
static double tree_logpr_stats(const a00_driver_t * d, const a00_tree_t * t, int * nc_out, double * t2h_out)
{
  int nin[A00_MAXPOP], nc[A00_MAXPOP], p, k; double times[MAXN], logpr = 0;
  for (p = 0; p < d->npop; ++p) nin[p] = 0;
  for (k = 0; k < t->tips; ++k) nin[t->pop[k]]++;
  for (p = 0; p < d->npop; ++p)
  {
    int n = 0, a, b;
    if (p >= d->S) nin[p] = (nin[d->sp_left[p]] - nc[d->sp_left[p]]) + (nin[d->sp_right[p]] - nc[d->sp_right[p]]);
    for (k = t->tips; k < t->n; ++k)
      if (t->pop[k] == p)
      {
        const double v = t->time[k];
        if (v < d->tau[p] || (d->sp_parent[p] >= 0 && v >= d->tau[d->sp_parent[p]])) return NAN;
        for (a = n++; a > 0 && times[a-1] > v; --a) times[a] = times[a-1];
        times[a] = v; (void)b;
      }
    nc[p] = n;
    if (n >= nin[p] && n > 0) return NAN;
    {
      const double T2h = a00_msc_t2h(d->tau[p], d->sp_parent[p] >= 0 ? d->tau[d->sp_parent[p]] : -1.0, nin[p], times, n);
      if (nc_out) { nc_out[p] = n; t2h_out[p] = T2h; }
      /* a00_msc_term(n, T2h, theta, 1) with log(2/theta) taken from a cache keyed by the theta it was made from (thetas
         stand still during a step's per-locus loop: every locus would take the same logarithm again) */
      {
        const double th = d->theta[p]; double term = 0;
        if (n)
        {
          a00_driver_t * dm = (a00_driver_t *)d;
          if (dm->l2t_of[p] != th) { dm->l2t[p] = log(2.0/(1.0*th)); dm->l2t_of[p] = th; }      /* (threads race to store the same value) */
          term += n*dm->l2t[p];
        }
        if (T2h) term -= T2h/(th*1.0);
        logpr += term;
      }
    }
  }
  return logpr;
}
static double tree_logpr(const a00_driver_t * d, const a00_tree_t * t) { return tree_logpr_stats(d, t, NULL, NULL); }

double a00_locus_logpr(const a00_driver_t * d, unsigned i) { return tree_logpr(d, d->trees + i); }

/* populations of the inner nodes from topology and ages; 0 if an age is below the tau of the
   common population of its children */
static int assign_pops(const a00_driver_t * d, a00_tree_t * t)
{
  int order[MAXN], n = 0, a, b, k;
  for (k = t->tips; k < t->n; ++k) order[n++] = k;
  for (a = 1; a < n; ++a) { int v = order[a]; for (b = a; b > 0 && t->time[order[b-1]] > t->time[v]; --b) order[b] = order[b-1]; order[b] = v; }
  for (a = 0; a < n; ++a)
  {
    const int v = order[a], c = lca_pop(d, t->pop[t->left[v]], t->pop[t->right[v]]);
    if (t->time[v] < d->tau[c]) return 0;
    t->pop[v] = climb(d, c, t->time[v]);
  }
  return 1;
}

int a00_initialize(a00_driver_t * d)
{
  unsigned i; int br[MAXN], nd[MAXN];
  if (!d->npop) return 0;                                /* a00_set_species_tree first */
  /* populations that can hold a coalescence: the inner ones, and a species with two sequences in some locus */
  { int p; for (p = 0; p < d->npop; ++p) d->has_theta[p] = p >= d->S; }
  for (i = 0; i < d->nloci; ++i)
  {
    int cnt[A00_MAXPOP] = {0}, k;
    for (k = 0; k < d->trees[i].tips; ++k) if (++cnt[d->trees[i].pop[k]] >= 2) d->has_theta[d->trees[i].pop[k]] = 1;
  }
  if (!staging_ready(d)) return 0;
  for (i = 0; i < d->nloci; ++i)
  {
    a00_tree_t * t = d->trees + i; int nb = 0, nn = 0, k;
    if (!assign_pops(d, t)) return 0;
    t->logpr = tree_logpr(d, t);
    if (t->logpr != t->logpr) return 0;
    for (k = 0; k < t->n; ++k) { if (t->parent[k] >= 0) br[nb++] = k; if (t->left[k] >= 0) nd[nn++] = k; }
    /* start-up evaluates into the current buffers: toggle twice = no toggle */
    for (k = 0; k < nb; ++k) swap_pmat(t, br[k]);
    for (k = 0; k < nn; ++k) swap_clv(t, nd[k]);
    install_local(d, i, br, nb, nd, nn, 0.0, t->logpr);
  }
  if (compact(d) != d->nloci || !step_eval(d, d->nloci)) return 0;
  for (i = 0; i < d->nloci; ++i) d->trees[i].lnl = d->s_lnl[i];
  return 1;
}

/* per-locus Metropolis-Hastings decision on  MSC density x likelihood  (gtree.c:5476-5480) */
static void decide(a00_driver_t * d, unsigned n)
{
  long s; unsigned long acc = 0;
#pragma omp parallel for schedule(static) num_threads(d->threads) reduction(+:acc) if (d->threads > 1)
  for (s = 0; s < (long)n; ++s)
  {
    const unsigned i = d->s_locus[s]; a00_tree_t * t = d->trees + i;
    const double lnacc = (d->s_logpr[s] - t->logpr) + (d->s_lnl[s] - t->lnl) + d->s_hast[s];
    if (accept(d, (long)i, lnacc, -1.0)) { t->lnl = d->s_lnl[s]; t->logpr = d->s_logpr[s]; ++acc; }
    else restore(d, i);                                  /* swap indices, ages, populations, topology back */
  }
  d->proposals += n; d->accepted += acc;
}

/* GAGE: the k-th inner node of every locus (propose_ages, gtree.c:4585-5532, the MSC branch) */
/* the decision of locus i's pending per-locus proposal (slot p_slot[i] of the view's last step), as `decide` takes it */
static int settle_one(a00_driver_t * d, unsigned i)
{
  const int s = d->p_slot[i]; a00_tree_t * t = d->trees + i;
  if (s < 0) return 0;
  if (accept(d, (long)i, (d->s_logpr[s] - t->logpr) + (d->s_lnl[s] - t->lnl) + d->s_hast[s], -1.0)) { t->lnl = d->s_lnl[s]; t->logpr = d->s_logpr[s]; return 1; }
  restore(d, i);
  return 0;
}

/* settle != 0: the locus's pending decision first, in the same pass (the cohort pipeline: one parallel region instead of two) */
static unsigned long gage_propose(a00_driver_t * d, int k, int settle)
{
  long li; unsigned long acc = 0;
#pragma omp parallel for schedule(static) num_threads(d->threads) reduction(+:acc) if (d->threads > 1)
  for (li = (long)d->c_lo; li < (long)d->c_hi; ++li)
  {
    const unsigned i = (unsigned)li;
    a00_tree_t * t = d->trees + i; int v = -1, c = 0, j, nb = 0, nn, p, l, r, br[4], nd[MAXN]; double lo, hi, u, tnew;
    if (settle) acc += (unsigned long)settle_one(d, i);
    d->w_nb[i] = -1;
    for (j = 0; j < t->n; ++j) if (t->left[j] >= 0 && c++ == k) { v = j; break; }
    if (v < 0) continue;
    u = draw_window(d, (long)i);
    l = t->left[v]; r = t->right[v]; p = t->parent[v];
    lo = fmax(t->time[l], t->time[r]);
    if (t->pop[l] != t->pop[r]) lo = fmax(lo, d->tau[lca_pop(d, t->pop[l], t->pop[r])]);
    hi = p >= 0 ? t->time[p] : 999.0;
    if (!(hi > lo)) { skip_u(d, (long)i); continue; }
    snapshot(d, i);
    tnew = a00_reflect(t->time[v] + d->ft_gage*u, lo, hi);
    t->time[v] = tnew;
    t->pop[v] = climb(d, t->pop[l], tnew);
    br[nb++] = l; br[nb++] = r; if (p >= 0) br[nb++] = v;
    nn = path_to_root(t, v, nd);
    install_local(d, i, br, nb, nd, nn, 0.0, tree_logpr(d, t));
  }
  return acc;
}

/* exchange the tree positions of node ids a and b (buffer indices stay with the ids) */
static void swap_ids(a00_tree_t * t, int a, int b)
{
  int i; int L[MAXN], R[MAXN], P[MAXN], Q[MAXN]; double T[MAXN];
#define M(x) ((x) == a ? b : (x) == b ? a : (x))
  for (i = 0; i < t->n; ++i)
  {
    const int o = M(i);
    L[i] = t->left[o] >= 0 ? M(t->left[o]) : -1; R[i] = t->right[o] >= 0 ? M(t->right[o]) : -1;
    P[i] = t->parent[o] >= 0 ? M(t->parent[o]) : -1; T[i] = t->time[o]; Q[i] = t->pop[o];
  }
  memcpy(t->left, L, (size_t)t->n*sizeof(int)); memcpy(t->right, R, (size_t)t->n*sizeof(int));
  memcpy(t->parent, P, (size_t)t->n*sizeof(int)); memcpy(t->time, T, (size_t)t->n*sizeof(double));
  memcpy(t->pop, Q, (size_t)t->n*sizeof(int));
  t->root = M(t->root);
#undef M
}

static int count_tips(const a00_tree_t * t, int v)
{
  return t->left[v] < 0 ? 1 : count_tips(t, t->left[v]) + count_tips(t, t->right[v]);
}

/* GSPR: the k-th non-root node of every locus is pruned and regrafted (propose_spr, gtree.c:6531-7610,
   the MSC branch with the plain target choice) */
static unsigned long gspr_propose(a00_driver_t * d, int k, int settle)
{
  long li; unsigned long acc = 0;
#pragma omp parallel for schedule(static) num_threads(d->threads) reduction(+:acc) if (d->threads > 1)
  for (li = (long)d->c_lo; li < (long)d->c_hi; ++li)
  {
    const unsigned i = (unsigned)li;
    a00_tree_t * t = d->trees + i;
    int a = -1, c = 0, j, p, s, g, pc, tgt, ntg = 0, nsrc = 1, targets[MAXN], pop0, popt, leaves, gl[A00_MAXPOP];
    int bset[4], br[4], nb = 0, nd[2*MAXN], nn = 0, root_before;
    double lo, tnew, u1, u2;
    if (settle) acc += (unsigned long)settle_one(d, i);
    d->w_nb[i] = -1;
    for (j = 0; j < t->n; ++j) if (j != t->root && c++ == k) { a = j; break; }
    if (a < 0) continue;
    u1 = draw_window(d, (long)i); u2 = draw_u(d, (long)i);
    p = t->parent[a]; s = t->left[p] == a ? t->right[p] : t->left[p]; g = t->parent[p];
    /* youngest population from a's upwards that holds gene tips outside a's subtree (gtree.c:6664-6669) */
    for (j = 0; j < d->npop; ++j) gl[j] = 0;
    for (j = 0; j < t->tips; ++j) { int q; for (q = t->pop[j]; q >= 0; q = d->sp_parent[q]) gl[q]++; }
    leaves = count_tips(t, a);
    for (pop0 = t->pop[a]; gl[pop0] <= leaves && d->sp_parent[pop0] >= 0; pop0 = d->sp_parent[pop0]) ;
    lo = fmax(t->time[a], d->tau[pop0]);
    tnew = a00_reflect(t->time[p] + d->ft_gspr*u1, lo, 999.0);
    popt = climb(d, t->pop[a], tnew);
    /* targets: the branches crossing tnew inside popt; above the root only the root */
    if (tnew >= t->time[t->root]) targets[ntg++] = t->root;
    else
      for (j = 0; j < t->n; ++j)
        if (j != a && j != t->root && t->time[j] <= tnew && t->time[t->parent[j]] > tnew && ((d->anc[t->pop[j]] >> popt) & 1u))
          targets[ntg++] = j == p ? s : j;
    /* sources: the branches the reverse move could pick at the old age (gtree.c:6760-6775) */
    if (p != t->root)
      for (j = 0; j < t->n; ++j)
        if (j != a && j != t->root && j != s && j != p && t->time[j] <= t->time[p] && t->time[t->parent[j]] > t->time[p] &&
            ((d->anc[t->pop[j]] >> t->pop[p]) & 1u))
          ++nsrc;
    if (!ntg) { skip_u(d, (long)i); continue; }
    tgt = targets[(int)(u2*ntg) % ntg];
    if (tgt == p) tgt = s;                                /* the father is the root and stays it: only its age moves */
    snapshot(d, i);
    root_before = t->root;
    /* prune: the sibling takes p's place */
    t->parent[s] = g;
    if (g >= 0) { if (t->left[g] == p) t->left[g] = s; else t->right[g] = s; } else t->root = s;
    /* regraft p (with a below it) on the branch above tgt, at age tnew in population popt */
    pc = t->parent[tgt];
    t->time[p] = tnew; t->pop[p] = popt;
    t->left[p] = a; t->right[p] = tgt; t->parent[a] = p; t->parent[tgt] = p; t->parent[p] = pc;
    if (pc >= 0) { if (t->left[pc] == tgt) t->left[pc] = p; else t->right[pc] = p; } else t->root = p;
    nn = path_to_root(t, p, nd);
    if (g >= 0) nn += path_to_root(t, g, nd + nn);
    bset[0] = a; bset[1] = tgt; bset[2] = p; bset[3] = s;
    if (t->root != root_before)
    {
      /* the root node object keeps its identity (gtree.c:6129-6175) */
      const int newtop = t->root;
      swap_ids(t, newtop, root_before);
      for (j = 0; j < nn; ++j) nd[j] = nd[j] == newtop ? root_before : nd[j] == root_before ? newtop : nd[j];
      for (j = 0; j < 4; ++j) bset[j] = bset[j] == newtop ? root_before : bset[j] == root_before ? newtop : bset[j];
      nn += path_to_root(t, newtop, nd + nn);
    }
    for (j = 0; j < 4; ++j)
    {
      int q, dup = 0;
      for (q = 0; q < nb; ++q) if (br[q] == bset[j]) dup = 1;
      if (!dup && t->parent[bset[j]] >= 0) br[nb++] = bset[j];
    }
    install_local(d, i, br, nb, nd, nn, log((double)ntg/(double)nsrc), tree_logpr(d, t));
  }
  return acc;
}

/* settle the cohorts' launches in flight: results, then the per-locus decisions */
static int flush_cohorts(a00_driver_t * d)
{
  int c, ok = 1;
  if (d->ncohort != 2) return 1;
  for (c = 0; c < 2; ++c)
  {
    struct a00_view * w = d->view + 1 + c;
    if (!w->inflight) continue;
    set_view(d, 1 + c);
    if (w->inflight == 1 && !d->co_wait(d->co_ctx[c], d->s_lnl, w->n)) ok = 0;
    if (ok) decide(d, w->n);
    w->inflight = 0;
  }
  set_view(d, 0);
  return ok;
}

/* an all-loci step with cohorts: each cohort's share is staged in its view and sent off as soon as it is there (the
   other cohort is proposed for meanwhile); cohort_collect waits for both.  The slots of a view sit at its first locus's
   place in the arrays: slot_lnl finds a locus's result in either form. */
static int cohort_submit(a00_driver_t * d, int c, unsigned n)
{
  struct a00_view * w = d->view + 1 + c; a00_step_t s;
  w->n = n; w->inflight = 0;
  if (!n) return 1;
  view_step(d, n, &s);
  d->steps++;
  return (w->inflight = d->co_submit(d->co_ctx[c], &s, d->s_lnl)) != 0;
}

static int cohort_collect(a00_driver_t * d)
{
  int c, ok = 1;
  for (c = 0; c < 2; ++c)
  {
    struct a00_view * w = d->view + 1 + c;
    set_view(d, 1 + c);
    if (w->inflight == 1 && !d->co_wait(d->co_ctx[c], d->s_lnl, w->n)) ok = 0;
    w->inflight = 0;
  }
  set_view(d, 0);
  return ok;
}

static double slot_lnl(const a00_driver_t * d, unsigned i)
{ return d->b_lnl[(d->ncohort == 2 && i >= d->co_split ? d->co_split : 0u) + (unsigned)d->p_slot[i]]; }

/* one per-locus step (kind 0: GAGE k, 1: GSPR k).  With cohorts: for each cohort in turn — collect its previous step
   (whose launch ran while the other cohort was being proposed for), decide it, propose this step, send it off. */
static int per_locus_step(a00_driver_t * d, int kind, int k)
{
  unsigned n; int c;
  if (!staging_ready(d)) return 0;
  if (d->ncohort != 2)
  {
    if (kind) (void)gspr_propose(d, k, 0); else (void)gage_propose(d, k, 0);
    n = compact(d);
    if (!step_eval(d, n)) return 0;
    decide(d, n);
    return 1;
  }
  for (c = 0; c < 2; ++c)
  {
    struct a00_view * w = d->view + 1 + c;
    a00_step_t s; int r;
    static int prof = -1; static double tp[5]; static unsigned calls;
    double t0, t1, t2, t3, t4;
    if (prof < 0) prof = A00_EXP_SWITCH("A00_PROF") != NULL;
    set_view(d, 1 + c);
    t0 = prof ? now_s() : 0;
    {
      const int settle = w->inflight != 0; unsigned long acc;
      if (w->inflight == 1 && !d->co_wait(d->co_ctx[c], d->s_lnl, w->n)) { w->inflight = 0; set_view(d, 0); return 0; }
      t1 = t2 = prof ? now_s() : 0;
      /* the pending decisions and the new proposals of the cohort's loci in ONE pass: a locus's decision, then its proposal */
      acc = kind ? gspr_propose(d, k, settle) : gage_propose(d, k, settle);
      if (settle) { d->proposals += w->n; d->accepted += acc; }
      w->inflight = 0;
    }
    t3 = prof ? now_s() : 0;
    w->n = n = compact(d);
    t4 = prof ? now_s() : 0;
    if (!n) continue;
    view_step(d, n, &s);
    d->steps++;
    if (!(r = d->co_submit(d->co_ctx[c], &s, d->s_lnl))) { set_view(d, 0); return 0; }
    w->inflight = r;                                  /* 1: in flight, 2: evaluated already (the lnL are there) */
    if (prof)
    {
      const double t5 = now_s();
      tp[0] += t1 - t0; tp[1] += t2 - t1; tp[2] += t3 - t2; tp[3] += t4 - t3; tp[4] += t5 - t4;
      if (++calls % 260 == 0)
      {
        fprintf(stderr, "[a00] per cohort step: wait %.3f ms, decide %.3f, propose %.3f, compact %.3f, submit %.3f\n",
                1e3*tp[0]/260, 1e3*tp[1]/260, 1e3*tp[2]/260, 1e3*tp[3]/260, 1e3*tp[4]/260);
        tp[0] = tp[1] = tp[2] = tp[3] = tp[4] = 0;
      }
    }
  }
  set_view(d, 0);
  return 1;
}

int a00_set_cohorts(a00_driver_t * d, unsigned split, a00_submit_fn submit, a00_wait_fn wait, void * ctx0, void * ctx1)
{
  if (!submit || !wait || split == 0 || split >= d->nloci) { d->ncohort = 0; return split == 0; }
  if (!flush_cohorts(d)) return 0;
  d->ncohort = 2; d->co_split = split; d->co_submit = submit; d->co_wait = wait; d->co_ctx[0] = ctx0; d->co_ctx[1] = ctx1;
  return 1;
}

/* THETA: every population that can hold a coalescence gets a sliding-window proposal (reflected at 0) with a
   gamma(alpha, beta) prior; only the MSC density changes — no likelihood call.  Given the gene trees the thetas are
   independent (the density is a sum of per-population terms), so all are proposed from the same state and each is
   decided on its own  sum over loci of [term(theta') - term(theta)] + prior ratio  (stree.c:3464-3560 family). */
static int theta_step_gibbs(a00_driver_t * d);
static int theta_step_all(a00_driver_t * d)
{
  unsigned i; int p; long li;
  double tnew[A00_MAXPOP], uacc[A00_MAXPOP], sum[A00_MAXPOP];
  if (d->kernel == A00_KERNEL_BPP && d->program_moves) return theta_step_gibbs(d);
  for (p = 0; p < d->npop; ++p)
  {
    sum[p] = 0; tnew[p] = d->theta[p];
    if (!d->has_theta[p]) continue;
    tnew[p] = a00_reflect(d->theta[p] + d->ft_theta*draw_window(d, -1), 0.0, 999.0);
    uacc[p] = d->kernel == A00_KERNEL_BPP ? -1.0 : draw_u(d, -1);
  }
#pragma omp parallel for schedule(static) num_threads(d->threads) if (d->threads > 1)
  for (li = 0; li < (long)d->nloci; ++li)
  {
    int nc[A00_MAXPOP], pp; double t2h[A00_MAXPOP];
    (void)tree_logpr_stats(d, d->trees + li, nc, t2h);
    for (pp = 0; pp < d->npop; ++pp)
      if (d->has_theta[pp])
        d->w_diff[(size_t)li*A00_MAXPOP + pp] = a00_msc_term(nc[pp], t2h[pp], tnew[pp], 1.0) - a00_msc_term(nc[pp], t2h[pp], d->theta[pp], 1.0);
  }
  for (i = 0; i < d->nloci; ++i)                          /* the sums in locus order, whatever the thread count */
    for (p = 0; p < d->npop; ++p)
      if (d->has_theta[p]) sum[p] += d->w_diff[(size_t)i*A00_MAXPOP + p];
  for (p = 0; p < d->npop; ++p)
  {
    double lnacc;
    if (!d->has_theta[p]) continue;
    lnacc = sum[p] + ((d->theta_alpha - 1)*log(tnew[p]/d->theta[p]) - d->theta_beta*(tnew[p] - d->theta[p]));
    d->proposals++;
    { const int acc_ = tnew[p] > 0 && accept(d, -1, lnacc, uacc[p]); declog("theta", p, lnacc, uacc[p], acc_); if (acc_) { d->accepted++; d->theta[p] = tnew[p]; } }
  }
#pragma omp parallel for schedule(static) num_threads(d->threads) if (d->threads > 1)
  for (li = 0; li < (long)d->nloci; ++li) d->trees[li].logpr = tree_logpr(d, d->trees + li);
  return 1;
}

/* k_p = coalescences in p, T_p = sum of T2h (2^-40 fixed point) over all loci, for the populations with a theta;
   returns 0 when a term is unusable (the device's rule: |T2h| >= 256) */
static int theta_sums(a00_driver_t * d, long long * ksum, long long * tsum)
{
  int p, bad = 0; long li;
  for (p = 0; p < d->npop; ++p) { ksum[p] = 0; tsum[p] = 0; }
#pragma omp parallel num_threads(d->threads) if (d->threads > 1)
  {
    long long kl[A00_MAXPOP], tl[A00_MAXPOP]; int pp, badl = 0;
    for (pp = 0; pp < d->npop; ++pp) { kl[pp] = 0; tl[pp] = 0; }
#pragma omp for schedule(static)
    for (li = 0; li < (long)d->nloci; ++li)
    {
      int nc[A00_MAXPOP]; double t2h[A00_MAXPOP];
      (void)tree_logpr_stats(d, d->trees + li, nc, t2h);
      for (pp = 0; pp < d->npop; ++pp)
        if (d->has_theta[pp])
        {
          if (!(fabs(t2h[pp]) < 256.0)) badl = 1;
          else { kl[pp] += nc[pp]; tl[pp] += llrint(t2h[pp]*1099511627776.0); }
        }
    }
#pragma omp critical
    { for (pp = 0; pp < d->npop; ++pp) { ksum[pp] += kl[pp]; tsum[pp] += tl[pp]; } bad |= badl; }
  }
  return !bad;
}

/* THETA the program's way (a00_set_program_moves, A00_KERNEL_BPP): per theta, in population order, the sliding
   window with probability slide_prob, else the metropolized Gibbs draw of bpp_amd_host.h — both decided from
   k_p = sum over loci of the coalescences in p and T_p = sum of T2h (2^-40 fixed point: no order).  Global stream:
   the choices (and the windows of the sliding ones) of all populations first, then per population the Gibbs variate
   and the acceptance number (drawn only when needed) — the order the device kernel can follow with ONE exchange.   */
static int theta_step_gibbs(a00_driver_t * d)
{
  int p, slide[A00_MAXPOP], bad = 0; long li;
  double tnew[A00_MAXPOP], fa[A00_MAXPOP], fb[A00_MAXPOP];
  long long ksum[A00_MAXPOP], tsum[A00_MAXPOP];
  for (p = 0; p < d->npop; ++p)
  {
    slide[p] = 0; tnew[p] = d->theta[p];
    if (!d->has_theta[p]) continue;
    slide[p] = a00_bpp_rndu(&d->gz) < d->theta_slide_prob;
    if (slide[p]) tnew[p] = a00_reflect(d->theta[p] + d->ft_theta*draw_window(d, -1), 0.0, 999.0);
  }
  bad = !theta_sums(d, ksum, tsum);
  d->run_ok = !bad;
  for (p = 0; p < d->npop; ++p) { d->run_k[p] = ksum[p]; d->run_T[p] = (double)tsum[p]*(1.0/1099511627776.0); }
  /* the Gibbs variates of all populations first (the device takes them side by side), then the decisions */
  for (p = 0; p < d->npop; ++p)
  {
    fa[p] = fb[p] = NAN;
    if (!d->has_theta[p] || bad || slide[p]) continue;
    a00_theta_conditional_invgamma(d->theta_alpha, d->theta_beta, (long)ksum[p], d->run_T[p], &fa[p], &fb[p]);
    if (fa[p] == fa[p]) tnew[p] = 1/(a00_bpp_rndgamma(&d->gz, fa[p])/fb[p]);
  }
  for (p = 0; p < d->npop; ++p)
  {
    double lnacc = NAN, T; int acc_ = 0;
    if (!d->has_theta[p]) continue;
    d->proposals++;
    T = d->run_T[p];
    if (!bad)
    {
      if (slide[p]) lnacc = a00_theta_lnacc((long)ksum[p], T, d->theta[p], tnew[p], d->theta_alpha, d->theta_beta);
      else if (fa[p] == fa[p])
        lnacc = a00_theta_lnacc((long)ksum[p], T, d->theta[p], tnew[p], d->theta_alpha, d->theta_beta)
              + a00_theta_gibbs_hastings(fa[p], fb[p], d->theta[p], tnew[p]);
      acc_ = lnacc == lnacc && tnew[p] > 0 && accept(d, -1, lnacc, -1.0);
    }
    declog(slide[p] ? "theta" : "thetag", p, lnacc, -1.0, acc_);
    if (acc_) { d->accepted++; d->theta[p] = tnew[p]; if (!slide[p]) d->gibbs_accepted++; }
    if (!slide[p]) d->gibbs_proposals++;
  }
#pragma omp parallel for schedule(static) num_threads(d->threads) if (d->threads > 1)
  for (li = 0; li < (long)d->nloci; ++li) d->trees[li].logpr = tree_logpr(d, d->trees + li);
  return 1;
}

/* TAU of inner population q: sliding window reflected into (older child's tau, parent's tau); the gene
   nodes of q and of its two children between those bounds move with it (rubber band,
   propose_tau_update_gtrees stree.c:4338-4479); ONE decision for all loci from
   sum(dlogpr + dlnL) + below*log(minfactor) + above*log(maxfactor)   (stree.c:6280) */
static int tau_step(a00_driver_t * d, int q)
{
  unsigned i, n = 0; long li; double sum = 0; int acc_, j, bad = 0, co;
  const int nco = d->ncohort == 2 ? 2 : 1;
  const int cl = d->sp_left[q], cr = d->sp_right[q], pq = d->sp_parent[q];
  /* the program's rubber band also re-draws the thetas of q and its two children (opt_rb_theta_update = 1, bpp.c:618;
     propose_tau, stree.c:5840-5990): each from the inverse-gamma fitted to its conditional given k_p and the sum of the
     T2h AFTER the move; the densities' change over all loci then follows from the sums, the loci contribute their
     likelihood change and their three new T2h */
  const int program = d->kernel == A00_KERNEL_BPP && d->program_moves && d->theta_alpha > 0;
  const int aff[3] = { q, cl, cr };
  long long cnew[3] = { 0, 0, 0 };
  double oldtheta[3];
  const double old = d->tau[q], lo = fmax(d->tau[cl], d->tau[cr]), hi = pq >= 0 ? d->tau[pq] : 999.0;
  const double tnew = a00_reflect(old + d->ft_tau*(d->pre_valid ? d->pre_window : draw_window(d, -1)), lo, hi);
  const double uacc = d->kernel == A00_KERNEL_BPP ? -1.0 : draw_u(d, -1);
  const double minf = (tnew - lo)/(old - lo), maxf = (tnew - hi)/(old - hi), lminf = log(minf), lmaxf = log(maxf);
  d->pre_valid = 0;
  if (!staging_ready(d)) return 0;
  d->tau[q] = tnew;
  for (co = 0; co < nco; ++co)
  {
  if (nco == 2) set_view(d, 1 + co);
#pragma omp parallel for schedule(static) num_threads(d->threads) if (d->threads > 1)
  for (li = (long)d->c_lo; li < (long)d->c_hi; ++li)
  {
    const unsigned i = (unsigned)li;
    a00_tree_t * t = d->trees + i; int br[MAXN], nd[MAXN], nb = 0, nn = 0, k, v, above = 0, below = 0;
    char isbr[MAXN], isnd[MAXN];
    d->w_nb[i] = -1;
    snapshot(d, i);
    memset(isbr, 0, (size_t)t->n); memset(isnd, 0, (size_t)t->n);
    for (k = t->tips; k < t->n; ++k)
    {
      const int pk = t->pop[k]; const double tk = t->time[k];
      if ((pk != q && pk != cl && pk != cr) || tk < lo || tk > hi) continue;
      if (tk >= old) { t->time[k] = hi + maxf*(tk - hi); ++above; } else { t->time[k] = lo + minf*(tk - lo); ++below; }
      isbr[t->left[k]] = isbr[t->right[k]] = 1; if (t->parent[k] >= 0) isbr[k] = 1;
      for (v = k; v >= 0; v = t->parent[v]) isnd[v] = 1;              /* gtree_return_partials, gtree.c:145-175 */
    }
    if (program)
    {
      int nc[A00_MAXPOP]; double t2h[A00_MAXPOP]; int jj;
      d->p_logpr[i] = tree_logpr_stats(d, t, nc, t2h);
      for (jj = 0; jj < 3; ++jj) d->w_diff[(size_t)i*A00_MAXPOP + jj] = d->p_logpr[i] == d->p_logpr[i] ? t2h[aff[jj]] : NAN;
      d->p_delta[i] = below*lminf + above*lmaxf;
    }
    else
    {
      d->p_logpr[i] = tree_logpr(d, t);
      d->p_delta[i] = (d->p_logpr[i] - t->logpr) + below*lminf + above*lmaxf;
    }
    if (!(above + below)) continue;
    for (k = 0; k < t->n; ++k) { if (isbr[k]) br[nb++] = k; if (isnd[k]) nd[nn++] = k; }
    install_local(d, i, br, nb, nd, nn, 0.0, d->p_logpr[i]);
  }
  n = compact(d);
  if (nco == 2 && !cohort_submit(d, co, n)) { set_view(d, 0); return 0; }
  }
  if (nco == 2) { if (!cohort_collect(d)) return 0; }
  else if (!step_eval(d, n)) return 0;
  for (i = 0; i < d->nloci; ++i)
    sum += d->p_slot[i] >= 0 ? (slot_lnl(d, i) - d->trees[i].lnl) + d->p_delta[i] : d->p_delta[i];
  if (pq < 0) sum += root_tau_prior_ratio(d, old, tnew);
  if (program)
  {
    for (i = 0; i < d->nloci; ++i)
      for (j = 0; j < 3; ++j)
      {
        const double x = d->w_diff[(size_t)i*A00_MAXPOP + j];
        if (!(fabs(x) < 256.0)) bad = 1; else cnew[j] += llrint(x*1099511627776.0);
      }
    for (j = 0; j < 3; ++j)
    {
      const int p = aff[j]; double a1, b1, a1o, b1o, g, tn, Cn; const double to = d->theta[p]; const long k = (long)d->run_k[p];
      oldtheta[j] = to;
      if (!d->has_theta[p]) continue;
      if (bad || !d->run_ok) { sum = NAN; continue; }
      Cn = (double)cnew[j]*(1.0/1099511627776.0);
      a00_theta_conditional_invgamma(d->theta_alpha, d->theta_beta, k, Cn, &a1, &b1);
      a00_theta_conditional_invgamma(d->theta_alpha, d->theta_beta, k, d->run_T[p], &a1o, &b1o);
      if (!(a1 == a1 && a1o == a1o)) { sum = NAN; continue; }
      g = a00_bpp_rndgamma(&d->gz, a1);
      tn = 1.0/(g/b1);
      sum += (a00_invgamma_logpdf(to, a1o, b1o) - a00_invgamma_logpdf(tn, a1, b1))
           + ((d->theta_alpha - 1)*log(tn/to) - d->theta_beta*(tn - to))
           + (k*(log(2.0/tn) - log(2.0/to)) - (Cn/tn - d->run_T[p]/to));
      d->theta[p] = tn;
    }
  }
  d->proposals++;
  acc_ = accept(d, -1, sum, uacc);
  declog("tau", q, sum, uacc, acc_);
  if (acc_)
  {
    d->accepted++;
    if (program)
    {
      for (j = 0; j < 3; ++j) if (d->has_theta[aff[j]]) d->run_T[aff[j]] = (double)cnew[j]*(1.0/1099511627776.0);
#pragma omp parallel for schedule(static) num_threads(d->threads) if (d->threads > 1)
      for (li = 0; li < (long)d->nloci; ++li) d->p_logpr[li] = tree_logpr(d, d->trees + li);        /* (with the new thetas) */
    }
#pragma omp parallel for schedule(static) num_threads(d->threads) if (d->threads > 1)
    for (li = 0; li < (long)d->nloci; ++li) { d->trees[li].logpr = d->p_logpr[li]; if (d->p_slot[li] >= 0) d->trees[li].lnl = slot_lnl(d, (unsigned)li); }
  }
  else
  {
    d->tau[q] = old;
    if (program) for (j = 0; j < 3; ++j) d->theta[aff[j]] = oldtheta[j];
#pragma omp parallel for schedule(static) num_threads(d->threads) if (d->threads > 1)
    for (li = 0; li < (long)d->nloci; ++li) if (d->p_slot[li] >= 0) restore(d, (unsigned)li);
  }
  return 1;
}

/* MIX: every gene-node age of every locus and every tau times c; ONE decision from
   sum(dlogpr + dlnL) + (ages + taus)*log c   (prop_mixing.c:203-205; thetas stay) */
static int mix_step(a00_driver_t * d)
{
  unsigned i; long li; int p, acc_, co; double sum = 0, lnacc, oldtau[A00_MAXPOP], oldtheta[A00_MAXPOP], lnacc_theta = 0;
  const int nco = d->ncohort == 2 ? 2 : 1;
  /* log c: finetune x BPP's window variate with its kernel (prop_mixing.c:300), uniform with ours */
  const double lnc = d->ft_mix*(d->kernel == A00_KERNEL_BPP ? draw_window(d, -1) : draw_u(d, -1) - 0.5), c = exp(lnc);
  const double uacc = d->kernel == A00_KERNEL_BPP ? -1.0 : draw_u(d, -1);
  const int program = d->kernel == A00_KERNEL_BPP && d->program_moves && d->theta_alpha > 0;
  if (!staging_ready(d)) return 0;
  for (p = 0; p < d->npop; ++p) oldtheta[p] = d->theta[p];
  if (program)
  {
    /* the thetas from their conditionals given the scaled trees (prop_mixing.c:272-425), in population order; the
       densities' change over all loci follows from the sums: k (log 2/theta' - log 2/theta) - (c T/theta' - T/theta) */
    for (p = 0; p < d->npop; ++p)
    {
      double a1, b1, a1o, b1o, Ts, g, tn; const double to = oldtheta[p]; const long k = (long)d->run_k[p];
      if (!d->has_theta[p]) continue;
      if (!d->run_ok) { lnacc_theta = NAN; continue; }
      Ts = d->run_T[p]*c;
      a00_theta_conditional_invgamma(d->theta_alpha, d->theta_beta, k, Ts, &a1, &b1);
      a00_theta_conditional_invgamma(d->theta_alpha, d->theta_beta, k, Ts/c, &a1o, &b1o);
      if (!(a1 == a1 && a1o == a1o)) { lnacc_theta = NAN; continue; }
      g = a00_bpp_rndgamma(&d->gz, a1);
      tn = 1.0/(g/b1);
      lnacc_theta += (a00_invgamma_logpdf(to, a1o, b1o) - a00_invgamma_logpdf(tn, a1, b1))
                   + ((d->theta_alpha - 1)*log(tn/to) - d->theta_beta*(tn - to))
                   + (k*(log(2.0/tn) - log(2.0/to)) - (Ts/tn - d->run_T[p]/to));
      d->theta[p] = tn;
    }
  }
  for (p = 0; p < d->npop; ++p) { oldtau[p] = d->tau[p]; d->tau[p] *= c; }
  for (co = 0; co < nco; ++co)
  {
  if (nco == 2) set_view(d, 1 + co);
#pragma omp parallel for schedule(static) num_threads(d->threads) if (d->threads > 1)
  for (li = (long)d->c_lo; li < (long)d->c_hi; ++li)
  {
    const unsigned i = (unsigned)li;
    a00_tree_t * t = d->trees + i; int br[MAXN], nd[MAXN], nb = 0, nn = 0, k;
    snapshot(d, i);
    for (k = 0; k < t->n; ++k)
    {
      if (t->left[k] >= 0) { t->time[k] *= c; nd[nn++] = k; }
      if (t->parent[k] >= 0) br[nb++] = k;
    }
    d->p_logpr[i] = tree_logpr(d, t);
    d->p_delta[i] = (program ? 0.0 : d->p_logpr[i] - t->logpr) + (double)nn*lnc;
    install_local(d, i, br, nb, nd, nn, 0.0, d->p_logpr[i]);
  }
  if (compact(d) != d->c_hi - d->c_lo) { set_view(d, 0); return 0; }       /* every locus has a slot: slot i = locus i (a cohort's slots start at its first locus's place) */
  if (nco == 2 && !cohort_submit(d, co, d->c_hi - d->c_lo)) { set_view(d, 0); return 0; }
  }
  if (nco == 2) { if (!cohort_collect(d)) return 0; }
  else if (!step_eval(d, d->nloci)) return 0;
  for (i = 0; i < d->nloci; ++i) sum += (d->b_lnl[i] - d->trees[i].lnl) + d->p_delta[i];
  lnacc = sum + (double)(d->S - 1)*lnc;
  if (d->tau_alpha > 0)                    /* all taus scale together: the Dirichlet part is unchanged */
    lnacc += (d->tau_alpha - 1)*lnc - d->tau_beta*(d->tau[d->npop-1] - oldtau[d->npop-1]) - (double)(d->S - 2)*lnc;
  lnacc += lnacc_theta;
  d->proposals++;
  acc_ = accept(d, -1, lnacc, uacc);
  declog("mix", 0, lnacc, uacc, acc_);
  if (acc_)
  {
    d->accepted++;
#pragma omp parallel for schedule(static) num_threads(d->threads) if (d->threads > 1)
    for (li = 0; li < (long)d->nloci; ++li) { d->trees[li].lnl = d->b_lnl[li]; d->trees[li].logpr = d->p_logpr[li]; }
    if (program) for (p = 0; p < d->npop; ++p) d->run_T[p] *= c;
  }
  else
  {
    for (p = 0; p < d->npop; ++p) { d->tau[p] = oldtau[p]; d->theta[p] = oldtheta[p]; }
#pragma omp parallel for schedule(static) num_threads(d->threads) if (d->threads > 1)
    for (li = 0; li < (long)d->nloci; ++li) restore(d, (unsigned)li);
  }
  return 1;
}

/* ---- substitution-parameter moves (bpp_amd_host.h) */
void a00_set_param_backend(a00_driver_t * d, a00_param_fn fn) { d->setpar = fn; }
void a00_set_subst_moves(a00_driver_t * d, double ft_freqs, double ft_qrates, double ft_alpha, double alpha_a, double alpha_b)
{ d->ft_freqs = ft_freqs; d->ft_qrates = ft_qrates; d->ft_alpha = ft_alpha; d->alpha_a = alpha_a; d->alpha_b = alpha_b; }

int a00_set_subst_model(a00_driver_t * d, unsigned i, const double * freqs, const double * qrates, double alpha, int ncat)
{
  double * m;
  if (i >= d->nloci || ncat < 0 || ncat > 16) return 0;
  if (!d->sm)
  {
    d->sm = (double *)calloc((size_t)d->nloci*11, sizeof(double));
    d->sm_ncat = (int *)calloc(d->nloci, sizeof(int));
    d->sm_old = (double *)calloc((size_t)d->nloci*2, sizeof(double));
  }
  m = d->sm + (size_t)i*11;
  memcpy(m, freqs, 4*sizeof(double)); memcpy(m + 4, qrates, 6*sizeof(double)); m[10] = alpha;
  d->sm_ncat[i] = ncat;
  return 1;
}
int a00_get_subst_model(const a00_driver_t * d, unsigned i, double * freqs, double * qrates, double * alpha)
{
  const double * m;
  if (i >= d->nloci || !d->sm) return 0;
  m = d->sm + (size_t)i*11;
  if (freqs) memcpy(freqs, m, 4*sizeof(double));
  if (qrates) memcpy(qrates, m + 4, 6*sizeof(double));
  if (alpha) *alpha = m[10];
  return 1;
}
/* hand the current values of component `which` of locus i to the likelihood back-end */
static int push_param(a00_driver_t * d, unsigned i, int which)
{
  double * m = d->sm + (size_t)i*11;
  void * ctx = d->ncohort == 2 ? d->co_ctx[i >= d->co_split] : d->ctx;
  if (which == 1) return d->setpar(ctx, i, 1, m, 4);
  if (which == 2) return d->setpar(ctx, i, 2, m + 4, 6);
  {
    double rates[16];
    const int nc = d->sm_ncat[i];
    if (nc < 2) return 1;
    if (!bpa_compute_gamma_cats(m[10], m[10], (unsigned)nc, rates)) return 0;       /* prop_gamma.c:93-97 */
    return d->setpar(ctx, i, 4, rates, (unsigned)nc);
  }
}

/* one component (which = 1 frequency j, 2 exchangeability j, 4 alpha) of every locus: propose, full recomputation, decide */
static int param_step(a00_driver_t * d, int which, int j)
{
  unsigned i, n = 0, s;
  const int ref = which == 1 ? 3 : 1;                      /* T / the A<->G rate (locus.c:2791, 3222) */
  if (!staging_ready(d)) return 0;
  for (i = 0; i < d->nloci; ++i) d->w_nb[i] = -1;
  for (i = 0; i < d->nloci; ++i)
  {
    a00_tree_t * t = d->trees + i; double * m = d->sm + (size_t)i*11;
    int br[MAXN], nd[MAXN], nb = 0, nn = 0, k;
    double hast;
    if (which == 4 && d->sm_ncat[i] < 2) continue;
    if (which == 4)
    {
      const double a_old = m[10], la_old = log(a_old);
      const double la_new = a00_reflect(la_old + d->ft_alpha*draw_window(d, (long)i), -99.0, 99.0);
      const double a_new = exp(la_new);
      d->sm_old[2*n] = a_old;
      m[10] = a_new;
      hast = (la_new - la_old) + ((d->alpha_a - 1)*log(a_new/a_old) - d->alpha_b*(a_new - a_old));      /* prop_gamma.c:72, 133 */
    }
    else
    {
      double * v = which == 1 ? m : m + 4;
      const double sum = v[j] + v[ref], lo = log(1e-5), hi = log(sum);
      const double l_old = log(v[j]);
      const double l_new = a00_reflect(l_old + (which == 1 ? d->ft_freqs : d->ft_qrates)*draw_window(d, (long)i), lo, hi);
      d->sm_old[2*n] = v[j]; d->sm_old[2*n+1] = v[ref];
      v[j] = exp(l_new); v[ref] = sum - v[j];
      hast = l_new - l_old;                                                                              /* locus.c:2867, 3296 */
    }
    if (!push_param(d, i, which)) return 0;
    snapshot(d, i);
    for (k = 0; k < t->n; ++k)
    {
      if (t->left[k] >= 0) nd[nn++] = k;
      if (t->parent[k] >= 0) br[nb++] = k;
    }
    install_local(d, i, br, nb, nd, nn, hast, t->logpr);
    ++n;                                           /* (sm_old is indexed by the slot this locus is about to get) */
  }
  if (compact(d) != n || !step_eval(d, n)) return 0;
  for (s = 0; s < n; ++s)
  {
    const unsigned li = d->s_locus[s]; a00_tree_t * t = d->trees + li; double * m = d->sm + (size_t)li*11;
    const double lnacc = (d->s_lnl[s] - t->lnl) + d->s_hast[s];
    d->proposals++;
    if (accept(d, (long)li, lnacc, -1.0)) { t->lnl = d->s_lnl[s]; d->accepted++; continue; }
    restore(d, li);
    if (which == 4) m[10] = d->sm_old[2*s];
    else { double * v = which == 1 ? m : m + 4; v[j] = d->sm_old[2*s]; v[ref] = d->sm_old[2*s+1]; }
    if (!push_param(d, li, which)) return 0;
  }
  return 1;
}

int a00_iterate(a00_driver_t * d)
{
  unsigned i; int k, maxtips = 0;
  for (i = 0; i < d->nloci; ++i) if (d->trees[i].tips > maxtips) maxtips = d->trees[i].tips;
  for (k = 0; k < maxtips - 1; ++k)   if (!per_locus_step(d, 0, k)) return 0;
  for (k = 0; k < 2*maxtips - 2; ++k) if (!per_locus_step(d, 1, k)) return 0;
  if (!flush_cohorts(d)) return 0;
  /* the program's moves: the first TAU's window comes before the THETA step's numbers in the global stream — the device
     kernel makes that TAU's proposal at the loci right after the sweep and brings its sums with THETA's in ONE exchange */
  if (d->kernel == A00_KERNEL_BPP && d->program_moves && d->theta_alpha > 0 && d->S < d->npop)
  {
    int p, any = 0;
    for (p = 0; p < d->npop; ++p) any |= d->has_theta[p];
    if (any) { d->pre_window = draw_window(d, -1); d->pre_valid = 1; }
  }
  if (d->theta_alpha > 0 && !theta_step_all(d)) return 0;
  for (k = d->S; k < d->npop; ++k)    if (!tau_step(d, k)) return 0;
  if (!mix_step(d)) return 0;
  /* the substitution-parameter moves come last (method.c:5699-5735) */
  if (d->sm && d->setpar)
  {
    if (d->ft_freqs > 0)  for (k = 0; k < 3; ++k) if (!param_step(d, 1, k)) return 0;
    if (d->ft_qrates > 0) for (k = 0; k < 6; ++k) if (k != 1 && !param_step(d, 2, k)) return 0;
    if (d->ft_alpha > 0 && !param_step(d, 4, 0)) return 0;
  }
  return 1;
}

int a00_backend_prior(void * ctx, const a00_step_t * step, double * lnl)
{
  unsigned i;
  (void)ctx;
  for (i = 0; i < step->nloci; ++i) lnl[i] = 0;
  return 1;
}

double a00_total_lnl(const a00_driver_t * d)
{
  double s = 0; unsigned i;
  for (i = 0; i < d->nloci; ++i) s += d->trees[i].lnl;
  return s;
}

void a00_counters(const a00_driver_t * d, unsigned long * proposals, unsigned long * accepted, unsigned long * steps)
{
  if (proposals) *proposals = d->proposals;
  if (accepted) *accepted = d->accepted;
  if (steps) *steps = d->steps;
}

/* ------------------------------------------------------------------------------------------
 * back-end on libbpp_amd.so: node terms -> explicit buffer indices (what locus_update_matrices
 * / locus_update_partials read off gnode_t, locus.c:2350, 2549-2569) -> one batched launch
 * ---------------------------------------------------------------------------------------- */
/* A00_PROF=1: where a step's wall time goes (marshalling here / bpa_batch_evaluate), printed every 130 steps */

static int backend_hip_run(void * vctx, const a00_step_t * s, double * lnl, int async)
{
  static int prof = -1; static double t_marsh = 0, t_eval = 0, t_last = 0, t_between = 0; static unsigned calls = 0;
  const double t0 = (prof < 0 ? (prof = A00_EXP_SWITCH("A00_PROF") != NULL) : prof) ? now_s() : 0;
  a00_hip_ctx_t * c = (a00_hip_ctx_t *)vctx;
  const unsigned n = s->nloci, nbr = s->br_off[n], nnd = s->nd_off[n];
  unsigned i, j; int ok;
  bpa_locus_t ** loci = (bpa_locus_t **)malloc(n*sizeof(*loci));
  unsigned * mp = (unsigned *)malloc((nbr + 1)*sizeof(unsigned));
  double * ml = (double *)malloc((nbr + 1)*sizeof(double));
  bpa_op_t * ops = (bpa_op_t *)malloc((nnd + 1)*sizeof(bpa_op_t));
  unsigned * rc = (unsigned *)malloc(n*sizeof(unsigned));
  int * rs = (int *)malloc(n*sizeof(int));
  bpa_batch_t b;
  if (!loci || !mp || !ml || !ops || !rc || !rs)
  {
    free(loci); free(mp); free(ml); free(ops); free(rc); free(rs);
    return 0;
  }
#pragma omp parallel for schedule(static) private(j) num_threads(marshal_threads) if (marshal_threads > 1)
  for (i = 0; i < n; ++i)
  {
    const a00_tree_t * t = s->tree[i];
    loci[i] = c->loci[s->locus[i]];
    for (j = s->br_off[i]; j < s->br_off[i+1]; ++j)
    {
      const int x = s->branches[j];
      mp[j] = (unsigned)t->pmat[x];
      ml[j] = (t->time[t->parent[x]] - t->time[x])*t->rate_mui;          /* locus.c:2350 */
    }
    for (j = s->nd_off[i]; j < s->nd_off[i+1]; ++j)
    {
      const int x = s->nodes[j], l = t->left[x], r = t->right[x];
      ops[j].parent_clv = (unsigned)t->clv[x]; ops[j].parent_scaler = t->scaler[x];
      ops[j].left_clv = (unsigned)t->clv[l];   ops[j].left_pmatrix = (unsigned)t->pmat[l];  ops[j].left_scaler = t->scaler[l];
      ops[j].right_clv = (unsigned)t->clv[r];  ops[j].right_pmatrix = (unsigned)t->pmat[r]; ops[j].right_scaler = t->scaler[r];
    }
    rc[i] = (unsigned)t->clv[t->root]; rs[i] = t->scaler[t->root];
  }
  b.nloci = n; b.loci = loci; b.mat_off = s->br_off; b.mat_pmatrix = mp; b.mat_length = ml;
  b.op_off = s->nd_off; b.ops = ops; b.root_clv = rc; b.root_scaler = rs;
  {
    const double t1 = prof ? now_s() : 0;
    /* the record image of the step: the worker threads write it, a share of the loci each (bpa_batch_fill) */
    const int how = marshal_threads > 1 ? bpa_batch_begin(c->engine, &b) : 2;
    if (how == 1)
    {
      const int parts = marshal_threads; int q;
      static double tb[3]; static unsigned nb_;
      const double u0 = prof ? now_s() : 0; double u1, u2;
#pragma omp parallel for schedule(static) num_threads(marshal_threads)
      for (q = 0; q < parts; ++q)
        (void)bpa_batch_fill(c->engine, &b, (unsigned)((unsigned long long)n*(unsigned)q/(unsigned)parts), (unsigned)((unsigned long long)n*(unsigned)(q + 1)/(unsigned)parts));
      u1 = prof ? now_s() : 0;
      ok = async ? bpa_batch_end_async(c->engine, &b) : bpa_batch_end(c->engine, &b, lnl);
      if (prof)
      {
        u2 = now_s(); tb[0] += u0 - t1; tb[1] += u1 - u0; tb[2] += u2 - u1;
        if (++nb_ % 260 == 0) { fprintf(stderr, "[a00] per batch: begin %.3f ms, fill %.3f, end %.3f\n", 1e3*tb[0]/260, 1e3*tb[1]/260, 1e3*tb[2]/260); tb[0] = tb[1] = tb[2] = 0; }
      }
      if (ok == 2) ok = bpa_batch_evaluate(c->engine, &b, lnl) ? (async ? 2 : 1) : 0;          /* (not the one-image path after all) */
    }
    else ok = how == 2 ? (bpa_batch_evaluate(c->engine, &b, lnl) ? (async ? 2 : 1) : 0) : 0;
    if (prof)
    {
      const double t2 = now_s();
      if (t_last > 0) t_between += t0 - t_last;
      t_marsh += t1 - t0; t_eval += t2 - t1; t_last = t2;
      if (++calls % 130 == 0)
        fprintf(stderr, "[a00] per step: driver %.3f ms, marshalling %.3f ms, bpa_batch_evaluate %.3f ms\n",
                1e3*t_between/calls, 1e3*t_marsh/calls, 1e3*t_eval/calls);
    }
  }
  free(loci); free(mp); free(ml); free(ops); free(rc); free(rs);
  return ok;
}

int a00_backend_hip(void * vctx, const a00_step_t * s, double * lnl) { return backend_hip_run(vctx, s, lnl, 0); }

/* the wait of a backend whose evaluation function is its own submit (it returns with the lnL in place) */
int a00_backend_wait_none(void * ctx, double * lnl, unsigned n) { (void)ctx; (void)lnl; (void)n; return 1; }

/* the cohort form (a00_set_cohorts): the step's image is written and sent, the launch queued — 1: bpa_batch_wait will have
   the lnL (a00_backend_hip_wait), 2: the batch took the general path and lnl is filled already, 0: error */
int a00_backend_hip_submit(void * vctx, const a00_step_t * s, double * lnl) { return backend_hip_run(vctx, s, lnl, 1); }
int a00_backend_hip_wait(void * vctx, double * lnl, unsigned n)
{
  a00_hip_ctx_t * c = (a00_hip_ctx_t *)vctx;
  (void)n;
  return bpa_batch_wait(c->engine, lnl);
}

/* the driver's substitution-parameter moves on libbpp_amd.so: the library's setters (the eigensystem of a locus whose
   frequencies / exchangeabilities changed is refreshed on the device before the next evaluation) */
int a00_backend_hip_params(void * vctx, unsigned locus, int which, const double * values, unsigned n)
{
  a00_hip_ctx_t * c = (a00_hip_ctx_t *)vctx;
  (void)n;
  if (which == 1) bpa_set_frequencies(c->loci[locus], 0, values);
  else if (which == 2) bpa_set_subst_params(c->loci[locus], 0, values);
  else bpa_set_category_rates(c->loci[locus], values);
  return 1;
}
