/*
 * bpp_amd.h — C ABI of libbpp_amd.so, the MI355X-native (HIP, gfx950) per-locus
 * partial-likelihood engine.  It is the drop-in boundary for the likelihood hot
 * path of BPP (bpp v4.8.7): every entry point below replaces one function of the
 * reference's locus API (cited as reference file:line), with the same argument
 * meaning and error behaviour, but operating on an opaque device-resident locus
 * instead of the host struct `locus_t` and on flat index arrays instead of
 * `gnode_t*` (INTEGRATION.md shows the 60-line C shim that maps one onto the
 * other inside BPP).
 *
 * Conventions
 *  - plain C, no HIP/torch types; `void * stream` is a hipStream_t or NULL.
 *  - functions returning int: 1 = success, 0 = failure (BPP_SUCCESS/BPP_FAILURE,
 *    bpp.h:175-176); bpa_last_error() describes the last failure.  The reference
 *    calls fatal() (util.c:30) where this library returns 0/NaN and sets the error.
 *  - CLVs cross the boundary in the reference's layout  [pattern][rate][state]
 *    (core_partials.c:741-743), P-matrices as [rate][row][col] (locus.c:765-771),
 *    scalers as unsigned[pattern]; on the device they are laid out differently
 *    (DESIGN.md §3).
 *  - all arithmetic is IEEE fp64 with the summation order of the reference's
 *    AVX/AVX2 back-end, so CLVs are bit-identical to it.
 *  - there is NO CPU fallback: without a GPU every compute call fails.
 */
#ifndef BPP_AMD_H
#define BPP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* data types / models: numeric values of the reference (bpp.h:208-247) */
#define BPA_DATA_DNA          0
#define BPA_DATA_AA           1
#define BPA_DNA_MODEL_JC69    0
#define BPA_DNA_MODEL_K80     1
#define BPA_DNA_MODEL_F81     2
#define BPA_DNA_MODEL_HKY     3
#define BPA_DNA_MODEL_T92     4
#define BPA_DNA_MODEL_TN93    5
#define BPA_DNA_MODEL_F84     6
#define BPA_DNA_MODEL_GTR     7
#define BPA_AA_MODEL_MIN      9
#define BPA_AA_MODEL_LG      10
#define BPA_AA_MODEL_MAX     27

/* the architecture bit a BPP build would add next to PLL_ATTRIB_ARCH_AVX2
   (bpp.h:364-370); other attribute bits are accepted and ignored            */
#define BPA_ATTRIB_ARCH_HIP   (1u << 6)
#define BPA_SCALE_BUFFER_NONE (-1)            /* PLL_SCALE_BUFFER_NONE, bpp.h:380 */

typedef struct bpa_engine bpa_engine_t;       /* one per process / GPU           */
typedef struct bpa_locus  bpa_locus_t;        /* device twin of locus_t (bpp.h:863) */
typedef struct bpa_plan   bpa_plan_t;         /* a resident batched proposal step */

/* One node update = one call of pll_core_update_partial_ii (core_partials.c:585)
   as issued by locus_update_partials (locus.c:2549-2569): the buffer indices are
   the gnode_t fields clv_index / scaler_index / pmatrix_index (bpp.h:716-718) of
   the node and of its two children.  scaler = BPA_SCALE_BUFFER_NONE for "NULL". */
typedef struct bpa_op
{
  uint32_t parent_clv;
  int32_t  parent_scaler;
  uint32_t left_clv;
  uint32_t left_pmatrix;
  int32_t  left_scaler;
  uint32_t right_clv;
  uint32_t right_pmatrix;
  int32_t  right_scaler;
} bpa_op_t;

/* ----------------------------------------------------------------- engine -- */
const char * bpa_version(void);
const char * bpa_last_error(void);
/* 1 when the library was built with -DBPA_EXPERIMENTAL: csrc/experimental/ (superseded kernel generations) compiled in, their A/B
   switches read from the environment; 0: the default build, where each of those switches is the constant "not set" */
int bpa_experimental_build(void);
int          bpa_device_count(void);           /* 0 when no GPU is visible        */

bpa_engine_t * bpa_engine_create(int device, void * stream);
void           bpa_engine_destroy(bpa_engine_t *);
int            bpa_engine_synchronize(bpa_engine_t *);
/* explicit parameters for the globals the reference path reads:
   opt_usedata (locus.c:2424,2540,2581) and opt_bfbeta (locus.c:2630)          */
void           bpa_engine_set_options(bpa_engine_t *, int usedata, double bfbeta);

/* ------------------------------------------------------------------ locus -- */
/* locus_create (locus.c:622, bpp.h:2032): same arguments, plus the engine.
   clv indices 0..tips-1 are tips, tips..tips+clv_buffers-1 inner buffers.      */
bpa_locus_t * bpa_locus_create(bpa_engine_t *, unsigned dtype, unsigned model,
                               unsigned tips, unsigned clv_buffers, unsigned states,
                               unsigned sites, unsigned rate_matrices,
                               unsigned prob_matrices, unsigned rate_cats,
                               unsigned scale_buffers, unsigned attributes);
void bpa_locus_destroy(bpa_locus_t *);                              /* locus.c:872  */
/* pll_set_tip_states (locus.c:561): map = pll_map_nt / pll_map_aa style table
   (256 entries, 0 = illegal character -> returns 0 where the reference aborts)  */
int  bpa_set_tip_states(bpa_locus_t *, unsigned tip_index, const unsigned * map,
                        const char * sequence);
void bpa_set_pattern_weights(bpa_locus_t *, const unsigned * weights); /* locus.c:250 */
void bpa_set_frequencies(bpa_locus_t *, unsigned index, const double * f);  /* locus.c:889 */
void bpa_set_subst_params(bpa_locus_t *, unsigned index, const double * p); /* locus.c:877 */
void bpa_set_category_rates(bpa_locus_t *, const double * rates);  /* writes locus->rates, prop_gamma.c:93 */
void bpa_set_category_weights(bpa_locus_t *, const double * w);    /* locus->rate_weights, locus.c:847 */
void bpa_set_param_indices(bpa_locus_t *, const unsigned * idx);   /* locus->param_indices, locus.c:731 */
/* diploid loci: the fields method.c:4173-4196 installs on locus_t               */
int  bpa_set_diploid(bpa_locus_t *, int unphased_length,
                     const unsigned long * resolution_count,
                     const unsigned long * mapping, unsigned long mapping_len,
                     const unsigned * unphased_weights);
/* state tables equal to pll_map_nt / pll_map_aa (maps.c:26,126)                 */
const unsigned * bpa_map_nt(void);
const unsigned * bpa_map_aa(void);

/* ----------------------------------------------------- update API (1 locus) -- */
/* locus_update_matrices (locus.c:2417): branch i gets P(t_i) into P-matrix
   buffer pmatrix_indices[i]; t_i is the branch length the reference derives as
   (parent.time - time) * rate_mui (locus.c:2350) — the caller's shim computes it
   and stores node->length.  JC69 closed form (locus.c:2342-2414); GTR / amino-acid
   through the eigendecomposition (core_pmatrix.c:674-783), refreshed on the
   device when frequencies / exchangeabilities changed (locus.c:2462-2476).       */
int  bpa_locus_update_matrices(bpa_locus_t *, const unsigned * pmatrix_indices,
                               const double * branch_lengths, unsigned count);
/* locus_update_partials (locus.c:2530): ops in children-first order             */
int  bpa_locus_update_partials(bpa_locus_t *, const bpa_op_t * ops, unsigned count);
/* locus_root_loglikelihood (locus.c:2573): root given by its clv/scaler index;
   freqs_indices may be NULL (= param_indices); persite_lnl may be NULL.
   Returns NaN on failure.                                                       */
double bpa_locus_root_loglikelihood(bpa_locus_t *, unsigned root_clv, int root_scaler,
                                    const unsigned * freqs_indices, double * persite_lnl);

/* The lazy contract of the three calls above: locus_update_matrices / locus_update_partials only queue their work on
   the locus; it runs — as ONE launch together with the root term — when bpa_locus_root_loglikelihood asks for the
   value, or before anything else reads or writes the locus's buffers (a plan launch, a buffer access, a sampler).
   A proposal of the reference (matrices -> partials -> lnL on one locus: gtree.c:5447-5467, 7484-7566,
   stree.c:4727-4749) is then one launch and one synchronisation instead of three.                                */

/* locus_update_all_matrices (locus.c:1922) / locus_update_all_partials (locus.c:2523): the gene tree crosses the
   boundary as a flat view of the gnode_t fields the reference's recursions read (bpp.h:692-732), one entry per node
   (tips first; left/right = -1 for tips, parent = -1 for the root).  Branch lengths are the strict clock's
   (parent.time - time) * rate_mui (locus.c:1826) for every node but the root, visited in the reference's pre-order
   from root->left then root->right; lengths_out (nodes entries, or NULL) receives them as the reference stores
   them in node->length.  Partials: every inner node in the post-order of locus.c:2482-2521.  Relaxed clocks derive
   the length from species-tree rates (locus.c:1105-1193): the caller computes it and uses bpa_locus_update_matrices. */
typedef struct bpa_gtree_view
{
  unsigned         nodes;                /* 2*tips - 1 */
  int              root;
  const int *      left;
  const int *      right;
  const int *      parent;
  const double *   time;
  const unsigned * clv_index;
  const int *      scaler_index;
  const unsigned * pmatrix_index;
  double           rate_mui;             /* gtree_t.rate_mui (bpp.h:757) */
} bpa_gtree_view_t;
int  bpa_locus_update_all_matrices(bpa_locus_t *, const bpa_gtree_view_t *, double * lengths_out);
int  bpa_locus_update_all_partials(bpa_locus_t *, const bpa_gtree_view_t *);

/* pll_core_update_pmatrix (core_pmatrix.c:785, bpp.h:2349): library form over
   host arrays (expm1(lambda*rate*t), identity when t == 0); evaluated on the
   device of `engine`.                                                          */
int  bpa_core_update_pmatrix(bpa_engine_t * engine, double ** pmatrix, unsigned states,
                             unsigned rate_cats, const double * rates,
                             const double * branch_lengths,
                             const unsigned * matrix_indices,
                             const unsigned * param_indices,
                             double * const * eigenvals, double * const * eigenvecs,
                             double * const * inv_eigenvecs, unsigned count,
                             unsigned attrib);
/* pll_update_eigen (core_pmatrix.c:239) on the device; arrays are S*S / S        */
int  bpa_update_eigen(bpa_engine_t * engine, double * eigenvecs, double * inv_eigenvecs,
                      double * eigenvals, const double * freqs,
                      const double * subst_params, unsigned states);
/* pll_compute_gamma_cats, mean mode (gamma.c:221) — host scalar code, as in the
   reference                                                                     */
int  bpa_compute_gamma_cats(double alpha, double beta, unsigned categories, double * rates);
/* compress_site_patterns (compress.c:218): in-place, returns number of patterns
   (0 on failure); weights must hold *length entries.  jc69 != 0 = COMPRESS_JC69  */
int  bpa_compress_site_patterns(char ** sequences, const unsigned * map, int count,
                                int * length, int jc69, unsigned * weights);

/* buffer access in the reference's layouts (dump.c:947-1046 / debug printers)    */
int  bpa_locus_get_clv(bpa_locus_t *, unsigned clv_index, double * out);
int  bpa_locus_set_clv(bpa_locus_t *, unsigned clv_index, const double * in);
int  bpa_locus_get_pmatrix(bpa_locus_t *, unsigned pmatrix_index, double * out);
int  bpa_locus_set_pmatrix(bpa_locus_t *, unsigned pmatrix_index, const double * in);
int  bpa_locus_get_scaler(bpa_locus_t *, unsigned scaler_index, unsigned * out);
/* write scale buffer `scaler_index` ([sites] counters, locus->scale_buffer[i] of locus.c:800): test access, as bpa_locus_set_clv */
int  bpa_locus_set_scaler(bpa_locus_t *, unsigned scaler_index, const unsigned * in);
int  bpa_locus_get_eigen(bpa_locus_t *, unsigned index, double * eigenvecs,
                         double * inv_eigenvecs, double * eigenvals);

/* ------------------------------------------------- batched update (N loci) --- */
/* One proposal step for many loci in a single fused launch sequence: for locus
   loci[i], branches mat_*[mat_off[i]..mat_off[i+1]) are updated, then node
   updates ops[op_off[i]..op_off[i+1]) run, then the root log-likelihood is taken
   at (root_clv[i], root_scaler[i]).  This is the body every proposal of the
   reference executes per locus (gtree.c:5447-5467, 7484-7566; stree.c:4727-4749;
   prop_mixing.c:117-131) hoisted over the loci loop that threads.c:87-200 shards. */
typedef struct bpa_batch
{
  unsigned             nloci;
  bpa_locus_t * const * loci;
  const unsigned *     mat_off;          /* nloci+1 */
  const unsigned *     mat_pmatrix;      /* P-matrix buffer index per branch      */
  const double *       mat_length;       /* branch length per branch              */
  const unsigned *     op_off;           /* nloci+1 */
  const bpa_op_t *     ops;
  const unsigned *     root_clv;         /* nloci   */
  const int *          root_scaler;      /* nloci   */
} bpa_batch_t;

/* upload the descriptors once; the plan stays resident in HBM                     */
bpa_plan_t * bpa_plan_create(bpa_engine_t *, const bpa_batch_t *);
void         bpa_plan_destroy(bpa_plan_t *);
/* replace the branch lengths of a resident plan (same shape)                      */
int          bpa_plan_set_lengths(bpa_plan_t *, const double * mat_length);
/* enqueue the step on the engine's stream; results stay on the device             */
int          bpa_plan_launch(bpa_plan_t *);
/* enqueue several steps back to back (one host call).  Consecutive per-locus steps — JC69 plans on the engine's
   packing without a plan total (bpa_plan_enable_sum / _partial_sums mark the all-loci steps) — go out as ONE chain
   launch in which a workgroup walks its loci through all of them (threads.c:87-200: a worker walks its loci's
   proposals without a barrier); results land in each plan's own arrays as if launched one by one.               */
int          bpa_plans_launch(bpa_plan_t * const * plans, unsigned count);
/* copy the nloci log-likelihoods of the last launch to the host (synchronises)    */
int          bpa_plan_get_lnl(bpa_plan_t *, double * lnl);
/* device address of the nloci log-likelihoods / of their sum (double)             */
void *       bpa_plan_lnl_device(bpa_plan_t *);
/* Sum of the plan's log-likelihoods over its loci, produced on the device by every
   launch — the per-proposal reduction of threads.c:544-559,583-591 that all-loci
   proposals (tau, mixing) need, and the only quantity exchanged between GPUs.
   device_out: where to write it (e.g. a buffer RCCL all-reduces), or NULL for an
   internal one read back with bpa_plan_get_sum.                                    */
int          bpa_plan_enable_sum(bpa_plan_t *, void * device_out);
/* The same sum delivered as partial sums by the step kernel itself, without a launch of its own (4.3 us each, 14 % of
   a config-2 iteration): *count doubles — one per workgroup — whose total is the plan's sum; the consumer (a decision
   kernel, or the host after ONE all-reduce of the *count doubles over the GPUs) adds them up.  device_out: where to
   write them, with its capacity in *count on entry, or NULL for an internal buffer; on return *count is the number
   in use (1 = the plain total, for plans that do not run on the engine's packing or when the capacity is too small). */
int          bpa_plan_enable_partial_sums(bpa_plan_t *, void * device_out, unsigned * count);
/* the total of the last launch (adds the partial sums on the host when there are several; synchronises) */
int          bpa_plan_get_sum(bpa_plan_t *, double * sum);
/* Batched substitution-parameter proposal for the plan's loci (the per-locus proposals of locus.c:2782-3419
   — base frequencies, exchangeabilities — and prop_gamma.c:52-224 — alpha, whose category rates the caller
   computes with bpa_compute_gamma_cats as BPP does): which = 1 frequencies of rate matrix 0 (states values per
   locus), 2 substitution parameters (states(states-1)/2), 4 category rates (rate_cats); values packed
   [locus of the plan][value].  ONE transfer (or none: the _device form takes device memory), one kernel that
   installs them and refreshes the touched eigensystems on the device (K6, pll_update_eigen).  The next launch
   of a plan that updates all matrices and partials then evaluates the proposal.                            */
/* a caller-side tape resident in HBM: copy `bytes` from the host into device memory owned by the engine (freed
   with it) and return the device address, e.g. the value arrays bpa_plan_set_params_device reads            */
void *       bpa_engine_stage(bpa_engine_t *, const void * host, size_t bytes);
int          bpa_plan_set_params(bpa_plan_t *, int which, const double * values);
int          bpa_plan_set_params_device(bpa_plan_t *, int which, const double * device_values);
/* convenience: create + launch + get + destroy                                    */
int          bpa_batch_evaluate(bpa_engine_t *, const bpa_batch_t *, double * lnl);
/* The same in three parts, for a caller with worker threads (the C host driver's: csrc/host/a00_driver.c) — the step's
   record image is the serial part of a host-driven step, and the records of a locus touch nothing but that locus's slot:
     bpa_batch_begin   checks and sizes; returns 1 = go on with fill + end, 2 = this batch does not take the one-image path
                       (loci outside the engine's packing, 20 states, scalers with several categories ...): call
                       bpa_batch_evaluate instead, 0 = error;
     bpa_batch_fill    writes the records of the batch's loci [t0, t1); disjoint ranges may be filled from several threads
                       at once (no lock is taken); an invalid index makes bpa_batch_end fail;
     bpa_batch_end     uploads the image (one copy), launches, returns the per-locus lnL like bpa_batch_evaluate (1), or 2 when
                       a fill found a locus this path does not take (call bpa_batch_evaluate instead), 0 = error.
   One batch at a time per engine; nothing else may touch the engine between begin and end.                              */
int          bpa_batch_begin(bpa_engine_t *, const bpa_batch_t *);
int          bpa_batch_fill(bpa_engine_t *, const bpa_batch_t *, unsigned t0, unsigned t1);
int          bpa_batch_end(bpa_engine_t *, const bpa_batch_t *, double * lnl);
/* bpa_batch_end without the wait: upload, launch and the copy of the results back are queued on the engine's stream and the
   call returns (1; 2 and 0 as bpa_batch_end) — the caller proposes for OTHER loci (another engine's: the host driver's
   cohorts, a00_set_cohorts) meanwhile; bpa_batch_wait then waits for the stream and hands out the per-locus lnL of that
   batch.  Nothing else may touch this engine between the two calls.                                                   */
int          bpa_batch_end_async(bpa_engine_t *, const bpa_batch_t *);
int          bpa_batch_wait(bpa_engine_t *, double * lnl);

/* ------------------------------------- device-resident proposal control (next) --- */
/* The multispecies-coalescent sampler of include/bpp_amd_host.h with everything resident on the
   device: the per-locus proposals of an iteration (gene-node ages, gtree.c:4585; prune/regraft,
   gtree.c:6531), the all-loci steps (tau with its rubber band, stree.c:5512; mixing,
   prop_mixing.c:52), the MSC density (gtree_logprob, gtree.c:3957), the gene trees with their
   populations and buffer-index bookkeeping, the random streams and the accept/reject decisions:
   same arithmetic, same streams, same trajectory as the host driver, without a host round trip per
   proposal.  Four implementations behind the one interface (bpa_sampler_kind tells which one runs).  Where every
   locus is JC69 with one rate category, <= 8 tips and <= 64 patterns: on one GPU the PERSISTENT ITERATION KERNEL
   (csrc/sweep2.hpp: the state of all loci stays in LDS for a whole call of bpa_sampler_iterate — many iterations —,
   a group of 8 or 16 lanes per locus runs the proposals, the all-loci decisions come from device-scope fixed-point
   accumulators inside the launch); with an all-reduce callback installed (several ranks), more loci than stay
   resident at once or BPA_SMP_V1=1, the sweep kernel with one launch per step (csrc/sampler.hpp); otherwise — several rate categories, GTR, up to 16 tips, < 256
   patterns x categories — a generic path that proposes on the device and evaluates with the engine's batched step
   kernels (csrc/gsampler.hpp; 20-state loci too: their steps are written as the records of the tiled kernels); a
   4-state locus of more than 16 tips (<= 64), with scale buffers or as an unphased diploid makes it the big-tree path
   (csrc/bigsampler.hpp: the host driver's proposal code on trees in HBM / LDS, the engine's general kernels with
   scaling and phase averaging; one rank).  <= 15 populations.  Trees use the node numbering of a00_tree_t (tips first;
   arrays of 2*tips-1 entries); the species tree that of a00_set_species_tree.                                   */
typedef struct bpa_sampler bpa_sampler_t;
bpa_sampler_t * bpa_sampler_create(bpa_engine_t *, bpa_locus_t * const * loci, unsigned nloci,
                                   unsigned long seed);
void bpa_sampler_destroy(bpa_sampler_t *);
int  bpa_sampler_set_tree(bpa_sampler_t *, unsigned i, const int * left, const int * right,
                          const double * times, int root);
/* the species tree (a00_set_species_tree's arguments: tips first, children before parents; arrays of
   2*species-1 entries); the taus then live on the device.  Required before initialize.            */
int  bpa_sampler_set_species_tree(bpa_sampler_t *, int species, const int * parent, const double * tau,
                                  const double * theta);
int  bpa_sampler_set_tip_species(bpa_sampler_t *, unsigned i, const int * species);   /* default: tip k = species k */
void bpa_sampler_set_finetune(bpa_sampler_t *, double gage, double gspr, double tau, double mix);
/* The burn-in's step-length rule (reset_finetune_onestep, method.c:1122-1136: step *= tan(pi/2 pjump) / tan(pi/2 0.3), / 100
   below 0.001, x 100 above 0.999, at most 99).  bpa_sampler_adapt_finetune applies it to the five step lengths — gene-node
   age, prune/regraft, tau, mixing, theta window, in that order — from the acceptance proportions of each move type since
   the last call (counted by the persistent iteration kernel; pjump[m] < 0: never proposed, step length kept) and clears the
   counters, as reset_finetune + pjump_reset do (method.c:1508-1516, 5364-5377); pjump / finetune (5 each) may be null.
   bpa_sampler_burnin runs `iterations` iterations the program's way: the rule where the program's loop applies it
   (method.c:5364-5417: at the top of iteration i = -burnin .. -1 when i % (burnin/4) == 0 and at least 100 iterations have run
   since the last reset, and at i = 0 — 400: after 100 / 200 / 300 / 400 iterations, 300: after 150 / 300, 402: after 102 / 202 /
   302 / 402; below 200 iterations the program resets nothing and neither does this); bpa_burnin_schedule writes those points
   (at most cap of them) and returns their number.  finetune (5, may be null) receives the step lengths the burn-in ends with.
   A sampler the rule cannot run on (loci of several kinds, the big-tree sampler, a generic sampler without the program's moves)
   fails BEFORE any iteration has run.
   Where: the persistent iteration kernel (device counters by move type) and the generic sampler with the program's moves
   (the trees carry the age / prune-regraft counts, the host its own decisions').  Several ranks: the per-locus moves' counts
   are pooled over the ranks first (the callback, or the mailboxes' one-shot exchange), so every rank ends at the same step
   lengths, the whole data set's — every rank calls it at the same point of its run.                                      */
double bpa_finetune_onestep(double pjump, double finetune);
int  bpa_sampler_adapt_finetune(bpa_sampler_t *, double * pjump, double * finetune);
int  bpa_sampler_burnin(bpa_sampler_t *, unsigned iterations, double * finetune);
unsigned bpa_burnin_schedule(unsigned burnin, unsigned * after, unsigned cap);
/* which generator and window the moves draw from (a00_set_proposal_kernel of bpp_amd_host.h; before initialize):
   BPA_KERNEL_UNIFORM (default) our 64-bit streams, window = finetune x (u - 1/2), the acceptance number always drawn;
   BPA_KERNEL_BPP     the reference's own — legacy_rndu (random.c:104-122) and the Bactrian-Laplace variate of
                      legacy_rnd_symmetrical (random.c:192-238) for the ages, taus and thetas, acceptance "lnacc >= -1e-10 or
                      rndu < exp(lnacc)" with the number drawn only when needed (gtree.c:5476, stree.c:6286): the finetunes
                      then mean what they mean in a BPP control file.  Same trajectory as the host driver with
                      A00_KERNEL_BPP.  The persistent iteration kernel, and (round 5) the generic sampler — there together
                      with bpa_sampler_set_program_moves and a theta prior; not the big-tree sampler (bpa_sampler_kind).  */
#define BPA_KERNEL_UNIFORM 0
#define BPA_KERNEL_BPP     1
int  bpa_sampler_set_proposal_kernel(bpa_sampler_t *, int kind);
/* THETA, TAU and MIX as the program runs them (a00_set_program_moves of bpp_amd_host.h; needs BPA_KERNEL_BPP; default off):
     THETA  sliding window with probability slide_prob (stree_propose_theta, stree.c:3957: opt_theta_slide_prob = 0.1), the
            metropolized Gibbs draw of propose_theta_gibbs (stree.c:3645: an inverse-gamma fitted to the conditional given the
            gene trees, get_gamma_conditional_approx stree.c:3384) otherwise;
     TAU    the rubber band re-draws the thetas of the population and its two children (opt_rb_theta_update, stree.c:5840);
     MIX    re-draws every theta with the scaled trees (opt_mix_theta_update, prop_mixing.c:272).
   All decided from two sums over the loci per theta — coalescences and T2h, carried through the iteration.  The persistent
   kernel decides inside the launch (its control wave): one exchange per step, as without.  The generic sampler (any model,
   <= 16 tips) takes the decision ON THE DEVICE since round 6 — one wave (gsm::gdec_kernel) runs the persistent kernel's
   control-wave functions on the loci's sums, installs the decision and makes the coming step's species-tree proposal: no host
   synchronisation inside an iteration (several ranks: the sums pass through the all-reduce callback on the stream first, so
   with a stream-ordered collective none either).  BPA_GS_HOSTDEC=1 keeps round 5's form:
   the sums come to the HOST — 24 to 72 bytes and one synchronisation per all-loci step — which takes the
   decision with the statements of a00_driver.c (theta_step_gibbs / tau_step / mix_step) and sends it back as a one-lane
   launch.  With an all-reduce callback (several ranks) every rank's host decides from the sums over ALL ranks' loci: they go
   through the callback first, BPA_SAMPLER_SUMS doubles at a time — the integer sums (counts, 2^-40 fixed-point T2h) as two
   doubles each, exact; the likelihood + Jacobian sum as a double.  Same trajectory as the host driver with
   a00_set_program_moves.                                                                                                */
int  bpa_sampler_set_program_moves(bpa_sampler_t *, int on, double slide_prob);
int  bpa_sampler_gibbs_counters(bpa_sampler_t *, unsigned long * proposals, unsigned long * accepted);
void bpa_sampler_set_tau_prior(bpa_sampler_t *, double alpha, double beta);           /* a00_set_tau_prior */
void bpa_sampler_set_theta_prior(bpa_sampler_t *, double alpha, double beta, double finetune); /* a00_set_theta_prior */
int  bpa_sampler_get_thetas(bpa_sampler_t *, double * theta); /* 2*species-1 entries; returns their number */
int  bpa_sampler_get_taus(bpa_sampler_t *, double * tau);     /* 2*species-1 entries; returns their number */
/* Several GPUs (threads.c:544-591): every rank samples its own loci with the same seed, so
   the global stream gives all ranks the same window and acceptance numbers; the only exchange is the sum of an
   all-loci step's per-locus terms.  After summing its loci into device doubles the sampler calls fn(ctx, their
   address, count, the engine's hipStream_t); fn must enqueue a sum all-reduce over the ranks on that stream
   (RCCL: ncclAllReduce(p, p, count, ncclDouble, ncclSum, comm, stream)) and return non-zero.  count is 1 for a TAU or
   MIX step and the number of populations for the THETA step (one sum per theta: one collective for all of them).
   device_sums: caller-owned device memory for BPA_SAMPLER_SUMS doubles to use for the sums (e.g. memory a
   framework's collective can address) or NULL.  first_locus: global index of this rank's first locus — the
   per-locus random streams are keyed by global index, so a sharded run walks the trajectory of the single-GPU run.
   A sampler over loci of several kinds (BPA_SAMPLER_COMPOSITE) calls fn once per step on the total of its parts, i.e. as
   a plain sampler does: ranks with mixed shares and ranks with shares of one kind can be paired.                    */
#define BPA_SAMPLER_SUMS 16
typedef int (*bpa_allreduce_fn)(void * ctx, double * device_sums, unsigned count, void * stream);
int  bpa_sampler_set_allreduce(bpa_sampler_t *, bpa_allreduce_fn fn, void * ctx, double * device_sums,
                               unsigned first_locus);
/* The same with the exchange INSIDE the persistent iteration kernel (no callback, no launch boundary): `p2p` is a
   connected bpa_p2p_t of this engine (below: mailboxes in every rank's memory, mapped over xGMI).  After its own
   workgroups' sums are complete a rank stores them — 2^-40 fixed point, so the total is the same on every rank whatever
   the order — into its slot of every rank's mailbox, raises the slot's sequence flag, and every workgroup adds up the N
   slots of its own mailbox once their flags show the exchange.  Waits are bounded by bpa_p2p_set_timeout; a time-out is
   reported by the next call that reads the sampler's state.  Needs the persistent kernel (bpa_sampler_kind); every
   rank must call bpa_sampler_iterate with the same counts.  NULL takes it out again.                                  */
struct bpa_p2p;
int  bpa_sampler_set_p2p(bpa_sampler_t *, struct bpa_p2p * p2p, unsigned first_locus);
int  bpa_sampler_initialize(bpa_sampler_t *);                 /* all matrices, partials, lnL */
/* asynchronous on the engine stream.  The persistent iteration kernel's workgroups wait for each other's sums inside the
   launch, so ALL of them must be resident at once: the library checks the workgroup count against the device's compute
   units, not against other tenants — on a GPU shared with another process (or partitioned: CPX) some may never start.
   Every wait is bounded (0.5 s); a launch that times out leaves all loci as it found them, and the next call that reads
   the sampler's state runs its iterations again — by the same kernel (up to three times), then, with the library's own
   proposal kernel, by the one-launch-per-step path (BPA_SAMPLER_SWEEP), which the sampler then keeps to; BPP's proposal
   kernel has no such path: the call fails.  An all-loci step ACCEPTED with some locus's term summed through the coarse
   companion accumulator (|term| >= 256 log units: resolution 2^-10) is reported as an error by that same call. */
int  bpa_sampler_iterate(bpa_sampler_t *, unsigned iterations);
/* current state of locus i (any output may be NULL); asking for locus 0 refreshes the host copy */
int  bpa_sampler_get_tree(bpa_sampler_t *, unsigned i, int * left, int * right, int * parent,
                          double * times, int * clv, int * pmat, int * root, double * lnl);
int  bpa_sampler_get_tree_msc(bpa_sampler_t *, unsigned i, int * pop, double * logpr);
int  bpa_sampler_summary(bpa_sampler_t *, double * total_lnl, unsigned long * proposals,
                         unsigned long * accepted, unsigned long * launches);
/* The per-locus substitution-parameter moves of a GTR + Gamma analysis (propose_freqs locus.c:2782, propose_qrates
   locus.c:3168, propose_alpha prop_gamma.c:52; after the mixing step as in cmd_run, method.c:5699-5735): each base
   frequency but T and each exchangeability but A<->G in turn — a sliding window on its logarithm, the reference
   component taking up the difference — and the gamma shape alpha with a gamma(alpha_a, alpha_b) prior; every proposal
   a full recomputation of the locus with a per-locus decision, all on the device (the category rates of a proposed
   alpha by pll_compute_gamma_cats there).  Loci with several rate categories and an eigendecomposition only (the
   generic path); set every locus's starting values first.  Window width 0 = that move is off (default: all off).   */
int  bpa_sampler_set_subst_model(bpa_sampler_t *, unsigned i, const double * freqs /* 4 */, const double * qrates /* 6 */, double alpha);
int  bpa_sampler_get_subst_model(bpa_sampler_t *, unsigned i, double * freqs, double * qrates, double * alpha);
void bpa_sampler_set_subst_moves(bpa_sampler_t *, double ft_freqs, double ft_qrates, double ft_alpha, double alpha_a, double alpha_b);
/* measurement: HIP start/stop events on every stride-th launch of the sampler's likelihood-carrying kernels
   (stride 0 = off); bpa_sampler_timing returns the milliseconds and launch counts accumulated since, by kind: the
   sweep (the per-locus GAGE + GSPR proposals of an iteration, one launch) and the all-loci steps (TAU, MIX)       */
int  bpa_sampler_enable_timing(bpa_sampler_t *, unsigned stride);
int  bpa_sampler_timing(bpa_sampler_t *, double * sweep_ms, unsigned long * sweep_launches,
                        double * allloci_ms, unsigned long * allloci_launches);
/* algorithmic work of the sampler's likelihood launches so far (K1 bytes per node update actually
   run, K2 per evaluated proposal, K4 per fresh P-matrix where the step kernel computes them itself), their node and
   pattern-node updates, and the number of those launches: the sweeps of the LDS kernel (its per-locus proposals only),
   or, on the generic path, every launch of the engine's step kernel                                                */
int  bpa_sampler_work(bpa_sampler_t *, double * bytes, unsigned long * node_updates,
                      unsigned long * pattern_updates, unsigned long * sweeps);
/* which implementation bpa_sampler_iterate runs (known after bpa_sampler_initialize): BPA_SAMPLER_SWEEP (one launch per
   step, csrc/sampler.hpp), BPA_SAMPLER_GENERIC (csrc/gsampler.hpp, gsampler2.hpp) or BPA_SAMPLER_PERSISTENT (csrc/sweep2.hpp: its
   launches are what bpa_sampler_timing reports as `sweep`, its `sweeps` of bpa_sampler_work are iterations, and the
   work includes the all-loci steps' node updates) */
#define BPA_SAMPLER_SWEEP      0
#define BPA_SAMPLER_GENERIC    1
#define BPA_SAMPLER_PERSISTENT 2
#define BPA_SAMPLER_HYBRID     3       /* an all-reduce callback is installed (several ranks): the per-locus sweep of an iteration is
                                          ONE launch of the persistent kernel, the all-loci steps one launch each of csrc/sampler.hpp's */
#define BPA_SAMPLER_COMPOSITE  5       /* loci of several kinds (JC69 LDS-kernel loci, generic JC69, multi-category, 20-state): a part per kind,
                                          stepped together through the parts' all-reduce callbacks (csrc/composite.hpp); the library's own moves;
                                          several ranks through bpa_sampler_set_allreduce (one call per step on the parts' total), not the mailboxes */
#define BPA_SAMPLER_BIG        4       /* loci of more than 16 tips, with scalers or unphased diploids (csrc/bigsampler.hpp: trees in HBM,
                                          one lane per locus, the engine's general 4-state kernels; <= 64 tips) */
int  bpa_sampler_kind(bpa_sampler_t *);
/* 2 when the generic sampler runs its per-locus steps as two half-batches of the loci on two streams (enough workgroups of
   the packing to halve, loci in slot order: csrc/gsampler_host.hpp gs_fork) — twice the launches of bpa_sampler_work's
   `sweeps`, each over half the loci, overlapping in time —, else 1 */
int  bpa_sampler_streams(bpa_sampler_t *);

/* ------------------------------------------------------ work / measurement --- */
/* Algorithmic work of one launch of the plan, by the formulas of SURVEY.md §8(d):
   K1 node update: bytes = 3*Np*R*S*8 + 2*R*S^2*8 (+12*Np with scaling),
   flops = Np*R*(4S^2-S); K2: bytes = Np*R*S*8 + 4*Np; P-matrix: R*S^2*8 written.  */
int  bpa_plan_work(bpa_plan_t *, double * bytes_partials, double * flops_partials,
                   double * bytes_pmatrix, unsigned long * node_updates,
                   unsigned long * pattern_updates);
/* HIP-event timing on the engine's stream: while enabled, every bpa_plan_launch is
   bracketed by events; bpa_engine_timing returns the milliseconds accumulated in the
   P-matrix, partials(+site lnL) and per-locus reduction kernels and the number of
   launches since timing was enabled.                                              */
void bpa_engine_enable_timing(bpa_engine_t *, int on);
/* attach the events to every stride-th launch only (each event pair costs ~4 us of stream time) */
void bpa_engine_set_timing_stride(bpa_engine_t *, unsigned stride);
int  bpa_engine_timing(bpa_engine_t *, double * pmatrix_ms, double * partials_ms,
                       double * reduce_ms, unsigned long * launches);
/* what those timed launches covered: the proposal steps (a chain launch of bpa_plans_launch covers several) and the
   algorithmic bytes, by the formulas above, of the kernels partials_ms is the time of                              */
int  bpa_engine_timing_work(bpa_engine_t *, unsigned long * steps, double * bytes);
/* The same K1 + K2 work priced as the kernels hold the data (bench.py's `frac_codes`): a tip child is its state codes —
   1 B (DNA) / 4 B (amino acids) per pattern, once for all rate categories, where SURVEY 8(d) charges the one-hot CLV of
   core_partials.c:585's tip-as-CLV form, Np*R*S*8 —, and a child that is the previous node update's parent is forwarded
   in registers, not read again.  Parent stores, P-matrices, weights and scalers as above.                         */
int  bpa_plan_work_codes(bpa_plan_t *, double * bytes_codes);
int  bpa_engine_timing_work_codes(bpa_engine_t *, double * bytes_codes);

/* ---- several GPUs of one node: one-shot sum all-reduce of a few hundred doubles over xGMI peer mappings -----------
   The only exchange of a sharded run is the sum an all-loci proposal is decided on (SURVEY.md section 8e;
   threads.c:544-591), a few times per MCMC iteration and on the critical path.  Every rank stores its values into a
   mailbox of every rank (fine-grained device memory shared through hipIpc handles), raises a flag, waits for all flags
   in its own mailbox and adds the vectors up in rank order: one hop instead of a ring's 2(N-1), the same bits on every
   rank.  Set-up: every rank creates its end, the BPA_P2P_HANDLE_BYTES-byte handles are exchanged by the caller
   (any side channel: MPI, torch.distributed.all_gather_object) and passed, in rank order, to bpa_p2p_connect.
   bpa_p2p_allreduce enqueues on the engine's stream; waits inside it are bounded — bpa_p2p_status() (synchronises)
   returns 0 when every exchange so far completed, 1 after a time-out (the object is then unusable: use RCCL).       */
#define BPA_P2P_HANDLE_BYTES 64
typedef struct bpa_p2p bpa_p2p_t;
bpa_p2p_t *  bpa_p2p_create(bpa_engine_t *, int rank, int world, unsigned max_doubles, void * handle_out);
int          bpa_p2p_connect(bpa_p2p_t *, const void * handles);
int          bpa_p2p_allreduce(bpa_p2p_t *, double * device_values, unsigned n);
/* bpa_plans_launch followed by bpa_p2p_allreduce in one host call (a sharded step: launch, then exchange its sums) */
int          bpa_plans_launch_exchange(bpa_plan_t * const * plans, unsigned count, bpa_p2p_t *, double * device_values, unsigned n);
/* bound of a wait inside an exchange (default 3 000 ms): after a time-out every later exchange returns at once and
   bpa_p2p_status reports 1 — inside a timed region choose milliseconds, so a lost flag voids the run, not the clock */
void         bpa_p2p_set_timeout(bpa_p2p_t *, unsigned milliseconds);
int          bpa_p2p_status(bpa_p2p_t *);
void         bpa_p2p_destroy(bpa_p2p_t *);

#ifdef __cplusplus
}
#endif
#endif /* BPP_AMD_H */
