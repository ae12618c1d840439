/*
 * locus_hip.c — the reference-side binding of libbpp_amd.so: BPP's locus API (bpp.h:2032-2091) with the
 * reference's own names and signatures, forwarding to the C ABI of include/bpp_amd.h.
 *
 * It is compiled against the REAL bpp.h (cc -I$(REF)) and linked with the reference's own, unmodified
 * objects (oracle/Makefile, target _ref/bpp_hip): every caller of the locus API — method.c (init and the
 * MCMC loop), gtree.c / stree.c / prop_mixing.c / prop_gamma.c / prop_rj.c (the proposals), load.c
 * (resume) and the substitution-parameter proposals inside locus.c itself — then runs its likelihood on
 * the MI355X without a changed line.  How the definitions here take the place of locus.c's:
 *
 *   - the five update functions (locus_update_matrices :2417, locus_update_all_matrices :1922,
 *     locus_update_partials :2530, locus_update_all_partials :2523, locus_root_loglikelihood :2573) are
 *     made WEAK in the reference's position-independent locus.o (objcopy --weaken-symbol), so the strong
 *     definitions below win for every reference to them, including the calls inside locus.c
 *     (locus.c:2704-2720, 2847-2863, 3091-3107, 3279-3295);
 *   - locus_create :622, locus_destroy :872, pll_set_tip_states :561 and pll_set_pattern_weights :250 keep
 *     doing their host-side work — callers read and write locus_t fields directly —
 *     under the names bppref_* (objcopy --redefine-sym); the definitions below call them and mirror
 *     the result onto the device twin.
 *
 * The host struct stays the source of truth for everything callers edit in place (rates: prop_gamma.c:93,
 * frequencies: locus.c:2832, substitution parameters, eigen_decomp_valid = 0: locus.c:2846, the diploid
 * fields: method.c:4173-4196): before each matrix update the shim compares them with what the device
 * holds and pushes what changed.  CLVs, P-matrices and scalers live on the device only.
 *
 * This file is the integration a BPP maintainer would add (one more back-end); it is built and run by
 * tests/test_gpu_bpp_hip.py.  It is not part of libbpp_amd.so.
 */
#include "bpp.h"
#include "bpp_amd.h"

#include <pthread.h>

/* the reference's own host-side functions, renamed in its object file (see above) */
locus_t * bppref_locus_create(unsigned int dtype, unsigned int model, unsigned int tips,
                              unsigned int clv_buffers, unsigned int states, unsigned int sites,
                              unsigned int rate_matrices, unsigned int prob_matrices,
                              unsigned int rate_cats, unsigned int scale_buffers,
                              unsigned int attributes);
void bppref_locus_destroy(locus_t * locus);
int  bppref_pll_set_tip_states(locus_t * locus, unsigned int tip_index, const unsigned int * map,
                               const char * sequence);
void bppref_pll_set_pattern_weights(locus_t * locus, const unsigned int * pattern_weights);

/* ------------------------------------------------------------------ device twins -- */
typedef struct twin_s
{
  locus_t * host;
  bpa_locus_t * dev;
  /* what the device holds of the fields callers edit in place */
  double * rates, * rate_weights;          /* rate_cats each */
  double * freqs, * subst;                 /* rate_matrices x states / x states(states-1)/2 */
  unsigned int * param_indices;            /* rate_cats */
  int pushed;                              /* 0 until the first push */
  int diploid_done;
  unsigned char * tip_set;                 /* [tips] 1: the tip's states reached the twin through pll_set_tip_states */
  int weights_set;                         /* likewise pll_set_pattern_weights */
  /* scratch for one call */
  unsigned int * idx; double * len; bpa_op_t * ops; unsigned int cap;
} twin_t;

static bpa_engine_t * engine;
static pthread_once_t engine_once = PTHREAD_ONCE_INIT;
static pthread_rwlock_t table_lock = PTHREAD_RWLOCK_INITIALIZER;
static twin_t ** table;                    /* open addressing, keyed by the locus_t address */
static size_t table_size, table_used;

static void engine_init(void)
{
  const char * dev = getenv("BPP_HIP_DEVICE");
  engine = bpa_engine_create(dev ? atoi(dev) : 0, NULL);
  if (!engine)
    fatal("[bpp_hip] %s", bpa_last_error());
  if (!opt_quiet)
    fprintf(stdout, "Likelihood back-end: %s\n", bpa_version());
}

static size_t slot_of(const locus_t * l, size_t size)
{
  return (size_t)(((uintptr_t)l >> 4) * 0x9E3779B97F4A7C15ull) & (size - 1);
}

static void table_insert_nolock(twin_t * t)
{
  size_t i;
  if (2*(table_used + 1) > table_size)
  {
    size_t nsize = table_size ? 2*table_size : 1024, k;
    twin_t ** nt = (twin_t **)xcalloc(nsize, sizeof(twin_t *));
    for (k = 0; k < table_size; ++k)
      if (table[k])
      {
        i = slot_of(table[k]->host, nsize);
        while (nt[i]) i = (i + 1) & (nsize - 1);
        nt[i] = table[k];
      }
    free(table);
    table = nt; table_size = nsize;
  }
  i = slot_of(t->host, table_size);
  while (table[i]) i = (i + 1) & (table_size - 1);
  table[i] = t;
  ++table_used;
}

static twin_t * twin_of(const locus_t * l)
{
  twin_t * t = NULL;
  size_t i;
  pthread_rwlock_rdlock(&table_lock);
  if (table_size)
    for (i = slot_of(l, table_size); table[i]; i = (i + 1) & (table_size - 1))
      if (table[i]->host == l) { t = table[i]; break; }
  pthread_rwlock_unlock(&table_lock);
  if (!t)
    fatal("[bpp_hip] locus %p was not made by locus_create", (const void *)l);
  return t;
}

static void reserve(twin_t * t, unsigned int n)
{
  if (n <= t->cap) return;
  t->cap = n + 16;
  t->idx = (unsigned int *)xrealloc(t->idx, t->cap*sizeof(unsigned int));
  t->len = (double *)xrealloc(t->len, t->cap*sizeof(double));
  t->ops = (bpa_op_t *)xrealloc(t->ops, t->cap*sizeof(bpa_op_t));
}

/* ---------------------------------------------------------------- life cycle -- */
locus_t * locus_create(unsigned int dtype, unsigned int model, unsigned int tips,
                       unsigned int clv_buffers, unsigned int states, unsigned int sites,
                       unsigned int rate_matrices, unsigned int prob_matrices,
                       unsigned int rate_cats, unsigned int scale_buffers,
                       unsigned int attributes)
{
  locus_t * l = bppref_locus_create(dtype, model, tips, clv_buffers, states, sites, rate_matrices,
                                    prob_matrices, rate_cats, scale_buffers, attributes);
  twin_t * t = (twin_t *)xcalloc(1, sizeof(twin_t));
  const size_t nsub = (size_t)states*(states - 1)/2;

  pthread_once(&engine_once, engine_init);
  bpa_engine_set_options(engine, (int)opt_usedata, opt_bfbeta);
  t->host = l;
  t->dev = bpa_locus_create(engine, dtype, model, tips, clv_buffers, states, sites, rate_matrices,
                            prob_matrices, rate_cats, scale_buffers, attributes);
  if (!t->dev)
    fatal("[bpp_hip] %s", bpa_last_error());
  t->rates = (double *)xcalloc(rate_cats, sizeof(double));
  t->rate_weights = (double *)xcalloc(rate_cats, sizeof(double));
  t->param_indices = (unsigned int *)xcalloc(rate_cats, sizeof(unsigned int));
  t->freqs = (double *)xcalloc((size_t)rate_matrices*states, sizeof(double));
  t->subst = (double *)xcalloc((size_t)rate_matrices*nsub, sizeof(double));
  t->tip_set = (unsigned char *)xcalloc(tips ? tips : 1, 1);

  pthread_rwlock_wrlock(&table_lock);
  table_insert_nolock(t);
  pthread_rwlock_unlock(&table_lock);
  return l;
}

void locus_destroy(locus_t * locus)
{
  twin_t * t = twin_of(locus);
  size_t i, j;

  pthread_rwlock_wrlock(&table_lock);
  /* remove and re-insert the rest of the probe run (open addressing) */
  for (i = slot_of(locus, table_size); table[i] != t; i = (i + 1) & (table_size - 1)) ;
  table[i] = NULL; --table_used;
  for (j = (i + 1) & (table_size - 1); table[j]; j = (j + 1) & (table_size - 1))
  {
    twin_t * m = table[j];
    table[j] = NULL; --table_used;
    table_insert_nolock(m);
  }
  pthread_rwlock_unlock(&table_lock);

  bpa_locus_destroy(t->dev);
  free(t->rates); free(t->rate_weights); free(t->param_indices); free(t->freqs); free(t->subst);
  free(t->idx); free(t->len); free(t->ops); free(t->tip_set);
  free(t);
  bppref_locus_destroy(locus);
}

int pll_set_tip_states(locus_t * locus, unsigned int tip_index, const unsigned int * map,
                       const char * sequence)
{
  int rc = bppref_pll_set_tip_states(locus, tip_index, map, sequence);
  twin_t * t = twin_of(locus);
  if (rc == BPP_SUCCESS && !bpa_set_tip_states(t->dev, tip_index, map, sequence))
    fatal("[bpp_hip] %s", bpa_last_error());
  if (rc == BPP_SUCCESS && tip_index < locus->tips) t->tip_set[tip_index] = 1;
  return rc;
}

void pll_set_pattern_weights(locus_t * locus, const unsigned int * pattern_weights)
{
  twin_t * t = twin_of(locus);
  bppref_pll_set_pattern_weights(locus, pattern_weights);
  bpa_set_pattern_weights(t->dev, pattern_weights);
  t->weights_set = 1;
}

/* ------------------------------------------------------------------- host -> device -- */
/* A locus restored from a checkpoint (load.c:2016-2140, `bpp --resume`) gets its tip CLVs and its pattern weights written
   straight into the host struct — no pll_set_tip_states, no pll_set_pattern_weights.  Before the first update, whatever
   did not arrive through those calls is read off the host struct: a tip's state code per pattern is the set of states its
   0/1 CLV entries mark (set_tipclv, locus.c:525-559, backwards); the codes go through bpa_set_tip_states with a map made
   on the spot (one character per distinct code). */
static void sync_loaded_locus(locus_t * l, twin_t * t)
{
  unsigned int tip, n, s;
  const unsigned int S = l->states, R = l->rate_cats;
  char * seq = NULL;
  for (tip = 0; tip < l->tips; ++tip)
  {
    unsigned int map[256], used = 1;                 /* character 0 stays unmapped */
    const double * clv = l->clv[tip];
    if (t->tip_set[tip]) continue;
    if (!seq) seq = (char *)xmalloc((size_t)l->sites + 1);
    memset(map, 0, sizeof(map));
    for (n = 0; n < l->sites; ++n)
    {
      unsigned int code = 0, c;
      for (s = 0; s < S; ++s) if (clv[((size_t)n*R)*l->states_padded + s] != 0.0) code |= 1u << s;
      if (!code) fatal("[bpp_hip] tip %u of a restored locus has an empty state set at pattern %u", tip, n);
      for (c = 1; c < used && map[c] != code; ++c) ;
      if (c == used)
      {
        if (used == 256) fatal("[bpp_hip] more than 255 distinct state sets in one tip sequence");
        map[used++] = code;
      }
      seq[n] = (char)c;
    }
    seq[l->sites] = 0;
    if (!bpa_set_tip_states(t->dev, tip, map, seq))
      fatal("[bpp_hip] %s", bpa_last_error());
    t->tip_set[tip] = 1;
  }
  free(seq);
  if (!t->weights_set && !l->diploid)
  {
    bpa_set_pattern_weights(t->dev, l->pattern_weights);
    t->weights_set = 1;
  }
}

/* the fields the reference's callers edit in place, pushed when they differ from what the device holds */
static void push_state(locus_t * l, twin_t * t)
{
  unsigned int i;
  const unsigned int S = l->states, R = l->rate_cats;
  const size_t nsub = (size_t)S*(S - 1)/2;
  const int first = !t->pushed;

  if (first) sync_loaded_locus(l, t);
  if (l->diploid && !t->diploid_done)
  {
    /* method.c:4173-4196 installs these after locus_create without a call */
    unsigned long maplen = 0;
    long u;
    for (u = 0; u < l->unphased_length; ++u) maplen += l->diploid_resolution_count[u];
    if (!bpa_set_diploid(t->dev, l->unphased_length, l->diploid_resolution_count, l->diploid_mapping,
                         maplen, l->pattern_weights))
      fatal("[bpp_hip] %s", bpa_last_error());
    t->diploid_done = 1;
  }
  if (first || memcmp(t->rates, l->rates, R*sizeof(double)))
  {
    bpa_set_category_rates(t->dev, l->rates);
    memcpy(t->rates, l->rates, R*sizeof(double));
  }
  if (first || memcmp(t->rate_weights, l->rate_weights, R*sizeof(double)))
  {
    bpa_set_category_weights(t->dev, l->rate_weights);
    memcpy(t->rate_weights, l->rate_weights, R*sizeof(double));
  }
  if (first || memcmp(t->param_indices, l->param_indices, R*sizeof(unsigned int)))
  {
    bpa_set_param_indices(t->dev, l->param_indices);
    memcpy(t->param_indices, l->param_indices, R*sizeof(unsigned int));
  }
  for (i = 0; i < l->rate_matrices; ++i)
  {
    /* the eigensystem is the device's: a changed frequency / exchangeability refreshes it there
       (pll_update_eigen, locus.c:2462-2476), and the host flag is settled as the reference settles it */
    if (first || memcmp(t->freqs + (size_t)i*S, l->frequencies[i], S*sizeof(double)))
    {
      bpa_set_frequencies(t->dev, i, l->frequencies[i]);
      memcpy(t->freqs + (size_t)i*S, l->frequencies[i], S*sizeof(double));
    }
    if (first || memcmp(t->subst + i*nsub, l->subst_params[i], nsub*sizeof(double)))
    {
      bpa_set_subst_params(t->dev, i, l->subst_params[i]);
      memcpy(t->subst + i*nsub, l->subst_params[i], nsub*sizeof(double));
    }
    if (!(l->dtype == BPP_DATA_DNA && l->model != BPP_DNA_MODEL_GTR))
      l->eigen_decomp_valid[i] = 1;
  }
  t->pushed = 1;
}

/* branch length exactly as the reference derives it (locus.c:2347-2359, core_pmatrix.c:711-723) */
static double branch_length(gtree_t * gtree, gnode_t * x, stree_t * stree, long msa_index, int store)
{
  double t;
  if (opt_clock == BPP_CLOCK_GLOBAL)
    t = x->length = (x->parent->time - x->time)*gtree->rate_mui;
  else if (opt_clock == BPP_CLOCK_SIMPLE)
  {
    t = update_branchlength_relaxed_clock_simple(stree, x, gtree->rate_mui);
    if (store) x->length = t;
  }
  else
  {
    t = update_branchlength_relaxed_clock(stree, x, msa_index);
    if (store) x->length = t;
  }
  return t;
}

/* ------------------------------------------------------------------ the update API -- */
void locus_update_matrices(locus_t * locus, gtree_t * gtree, gnode_t ** traversal, stree_t * stree,
                           long msa_index, unsigned int count)
{
  unsigned int i;
  twin_t * t;

  if (!opt_usedata) return;
  t = twin_of(locus);
  push_state(locus, t);
  reserve(t, count);
  for (i = 0; i < count; ++i)
  {
    gnode_t * x = traversal[i];
    t->idx[i] = x->pmatrix_index;
    t->len[i] = branch_length(gtree, x, stree, msa_index, 1);
  }
  if (!bpa_locus_update_matrices(t->dev, t->idx, t->len, count))
    fatal("[bpp_hip] %s", bpa_last_error());
}

static void all_branches(gtree_t * gtree, gnode_t * x, stree_t * stree, long msa_index,
                         int store, twin_t * t, unsigned int * n)
{
  /* pre-order, as locus_update_all_matrices_*_recursive (locus.c:1213, 1811) */
  t->idx[*n] = x->pmatrix_index;
  t->len[*n] = branch_length(gtree, x, stree, msa_index, store);
  ++*n;
  if (!x->left) return;
  all_branches(gtree, x->left, stree, msa_index, store, t, n);
  all_branches(gtree, x->right, stree, msa_index, store, t, n);
}

void locus_update_all_matrices(locus_t * locus, gtree_t * gtree, stree_t * stree, long msa_index)
{
  unsigned int n = 0;
  twin_t * t = twin_of(locus);
  /* the eigen form under a relaxed clock does not store node->length (locus.c:1249-1254) */
  const int closed = locus->dtype == BPP_DATA_DNA && locus->model != BPP_DNA_MODEL_GTR;

  push_state(locus, t);
  reserve(t, gtree->tip_count + gtree->inner_count);
  all_branches(gtree, gtree->root->left, stree, msa_index, closed, t, &n);
  all_branches(gtree, gtree->root->right, stree, msa_index, closed, t, &n);
  if (!bpa_locus_update_matrices(t->dev, t->idx, t->len, n))
    fatal("[bpp_hip] %s", bpa_last_error());
}

static void fill_op(bpa_op_t * o, const gnode_t * x)
{
  /* the fields locus.c:2549-2569 reads */
  o->parent_clv = x->clv_index;          o->parent_scaler = x->scaler_index;
  o->left_clv = x->left->clv_index;      o->left_pmatrix = x->left->pmatrix_index;
  o->left_scaler = x->left->scaler_index;
  o->right_clv = x->right->clv_index;    o->right_pmatrix = x->right->pmatrix_index;
  o->right_scaler = x->right->scaler_index;
}

void locus_update_partials(locus_t * locus, gnode_t ** traversal, unsigned int count)
{
  unsigned int i;
  twin_t * t;

  if (!opt_usedata) return;
  t = twin_of(locus);
  reserve(t, count);
  for (i = 0; i < count; ++i)
    fill_op(t->ops + i, traversal[i]);
  if (!bpa_locus_update_partials(t->dev, t->ops, count))
    fatal("[bpp_hip] %s", bpa_last_error());
}

static void all_inner(gnode_t * x, twin_t * t, unsigned int * n)
{
  /* post-order, as locus_update_all_partials_recursive (locus.c:2482) */
  if (!x->left) return;
  all_inner(x->left, t, n);
  all_inner(x->right, t, n);
  fill_op(t->ops + (*n)++, x);
}

void locus_update_all_partials(locus_t * locus, gtree_t * gtree)
{
  unsigned int n = 0;
  twin_t * t;

  if (!opt_usedata) return;
  t = twin_of(locus);
  reserve(t, gtree->inner_count);
  all_inner(gtree->root, t, &n);
  if (!bpa_locus_update_partials(t->dev, t->ops, n))
    fatal("[bpp_hip] %s", bpa_last_error());
}

double locus_root_loglikelihood(locus_t * locus, gnode_t * root, const unsigned int * freqs_indices,
                                double * persite_lnl)
{
  double logl;
  twin_t * t;

  if (!opt_usedata) return 0;
  t = twin_of(locus);
  push_state(locus, t);            /* frequencies and category weights enter the root term */
  /* opt_bfbeta (locus.c:2630) is applied by the library (bpa_engine_set_options) */
  logl = bpa_locus_root_loglikelihood(t->dev, root->clv_index, root->scaler_index, freqs_indices,
                                      persite_lnl);
  if (logl != logl)
    fatal("[bpp_hip] %s", bpa_last_error());
  return logl;
}

/* ------------------------------------------------ pll_core_update_pmatrix (bpp.h:2349) --
 * The library form of the eigen P-matrix update (core_pmatrix.c:785-872): explicit arrays, no locus_t.  Its one
 * caller in the program is the sequence simulator (evolve_gtr_recursive, simulate.c:694-705: one 4x4 matrix per
 * branch and site rate).  The reference's definition is made WEAK in core_pmatrix.o (oracle/Makefile); this one
 * forwards to the device (bpa_core_update_pmatrix: expm1(lambda rate t) per eigenvalue, inv_eigenvecs x diag x
 * eigenvecs in the reference's order, identity for a zero length), so `bpp_hip --simulate` draws its GTR sequences
 * from P-matrices made on the MI355X.  One launch + one synchronisation per call: the parity path, not a fast one.  */
int pll_core_update_pmatrix(double ** pmatrix,
                            unsigned int states,
                            unsigned int rate_cats,
                            const double * rates,
                            const double * branch_lengths,
                            const unsigned int * matrix_indices,
                            const unsigned int * param_indices,
                            double * const * eigenvals,
                            double * const * eigenvecs,
                            double * const * inv_eigenvecs,
                            unsigned int count,
                            unsigned int attrib)
{
  pthread_once(&engine_once, engine_init);
  if (!bpa_core_update_pmatrix(engine, pmatrix, states, rate_cats, rates, branch_lengths, matrix_indices,
                               param_indices, eigenvals, eigenvecs, inv_eigenvecs, count, attrib))
    fatal("[bpp_hip] pll_core_update_pmatrix: %s", bpa_last_error());
  return BPP_SUCCESS;
}
