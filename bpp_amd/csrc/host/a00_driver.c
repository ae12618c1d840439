/*
 * a00_driver.c — host-side MCMC control in plain C over the likelihood boundary
 * (include/bpp_amd_host.h).  Mirrors the way BPP's proposals drive the locus API:
 *
 *   propose_ages   gtree.c:4585-5532  set node->time, SWAP_PMAT_INDEX on the 2-3 touched
 *                  branches, SWAP_CLV_INDEX/SWAP_SCALER_INDEX on the path to the root,
 *                  locus_update_matrices, locus_update_partials, locus_root_loglikelihood,
 *                  accept or swap everything back
 *   propose_spr    gtree.c:6531-7610  prune + regraft (the root node object stays the root,
 *                  gtree.c:6129-6175), 3-4 branches, one or two root paths
 *   proposal_mixing prop_mixing.c:52-221  every age times c, everything recomputed, ONE
 *                  decision from the summed log-likelihood difference
 *
 * with the per-locus loops hoisted into lock-step batches ("step j of every locus").
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "bpp_amd_host.h"

#define MAXN 512                          /* nodes per gene tree handled on the stack */

struct a00_driver
{
  unsigned nloci;
  a00_tree_t * trees;
  a00_eval_fn eval;
  void * ctx;
  a00_rng_t * rng;                      /* one stream per locus */
  a00_rng_t grng;                       /* global stream (mixing step) */
  /* step scratch */
  unsigned * s_locus; a00_tree_t ** s_tree; unsigned * s_br_off, * s_nd_off;
  int * s_br, * s_nd; size_t cap_br, cap_nd;
  double * s_lnl, * s_hast;
  /* per-locus undo snapshot (whole small tree) */
  int ** u_left, ** u_right, ** u_parent, ** u_clv, ** u_pmat, ** u_scaler; double ** u_time; int * u_root;
  unsigned long proposals, accepted, steps;
  double taus[8]; unsigned ntaus;
};


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

static void snapshot(a00_driver_t * d, unsigned i)
{
  a00_tree_t * t = d->trees + i; const size_t n = (size_t)t->n;
  memcpy(d->u_left[i], t->left, n*sizeof(int));   memcpy(d->u_right[i], t->right, n*sizeof(int));
  memcpy(d->u_parent[i], t->parent, n*sizeof(int)); memcpy(d->u_time[i], t->time, n*sizeof(double));
  memcpy(d->u_clv[i], t->clv, n*sizeof(int));     memcpy(d->u_pmat[i], t->pmat, n*sizeof(int));
  memcpy(d->u_scaler[i], t->scaler, n*sizeof(int)); d->u_root[i] = t->root;
}
static void restore(a00_driver_t * d, unsigned i)
{
  a00_tree_t * t = d->trees + i; const size_t n = (size_t)t->n;
  memcpy(t->left, d->u_left[i], n*sizeof(int));   memcpy(t->right, d->u_right[i], n*sizeof(int));
  memcpy(t->parent, d->u_parent[i], n*sizeof(int)); memcpy(t->time, d->u_time[i], n*sizeof(double));
  memcpy(t->clv, d->u_clv[i], n*sizeof(int));     memcpy(t->pmat, d->u_pmat[i], n*sizeof(int));
  memcpy(t->scaler, d->u_scaler[i], n*sizeof(int)); t->root = d->u_root[i];
}

a00_driver_t * a00_create(unsigned nloci, a00_eval_fn eval, void * ctx, unsigned long seed)
{
  a00_driver_t * d = (a00_driver_t *)calloc(1, sizeof(*d));
  unsigned q;
  d->nloci = nloci; d->eval = eval; d->ctx = ctx;
  d->rng = (a00_rng_t *)calloc(nloci, sizeof(a00_rng_t));
  for (q = 0; q < nloci; ++q) d->rng[q] = a00_rng_seed(seed, q);
  d->grng = a00_rng_seed(seed, A00_GLOBAL_STREAM);
  d->trees = (a00_tree_t *)calloc(nloci, sizeof(a00_tree_t));
  d->s_locus = (unsigned *)calloc(nloci, sizeof(unsigned));
  d->s_tree = (a00_tree_t **)calloc(nloci, sizeof(a00_tree_t *));
  d->s_br_off = (unsigned *)calloc(nloci + 1, sizeof(unsigned));
  d->s_nd_off = (unsigned *)calloc(nloci + 1, sizeof(unsigned));
  d->s_lnl = (double *)calloc(nloci, sizeof(double));
  d->s_hast = (double *)calloc(nloci, sizeof(double));
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
    free(t->left); free(t->right); free(t->parent); free(t->time); free(t->clv); free(t->pmat); free(t->scaler);
    free(d->u_left[i]); free(d->u_right[i]); free(d->u_parent[i]); free(d->u_clv[i]); free(d->u_pmat[i]);
    free(d->u_scaler[i]); free(d->u_time[i]);
  }
  free(d->rng); free(d->trees); free(d->s_locus); free(d->s_tree); free(d->s_br_off); free(d->s_nd_off); free(d->s_br);
  free(d->s_nd); free(d->s_lnl); free(d->s_hast); free(d->u_left); free(d->u_right); free(d->u_parent);
  free(d->u_clv); free(d->u_pmat); free(d->u_scaler); free(d->u_time); free(d->u_root); free(d);
}

int a00_set_tree(a00_driver_t * d, unsigned i, int tips, const int * left, const int * right,
                 const double * times, int root, int scaling)
{
  a00_tree_t * t = d->trees + i;
  const int n = 2*tips - 1; int k;
  if (i >= d->nloci || n > MAXN || tips < 2) return 0;
  t->tips = tips; t->n = n; t->root = root; t->rate_mui = 1.0; t->lnl = 0;
#define DUP(dst, src, T) do { dst = (T *)malloc((size_t)n*sizeof(T)); memcpy(dst, src, (size_t)n*sizeof(T)); } while (0)
  DUP(t->left, left, int); DUP(t->right, right, int); DUP(t->time, times, double);
#undef DUP
  t->parent = (int *)malloc((size_t)n*sizeof(int)); t->clv = (int *)malloc((size_t)n*sizeof(int));
  t->pmat = (int *)malloc((size_t)n*sizeof(int)); t->scaler = (int *)malloc((size_t)n*sizeof(int));
  for (k = 0; k < n; ++k) t->parent[k] = -1;
  for (k = 0; k < n; ++k)
  {
    if (left[k] >= 0) { t->parent[left[k]] = k; t->parent[right[k]] = k; }
    t->clv[k] = k; t->pmat[k] = k;                               /* gtree.c:2433-2439, 2398 */
    t->scaler[k] = (scaling && k >= tips) ? k - tips : BPA_SCALE_BUFFER_NONE;
  }
  d->u_left[i] = (int *)malloc((size_t)n*sizeof(int)); d->u_right[i] = (int *)malloc((size_t)n*sizeof(int));
  d->u_parent[i] = (int *)malloc((size_t)n*sizeof(int)); d->u_clv[i] = (int *)malloc((size_t)n*sizeof(int));
  d->u_pmat[i] = (int *)malloc((size_t)n*sizeof(int)); d->u_scaler[i] = (int *)malloc((size_t)n*sizeof(int));
  d->u_time[i] = (double *)malloc((size_t)n*sizeof(double));
  return 1;
}

const a00_tree_t * a00_tree(const a00_driver_t * d, unsigned i) { return d->trees + i; }

/* ---- step assembly */
static void step_begin(a00_driver_t * d) { d->s_br_off[0] = d->s_nd_off[0] = 0; }
static void reserve(a00_driver_t * d, size_t br, size_t nd)
{
  if (br > d->cap_br) { d->cap_br = 2*br + 1024; d->s_br = (int *)realloc(d->s_br, d->cap_br*sizeof(int)); }
  if (nd > d->cap_nd) { d->cap_nd = 2*nd + 1024; d->s_nd = (int *)realloc(d->s_nd, d->cap_nd*sizeof(int)); }
}

/* install a proposal on locus i: toggle the buffers of the changed branches and of the nodes
   to recompute (sorted children-first = by age), append it to the step */
static void step_add(a00_driver_t * d, unsigned slot, unsigned i, const int * branches, int nb,
                     int * nodes, int nn)
{
  a00_tree_t * t = d->trees + i; int a, b;
  /* unique + sort nodes by age (a parent is always older than its children) */
  for (a = 0; a < nn; ++a) for (b = a + 1; b < nn; ++b) if (nodes[b] == nodes[a]) { nodes[b] = nodes[--nn]; --b; }
  for (a = 1; a < nn; ++a) { int v = nodes[a]; for (b = a; b > 0 && t->time[nodes[b-1]] > t->time[v]; --b) nodes[b] = nodes[b-1]; nodes[b] = v; }
  for (a = 0; a < nb; ++a) swap_pmat(t, branches[a]);
  for (a = 0; a < nn; ++a) swap_clv(t, nodes[a]);
  reserve(d, d->s_br_off[slot] + (size_t)nb, d->s_nd_off[slot] + (size_t)nn);
  memcpy(d->s_br + d->s_br_off[slot], branches, (size_t)nb*sizeof(int));
  memcpy(d->s_nd + d->s_nd_off[slot], nodes, (size_t)nn*sizeof(int));
  d->s_locus[slot] = i; d->s_tree[slot] = t;
  d->s_br_off[slot+1] = d->s_br_off[slot] + (unsigned)nb;
  d->s_nd_off[slot+1] = d->s_nd_off[slot] + (unsigned)nn;
}

static int step_eval(a00_driver_t * d, unsigned n)
{
  a00_step_t s;
  if (!n) return 1;
  s.nloci = n; s.locus = d->s_locus; s.tree = d->s_tree; s.br_off = d->s_br_off; s.branches = d->s_br;
  s.nd_off = d->s_nd_off; s.nodes = d->s_nd;
  d->steps++;
  return d->eval(d->ctx, &s, d->s_lnl);
}

static int path_to_root(const a00_tree_t * t, int v, int * out)
{
  int k = 0;
  for (; v >= 0; v = t->parent[v]) out[k++] = v;
  return k;
}

int a00_initialize(a00_driver_t * d)
{
  unsigned i; int br[MAXN], nd[MAXN];
  step_begin(d);
  for (i = 0; i < d->nloci; ++i)
  {
    a00_tree_t * t = d->trees + i; int nb = 0, nn = 0, k;
    for (k = 0; k < t->n; ++k) { if (t->parent[k] >= 0) br[nb++] = k; if (t->left[k] >= 0) nd[nn++] = k; }
    /* start-up evaluates into the current buffers: toggle twice = no toggle */
    for (k = 0; k < nb; ++k) swap_pmat(t, br[k]);
    for (k = 0; k < nn; ++k) swap_clv(t, nd[k]);
    step_add(d, i, i, br, nb, nd, nn);
  }
  if (!step_eval(d, d->nloci)) return 0;
  for (i = 0; i < d->nloci; ++i) d->trees[i].lnl = d->s_lnl[i];
  return 1;
}

/* per-locus Metropolis decision on the likelihood ratio */
static void decide(a00_driver_t * d, unsigned n)
{
  unsigned s;
  for (s = 0; s < n; ++s)
  {
    const unsigned i = d->s_locus[s]; a00_tree_t * t = d->trees + i;
    const double lnacc = d->s_lnl[s] - t->lnl + d->s_hast[s];
    const double u = a00_rndu(&d->rng[i]);
    d->proposals++;
    if (lnacc >= 0 || u < exp(lnacc)) { t->lnl = d->s_lnl[s]; d->accepted++; }
    else restore(d, i);                                  /* swap indices, ages, topology back */
  }
}

/* GAGE: the k-th inner node of every locus (gtree.c:4585) */
static int gage_step(a00_driver_t * d, int k)
{
  unsigned i, n = 0; int br[4], nd[MAXN];
  step_begin(d);
  for (i = 0; i < d->nloci; ++i)
  {
    a00_tree_t * t = d->trees + i; int v = -1, c = 0, j, nb = 0, nn, p; double lo, u;
    for (j = 0; j < t->n; ++j) if (t->left[j] >= 0 && c++ == k) { v = j; break; }
    if (v < 0) continue;
    u = a00_rndu(&d->rng[i]);
    snapshot(d, i);
    lo = fmax(t->time[t->left[v]], t->time[t->right[v]]);
    p = t->parent[v];
    d->s_hast[n] = 0;
    if (p >= 0) t->time[v] = lo + (0.02 + 0.96*u)*(t->time[p] - lo);
    else { const double c_ = exp(0.6*(u - 0.5)); t->time[v] = lo + (t->time[v] - lo)*c_; d->s_hast[n] = log(c_); }
    br[nb++] = t->left[v]; br[nb++] = t->right[v]; if (p >= 0) br[nb++] = v;
    nn = path_to_root(t, v, nd);
    step_add(d, n, i, br, nb, nd, nn);
    ++n;
  }
  if (!step_eval(d, n)) return 0;
  decide(d, n);
  return 1;
}

/* exchange the tree positions of node ids a and b (buffer indices stay with the ids) */
static void swap_ids(a00_tree_t * t, int a, int b)
{
  int i; int L[MAXN], R[MAXN], P[MAXN]; double T[MAXN];
#define M(x) ((x) == a ? b : (x) == b ? a : (x))
  for (i = 0; i < t->n; ++i)
  {
    const int o = M(i);
    L[i] = t->left[o] >= 0 ? M(t->left[o]) : -1; R[i] = t->right[o] >= 0 ? M(t->right[o]) : -1;
    P[i] = t->parent[o] >= 0 ? M(t->parent[o]) : -1; T[i] = t->time[o];
  }
  memcpy(t->left, L, (size_t)t->n*sizeof(int)); memcpy(t->right, R, (size_t)t->n*sizeof(int));
  memcpy(t->parent, P, (size_t)t->n*sizeof(int)); memcpy(t->time, T, (size_t)t->n*sizeof(double));
  t->root = M(t->root);
#undef M
}

/* GSPR: the k-th non-root node of every locus is pruned and regrafted (gtree.c:6531) */
static int gspr_step(a00_driver_t * d, int k)
{
  unsigned i, n = 0;
  step_begin(d);
  for (i = 0; i < d->nloci; ++i)
  {
    a00_tree_t * t = d->trees + i;
    int a = -1, c = 0, j, p, s, g, pc, tgt, ntg = 0, targets[MAXN], banned[MAXN], stack[MAXN], sp = 0;
    int bset[4], br[4], nb = 0, nd[2*MAXN], nn = 0, root_before;
    double lo, tnew, u1, u2;
    for (j = 0; j < t->n; ++j) if (j != t->root && c++ == k) { a = j; break; }
    if (a < 0) continue;
    u1 = a00_rndu(&d->rng[i]); u2 = a00_rndu(&d->rng[i]);
    snapshot(d, i);
    root_before = t->root;
    p = t->parent[a]; s = t->left[p] == a ? t->right[p] : t->left[p]; g = t->parent[p];
    /* prune: the sibling takes p's place */
    t->parent[s] = g;
    if (g >= 0) { if (t->left[g] == p) t->left[g] = s; else t->right[g] = s; } else t->root = s;
    /* regraft target: any node outside a's subtree (and not p) */
    memset(banned, 0, (size_t)t->n*sizeof(int)); banned[p] = 1; stack[sp++] = a;
    while (sp) { const int x = stack[--sp]; banned[x] = 1; if (t->left[x] >= 0) { stack[sp++] = t->left[x]; stack[sp++] = t->right[x]; } }
    for (j = 0; j < t->n; ++j) if (!banned[j]) targets[ntg++] = j;
    tgt = targets[(int)(u1*ntg) % ntg];
    pc = t->parent[tgt];
    lo = fmax(t->time[a], t->time[tgt]);
    if (pc >= 0 && t->time[pc] <= lo) { tgt = s; pc = t->parent[s]; lo = fmax(t->time[a], t->time[tgt]); }
    tnew = pc >= 0 ? lo + (0.02 + 0.96*u2)*(t->time[pc] - lo) : lo + (0.1 + u2)*fmax(lo, 1e-4)*0.5;
    t->time[p] = tnew; t->left[p] = a; t->right[p] = tgt; t->parent[a] = p; t->parent[tgt] = p; t->parent[p] = pc;
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
    d->s_hast[n] = 0;
    step_add(d, n, i, br, nb, nd, nn);
    ++n;
  }
  if (!step_eval(d, n)) return 0;
  decide(d, n);
  return 1;
}

int a00_set_taus(a00_driver_t * d, const double * taus, unsigned n)
{
  unsigned i;
  if (n > 8) return 0;
  for (i = 0; i < n; ++i) d->taus[i] = taus[i];
  d->ntaus = n;
  return 1;
}
unsigned a00_get_taus(const a00_driver_t * d, double * taus)
{
  unsigned i;
  for (i = 0; i < d->ntaus; ++i) taus[i] = d->taus[i];
  return d->ntaus;
}

/* TAU j: rubber-band rescaling of the gene-node ages around tau_j in every locus (stree.c:5512);
   loci without a node in the band are left out of the step; ONE decision for all loci */
static int tau_step(a00_driver_t * d, unsigned j)
{
  unsigned i, n = 0; double sum = 0;
  const double tau = d->taus[j], lo = j ? d->taus[j-1] : 0.0, hi = j + 1 < d->ntaus ? d->taus[j+1] : -1.0;
  const double tnew = a00_tau_proposal(a00_rndu(&d->grng), lo, tau, hi);
  const double uacc = a00_rndu(&d->grng);
  step_begin(d);
  for (i = 0; i < d->nloci; ++i)
  {
    a00_tree_t * t = d->trees + i; int br[MAXN], nd[MAXN], nb = 0, nn = 0, k, v, moved = 0;
    char isbr[MAXN], isnd[MAXN];
    snapshot(d, i);
    memset(isbr, 0, (size_t)t->n); memset(isnd, 0, (size_t)t->n);
    for (k = 0; k < t->n; ++k)
      if (t->left[k] >= 0)
      {
        const double tn = a00_rubber_band(t->time[k], lo, tau, tnew, hi);
        if (tn != t->time[k])
        {
          t->time[k] = tn; ++moved;
          isbr[t->left[k]] = isbr[t->right[k]] = 1; if (t->parent[k] >= 0) isbr[k] = 1;
          for (v = k; v >= 0; v = t->parent[v]) isnd[v] = 1;              /* gtree_return_partials */
        }
      }
    if (!moved) continue;
    for (k = 0; k < t->n; ++k) { if (isbr[k]) br[nb++] = k; if (isnd[k]) nd[nn++] = k; }
    step_add(d, n, i, br, nb, nd, nn);
    ++n;
  }
  if (!step_eval(d, n)) return 0;
  for (i = 0; i < n; ++i) sum += d->s_lnl[i] - d->trees[d->s_locus[i]].lnl;
  d->proposals++;
  if (sum >= 0 || uacc < exp(sum))
  {
    d->accepted++; d->taus[j] = tnew;
    for (i = 0; i < n; ++i) d->trees[d->s_locus[i]].lnl = d->s_lnl[i];
  }
  else for (i = 0; i < n; ++i) restore(d, d->s_locus[i]);
  return 1;
}

/* MIX: every age of every locus times c; ONE decision from the summed difference (prop_mixing.c:203-205) */
static int mix_step(a00_driver_t * d)
{
  unsigned i; int br[MAXN], nd[MAXN]; double sum = 0, lnacc; long ninner = 0;
  const double lnc = 0.1*(a00_rndu(&d->grng) - 0.5), c = exp(lnc);
  const double uacc = a00_rndu(&d->grng);
  step_begin(d);
  for (i = 0; i < d->nloci; ++i)
  {
    a00_tree_t * t = d->trees + i; int nb = 0, nn = 0, k;
    snapshot(d, i);
    for (k = 0; k < t->n; ++k)
    {
      if (t->left[k] >= 0) { t->time[k] *= c; nd[nn++] = k; ++ninner; }
      if (t->parent[k] >= 0) br[nb++] = k;
    }
    step_add(d, i, i, br, nb, nd, nn);
  }
  if (!step_eval(d, d->nloci)) return 0;
  for (i = 0; i < d->nloci; ++i) sum += d->s_lnl[i] - d->trees[i].lnl;
  lnacc = sum + (double)ninner*lnc;                       /* multiplier proposal on ninner ages */
  d->proposals++;
  if (lnacc >= 0 || uacc < exp(lnacc))
  {
    d->accepted++;
    for (i = 0; i < d->nloci; ++i) d->trees[i].lnl = d->s_lnl[i];
    for (i = 0; i < d->ntaus; ++i) d->taus[i] *= c;                  /* the species tree is scaled too */
  }
  else for (i = 0; i < d->nloci; ++i) restore(d, i);
  return 1;
}

int a00_iterate(a00_driver_t * d)
{
  unsigned i; int k, maxtips = 0;
  for (i = 0; i < d->nloci; ++i) if (d->trees[i].tips > maxtips) maxtips = d->trees[i].tips;
  for (k = 0; k < maxtips - 1; ++k)   if (!gage_step(d, k)) return 0;
  for (k = 0; k < 2*maxtips - 2; ++k) if (!gspr_step(d, k)) return 0;
  for (k = 0; k < (int)d->ntaus; ++k) if (!tau_step(d, (unsigned)k)) return 0;
  return mix_step(d);
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
int a00_backend_hip(void * vctx, const a00_step_t * s, double * lnl)
{
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
  ok = bpa_batch_evaluate(c->engine, &b, lnl);
  free(loci); free(mp); free(ml); free(ops); free(rc); free(rs);
  return ok;
}
