// bigsampler.hpp — the device-resident A00 sampler for loci beyond the generic sampler's 16 tips (gsampler.hpp): up to 64 tips,
// any 4-state model of the engine, scalers, unphased diploid loci — BASELINE config 1's kind (examples/frogs: 5 loci of 42-60
// tips after phasing).  Same API (bpa_sampler_t, chosen at creation), same moves, same random streams and the same trajectory
// as the C host driver (csrc/host/a00_driver.c) — the proposal code here IS the host driver's, statement for statement, run
// by one lane per locus on trees that live in HBM (plain loops over node arrays: no register trees, no node masks; with a
// handful of loci per data set the proposal kernel is not where the time goes).  A step is written, on the device, as the
// records of the engine's general kernels — per locus a range of OpDev node updates with their scaler indices, the fresh
// branches as entries of the P-matrix list (unused entries are holes), the root's buffer and scaler indices — and evaluated
// by pmatrix_s4_kernel, partials_lnl_s4_kernel and lnl_reduce_wave_kernel (which averages the phase resolutions of a
// diploid locus, locus.c:2586-2615).  One proposal step of all loci:
//
//     big_step_kernel   settle the previous step (accept / roll back), propose, MSC density, records
//     P-matrices, node updates + per-pattern terms, per-locus sums                     -> lnL per locus
//     all-loci steps only: gsum_decide_kernel (sum of the per-locus terms, ONE decision)
//
// Reference: gtree.c:4585 (ages), 6531 (SPR), stree.c:5512 / 4338 (tau + rubber band), prop_mixing.c:52, gtree.c:3957.
#pragma once

namespace gbig {
using smp::Species; using smp::MAXPOP; using smp::rndu; using smp::reflect;

constexpr int BT = 64;                    // tips per locus
constexpr int BN = 2*BT;                  // node slots (2 tips - 1 used)
constexpr int BBS = 64;                   // loci per workgroup

struct BTree                              // one per locus, in HBM (and its copy on the host)
{
  int16_t  left[BN], right[BN], parent[BN], clv[BN], pmat[BN], pop[BN], scaler[BN];
  double   time[BN];
  double   lnl, logpr;
  a00_rng_t rng;
  int32_t  root, tips;
  uint32_t proposals, accepted;
  uint32_t work_nupd, work_nbr, work_neval, pad_;
  int8_t   gl[MAXPOP];                    // gene tips below each population (never changes: tips stay in their species)
};

struct BArgs
{
  BTree * trees, * undo;                  // [T] current (or proposed, while a step is being evaluated) / the state before the pending step
  uint32_t T;
  uint32_t mode;                          // 0 GAGE k, 1 GSPR k, 2 TAU, 3 MIX, 4 settle (+ THETA statistics), 5 start-up evaluation
  uint32_t k;
  uint32_t pend;                          // the step to settle first: 0 none, 1 per-locus decisions, 2 an all-loci decision (flag / epoch), 3 commit (start-up)
  const double * lnl_new;                 // [T] lnL of the pending step's evaluation (task = locus)
  double * hast, * logpr_new;             // [T] Hastings term / proposed MSC density of the step being proposed
  double * delta;                         // [T] an all-loci step: this locus's density + Jacobian term
  double * lnl_cur;                       // [T] ... and its current lnL (the sum kernel reads compact arrays)
  uint8_t * active;                       // [T] the locus has a likelihood evaluation pending
  const uint32_t * flag; uint32_t epoch;  // an all-loci step was REJECTED when *flag == epoch
  // the step's records: ops[op_rng[2 i] .. op_rng[2 i + 1]), root_clv[i], root_scaler[i]; matrix entries [i maxmat + j]
  OpDev * ops; uint32_t * op_rng, * root_clv; int32_t * root_scaler;
  uint32_t * mat_task, * mat_pm; double * mat_length;
  uint32_t maxmat, maxops;
  const double * taus;                    // [3 MAXPOP] tau | theta | log(2/theta)
  uint32_t tau_q; double tau_u, mix_c, mix_lnc;
  int8_t * pop_nc; double * pop_t2h;      // [MAXPOP][T] mode 4: the statistics the THETA kernel reads
  uint32_t refresh_logpr;
  Species sp;
};

// lane 0's work arrays, in LDS (as private arrays they would live in scratch memory: a round trip to HBM per access)
struct Work
{
  int nd[2*BN], br[BN], targets[BN], stack[BN];
  int gl[MAXPOP], nin[MAXPOP], nc[MAXPOP];
  double times[BN];
  unsigned char isbr[BN], isnd[BN];
  // a prune-and-regraft between its two lane-0 parts: what the scan of all nodes (64 lanes) needs and leaves
  int g_a, g_p, g_s, g_g, g_popt, g_popp, g_ok;
  double g_tnew, g_tp, g_u2;
  unsigned long long g_tmask[2], g_smask[2];
};

// ---- the host driver's helpers (a00_driver.c: swap_clv / swap_pmat, lca_pop, climb, tree_logpr_stats, install_local)
__device__ inline void swap_clv(BTree & t, int i)
{
  const int inner = t.tips - 1;
  t.clv[i] = (int16_t)(t.tips + (t.clv[i] - t.tips + inner) % (2*inner));
  if (t.scaler[i] != BPA_SCALE_BUFFER_NONE) t.scaler[i] = (int16_t)((t.scaler[i] + inner) % (2*inner));
}
__device__ inline void swap_pmat(BTree & t, int i)
{
  const int edges = 2*t.tips - 2;
  t.pmat[i] = (int16_t)((t.pmat[i] + edges) % (2*edges));
}
__device__ inline int lca_pop(const Species & sp, int p, int q)
{
  while (!((sp.anc[q] >> p) & 1u)) p = sp.parent[p];
  return p;
}
__device__ inline int climb(const Species & sp, const double * tau, int p, double t)
{
  while (sp.parent[p] >= 0 && tau[sp.parent[p]] <= t) p = sp.parent[p];
  return p;
}
// gtree_logprob (gtree.c:3957) = the sum over populations, in stree->nodes order, of gtree_update_logprob_contrib; NaN if the
// tree does not fit the species tree.  Optionally leaves the coalescence counts and T2h of every population (THETA)
__device__ double tree_logpr(const BTree & t, const Species & sp, const double * tau, int8_t * nc_out, double * t2h_out, uint32_t stride, Work & W)
{
  int * nin = W.nin, * nc = W.nc;
  double * times = W.times, logpr = 0;
  const int n = 2*t.tips - 1;
  for (int p = 0; p < sp.npop; ++p) nin[p] = 0;
  for (int k = 0; k < t.tips; ++k) nin[t.pop[k]]++;
  for (int p = 0; p < sp.npop; ++p)
  {
    int cnt = 0;
    if (p >= sp.S) nin[p] = (nin[sp.left[p]] - nc[sp.left[p]]) + (nin[sp.right[p]] - nc[sp.right[p]]);
    for (int k = t.tips; k < n; ++k)
      if (t.pop[k] == p)
      {
        const double v = t.time[k];
        if (v < tau[p] || (sp.parent[p] >= 0 && v >= tau[sp.parent[p]])) return __longlong_as_double(0x7ff8000000000000ll);
        int a = cnt++;
        for (; a > 0 && times[a-1] > v; --a) times[a] = times[a-1];
        times[a] = v;
      }
    nc[p] = cnt;
    if (cnt >= nin[p] && cnt > 0) return __longlong_as_double(0x7ff8000000000000ll);
    // a00_msc_t2h / a00_msc_term (bpp_amd_host.h), heredity 1
    const double ptau = sp.parent[p] >= 0 ? tau[sp.parent[p]] : -1.0;
    int steps = cnt + (ptau >= 0 ? 1 : 0);
    if (nin[p] == steps) --steps;
    double T2h = 0, prev = tau[p];
    int m = nin[p];
    for (int k = 0; k < steps; ++k, --m)
    {
      const double tk = k < cnt ? times[k] : ptau;
      T2h += m*(m - 1)*(tk - prev);
      prev = tk;
    }
    double c = 0;
    if (cnt) c += cnt*tau[2*MAXPOP + p];
    if (T2h) c -= T2h/(tau[MAXPOP + p]*1.0);
    logpr += c;
    if (nc_out) { nc_out[(size_t)p*stride] = (int8_t)cnt; t2h_out[(size_t)p*stride] = T2h; }
  }
  return logpr;
}
// install a proposal: the nodes to recompute unique and sorted by age (children first), the buffers of the changed branches
// and of those nodes toggled
__device__ void install(BTree & t, const int * br, int nb, int * nd, int & nn)
{
  for (int a = 0; a < nn; ++a) for (int b = a + 1; b < nn; ++b) if (nd[b] == nd[a]) { nd[b] = nd[--nn]; --b; }
  for (int a = 1; a < nn; ++a) { const int v = nd[a]; int b = a; for (; b > 0 && t.time[nd[b-1]] > t.time[v]; --b) nd[b] = nd[b-1]; nd[b] = v; }
  for (int a = 0; a < nb; ++a) swap_pmat(t, br[a]);
  for (int a = 0; a < nn; ++a) swap_clv(t, nd[a]);
}
__device__ inline int path_to_root(const BTree & t, int v, int * out)
{
  int k = 0;
  for (; v >= 0; v = t.parent[v]) out[k++] = v;
  return k;
}
__device__ int count_tips(const BTree & t, int v, int * stack)
{
  // (iterative: a stack of the subtree's pending nodes)
  int top = 0, tips = 0;
  stack[top++] = v;
  while (top)
  {
    const int x = stack[--top];
    if (t.left[x] < 0) ++tips; else { stack[top++] = t.left[x]; stack[top++] = t.right[x]; }
  }
  return tips;
}
// exchange the tree positions of node ids a and b (buffer indices stay with the ids) — swap_ids of a00_driver.c, in place:
// every reference to a or b renamed, then the two rows exchanged (new row i = old row M(i) with M-renamed references)
__device__ void swap_ids(BTree & t, int a, int b)
{
  const int n = 2*t.tips - 1;
#define BIGM(x) ((x) == a ? b : (x) == b ? a : (x))
  for (int i = 0; i < n; ++i)
  {
    const int l = t.left[i], r = t.right[i], p = t.parent[i];
    if (l >= 0) t.left[i] = (int16_t)BIGM(l);
    if (r >= 0) t.right[i] = (int16_t)BIGM(r);
    if (p >= 0) t.parent[i] = (int16_t)BIGM(p);
  }
  { const int16_t x = t.left[a];   t.left[a] = t.left[b];     t.left[b] = x; }
  { const int16_t x = t.right[a];  t.right[a] = t.right[b];   t.right[b] = x; }
  { const int16_t x = t.parent[a]; t.parent[a] = t.parent[b]; t.parent[b] = x; }
  { const int16_t x = t.pop[a];    t.pop[a] = t.pop[b];       t.pop[b] = x; }
  { const double x = t.time[a];    t.time[a] = t.time[b];     t.time[b] = x; }
  t.root = BIGM(t.root);
#undef BIGM
}

// GSPR, lane 0's first part (gspr_step of a00_driver.c up to the target / source scans): the pruned node, the draws, the new
// age of its father and the population it falls in
__device__ void gspr_pre(const BArgs & A, BTree & t, const Species & sp, const double * s_tau, Work & W)
{
  const int n = 2*t.tips - 1;
  int a = -1, c = 0;
  for (int j = 0; j < n; ++j) if (j != t.root && c++ == (int)A.k) { a = j; break; }
  W.g_a = a; W.g_ok = 0;
  if (a < 0) return;
  const double u1 = rndu(&t.rng) - 0.5, u2 = rndu(&t.rng);
  const int p = t.parent[a], s = t.left[p] == a ? t.right[p] : t.left[p], g = t.parent[p];
  // youngest population from a's upwards that holds gene tips outside a's subtree (gtree.c:6664-6669)
  const int leaves = count_tips(t, a, W.stack);
  int pop0 = t.pop[a];
  for (; t.gl[pop0] <= leaves && sp.parent[pop0] >= 0; pop0 = sp.parent[pop0]) ;
  const double lo = fmax(t.time[a], s_tau[pop0]);
  const double tnew = reflect(t.time[p] + sp.ft_gspr*u1, lo, 999.0);
  W.g_p = p; W.g_s = s; W.g_g = g; W.g_popt = climb(sp, s_tau, t.pop[a], tnew); W.g_popp = t.pop[p];
  W.g_tnew = tnew; W.g_tp = t.time[p]; W.g_u2 = u2; W.g_ok = 1;
}
// ... the scans, by all 64 lanes (two nodes each): the branches crossing the new age inside the target population, and
// the branches the reverse move could pick at the old age (gtree.c:6760-6775), as bit masks in node order
__device__ void gspr_scan(const BTree & t, const Species & sp, Work & W, const uint32_t lane)
{
  const int n = 2*t.tips - 1, a = W.g_a, p = W.g_p, s = W.g_s, root = t.root;
  const double tnew = W.g_tnew, tp = W.g_tp;
  for (int h = 0; h < 2; ++h)
  {
    const int j = 64*h + (int)lane;
    bool ct = false, cs = false;
    if (j < n && j != a && j != root)
    {
      const double tj = t.time[j], tpar = t.time[t.parent[j]];
      const uint32_t anc = sp.anc[t.pop[j]];
      ct = tj <= tnew && tpar > tnew && ((anc >> W.g_popt) & 1u);
      cs = j != s && j != p && tj <= tp && tpar > tp && ((anc >> W.g_popp) & 1u);
    }
    const unsigned long long mt = __ballot(ct), ms = __ballot(cs);
    if (lane == 0) { W.g_tmask[h] = mt; W.g_smask[h] = ms; }
  }
}

// ---- 4. propose (gage_step / gspr_step / tau_step / mix_step / a00_initialize of a00_driver.c, one locus), then 5. the records
__device__ void big_propose(const BArgs & A, const uint32_t i, BTree & t, const Species & sp, const double * s_tau, const uint32_t MODE,
                            const double lminf, const double lmaxf, const double tq_old, const double tq_lo, const double tq_hi,
                            const double minf, const double maxf, Work & W)
{
  const int n = 2*t.tips - 1;
  bool evaluate = false;
  int * br = W.br, * nd = W.nd, nb = 0, nn = 0;
  if (MODE == 4)
  {
    A.active[i] = 0;
    (void)tree_logpr(t, sp, s_tau, A.pop_nc + i, A.pop_t2h + i, A.T, W);
    return;
  }
  if (MODE == 0)
  {
    int v = -1, c = 0;
    for (int j = 0; j < n; ++j) if (t.left[j] >= 0 && c++ == (int)A.k) { v = j; break; }
    if (v >= 0)
    {
      const double u = rndu(&t.rng) - 0.5;
      const int l = t.left[v], r = t.right[v], p = t.parent[v];
      double lo = fmax(t.time[l], t.time[r]);
      if (t.pop[l] != t.pop[r]) lo = fmax(lo, s_tau[lca_pop(sp, t.pop[l], t.pop[r])]);
      const double hi = p >= 0 ? t.time[p] : 999.0;
      if (!(hi > lo)) (void)rndu(&t.rng);
      else
      {
        
        const double tnew = reflect(t.time[v] + sp.ft_gage*u, lo, hi);
        t.time[v] = tnew;
        t.pop[v] = (int16_t)climb(sp, s_tau, t.pop[l], tnew);
        br[nb++] = l; br[nb++] = r; if (p >= 0) br[nb++] = v;
        nn = path_to_root(t, v, nd);
        install(t, br, nb, nd, nn);
        A.hast[i] = 0.0; A.logpr_new[i] = tree_logpr(t, sp, s_tau, nullptr, nullptr, 0, W);
        evaluate = true;
      }
    }
  }
  else if (MODE == 1)
  {
    const int a = W.g_a;
    if (W.g_ok)
    {
      const double u2 = W.g_u2, tnew = W.g_tnew;
      const int p = W.g_p, s = W.g_s, g = W.g_g, popt = W.g_popt;
      int * targets = W.targets, ntg = 0, nsrc = 1;
      if (tnew >= t.time[t.root]) targets[ntg++] = t.root;
      else
        for (int h = 0; h < 2; ++h)
          for (unsigned long long m = W.g_tmask[h]; m; m &= m - 1) { const int j = 64*h + __ffsll((long long)m) - 1; targets[ntg++] = j == p ? s : j; }
      if (p != t.root) nsrc += __popcll(W.g_smask[0]) + __popcll(W.g_smask[1]);
      if (!ntg) (void)rndu(&t.rng);
      else
      {
        int tgt = targets[(int)(u2*ntg) % ntg];
        if (tgt == p) tgt = s;
        
        const int root_before = t.root;
        t.parent[s] = (int16_t)g;
        if (g >= 0) { if (t.left[g] == p) t.left[g] = (int16_t)s; else t.right[g] = (int16_t)s; } else t.root = s;
        const int pc = t.parent[tgt];
        t.time[p] = tnew; t.pop[p] = (int16_t)popt;
        t.left[p] = (int16_t)a; t.right[p] = (int16_t)tgt; t.parent[a] = (int16_t)p; t.parent[tgt] = (int16_t)p; t.parent[p] = (int16_t)pc;
        if (pc >= 0) { if (t.left[pc] == tgt) t.left[pc] = (int16_t)p; else t.right[pc] = (int16_t)p; } else t.root = p;
        nn = path_to_root(t, p, nd);
        if (g >= 0) nn += path_to_root(t, g, nd + nn);
        int bset[4] = { a, tgt, p, s };
        if (t.root != root_before)
        {
          const int newtop = t.root;
          swap_ids(t, newtop, root_before);
          for (int j = 0; j < nn; ++j) nd[j] = nd[j] == newtop ? root_before : nd[j] == root_before ? newtop : nd[j];
          for (int j = 0; j < 4; ++j) bset[j] = bset[j] == newtop ? root_before : bset[j] == root_before ? newtop : bset[j];
          nn += path_to_root(t, newtop, nd + nn);
        }
        for (int j = 0; j < 4; ++j)
        {
          bool dup = false;
          for (int q = 0; q < nb; ++q) if (br[q] == bset[j]) dup = true;
          if (!dup && t.parent[bset[j]] >= 0) br[nb++] = bset[j];
        }
        install(t, br, nb, nd, nn);
        A.hast[i] = log((double)ntg/(double)nsrc); A.logpr_new[i] = tree_logpr(t, sp, s_tau, nullptr, nullptr, 0, W);
        evaluate = true;
      }
    }
  }
  else if (MODE == 2)
  {
    const int q = (int)A.tau_q, cl = sp.left[q], cr = sp.right[q];
    int above = 0, below = 0;
    
    unsigned char * isbr = W.isbr, * isnd = W.isnd;
    for (int k = 0; k < n; ++k) isbr[k] = isnd[k] = 0;
    for (int k = t.tips; k < n; ++k)
    {
      const int pk = t.pop[k]; const double tk = t.time[k];
      if ((pk != q && pk != cl && pk != cr) || tk < tq_lo || tk > tq_hi) continue;
      if (tk >= tq_old) { t.time[k] = tq_hi + maxf*(tk - tq_hi); ++above; } else { t.time[k] = tq_lo + minf*(tk - tq_lo); ++below; }
      isbr[t.left[k]] = isbr[t.right[k]] = 1; if (t.parent[k] >= 0) isbr[k] = 1;
      for (int v = k; v >= 0; v = t.parent[v]) isnd[v] = 1;
    }
    const double lp_new = tree_logpr(t, sp, s_tau, nullptr, nullptr, 0, W);
    A.logpr_new[i] = lp_new;
    A.delta[i] = ((lp_new - t.logpr) + below*lminf) + above*lmaxf;
    A.lnl_cur[i] = t.lnl;
    if (above + below)
    {
      for (int k = 0; k < n; ++k) { if (isbr[k]) br[nb++] = k; if (isnd[k]) nd[nn++] = k; }
      install(t, br, nb, nd, nn);
      evaluate = true;
    }
  }
  else
  {
    // mixing or start-up: every branch, every inner node
    int ninner = 0;
    for (int k = 0; k < n; ++k)
    {
      if (t.left[k] >= 0) { if (MODE == 3) t.time[k] *= A.mix_c; nd[nn++] = k; ++ninner; }
      if (t.parent[k] >= 0) br[nb++] = k;
    }
    if (MODE == 5)
    {
      for (int k = 0; k < nb; ++k) swap_pmat(t, br[k]);          // start-up evaluates in place: toggle twice = no toggle
      for (int k = 0; k < nn; ++k) swap_clv(t, nd[k]);
    }
    const double lp_new = tree_logpr(t, sp, s_tau, nullptr, nullptr, 0, W);
    A.logpr_new[i] = lp_new;
    A.delta[i] = (lp_new - t.logpr) + (double)ninner*A.mix_lnc;
    A.lnl_cur[i] = t.lnl;
    install(t, br, nb, nd, nn);
    evaluate = true;
  }
  A.active[i] = evaluate ? 1 : 0;
  if (evaluate && MODE != 5) { t.work_nupd += (uint32_t)nn; t.work_nbr += (uint32_t)nb; t.work_neval++; }

  // ---- 5. the step's records (what a00_backend_hip marshals on the host)
  const uint32_t e0 = i*A.maxmat, o0 = i*A.maxops;
  uint32_t nm = 0;
  if (evaluate)
  {
    for (int j = 0; j < nb; ++j, ++nm)
    {
      const int x = br[j];
      A.mat_task[e0 + nm] = i; A.mat_pm[e0 + nm] = (uint32_t)t.pmat[x];
      A.mat_length[e0 + nm] = (t.time[t.parent[x]] - t.time[x])*1.0;        // rate_mui = 1 (locus.c:2350)
    }
    for (int j = 0; j < nn; ++j)
    {
      const int x = nd[j], l = t.left[x], r = t.right[x];
      OpDev q;
      q.parent_clv = (uint32_t)t.clv[x]; q.parent_scaler = t.scaler[x];
      q.left_clv = (uint32_t)t.clv[l]; q.left_pmatrix = (uint32_t)t.pmat[l]; q.left_scaler = t.scaler[l];
      q.right_clv = (uint32_t)t.clv[r]; q.right_pmatrix = (uint32_t)t.pmat[r]; q.right_scaler = t.scaler[r];
      A.ops[o0 + j] = q;
    }
  }
  for (; nm < A.maxmat; ++nm) A.mat_task[e0 + nm] = 0xffffffffu;
  A.op_rng[2*i] = o0; A.op_rng[2*i + 1] = o0 + (evaluate ? (uint32_t)nn : 0u);
  A.root_clv[i] = (uint32_t)t.clv[t.root]; A.root_scaler[i] = t.scaler[t.root];
}

// One workgroup (one wave) per locus: the 64 lanes move the locus's tree between HBM and LDS and take the roll-back copies,
// lane 0 runs the host driver's proposal code on the LDS copy (a lone lane working on a tree in HBM waits a memory round
// trip per node access: 130 us per step on the frogs loci, against 15 with the tree in LDS)
constexpr uint32_t NODE_WORDS = (7*BN*sizeof(int16_t) + BN*sizeof(double))/4;      // the node arrays of a BTree, as 32-bit words
static_assert(offsetof(BTree, lnl) == NODE_WORDS*4, "BTree: the node arrays come first");
__device__ inline void copy_nodes(BTree & dst, const BTree & src, uint32_t lane)
{
  const uint32_t * a = reinterpret_cast<const uint32_t *>(&src);
  uint32_t * b = reinterpret_cast<uint32_t *>(&dst);
  for (uint32_t q = lane; q < NODE_WORDS; q += BBS) b[q] = a[q];
  if (lane == 0) { dst.root = src.root; dst.tips = src.tips; }
}

__global__ void __launch_bounds__(BBS) big_step_kernel(const BArgs A)
{
  __shared__ double s_tau[3*MAXPOP];
  __shared__ Species s_sp;
  __shared__ BTree s_t;
  __shared__ Work s_w;
  __shared__ int s_back;
  const uint32_t lane = threadIdx.x, i = blockIdx.x;
  {
    const uint32_t * src = reinterpret_cast<const uint32_t *>(&A.sp);
    uint32_t * dst = reinterpret_cast<uint32_t *>(&s_sp);
    for (uint32_t q = lane; q < sizeof(Species)/4; q += BBS) dst[q] = src[q];
    const uint32_t * tsrc = reinterpret_cast<const uint32_t *>(A.trees + i);
    uint32_t * tdst = reinterpret_cast<uint32_t *>(&s_t);
    for (uint32_t q = lane; q < sizeof(BTree)/4; q += BBS) tdst[q] = tsrc[q];
  }
  if (lane < (uint32_t)(3*MAXPOP)) s_tau[lane] = A.taus[lane];
  if (lane == 0) s_back = 0;
  __syncthreads();
  const Species & sp = s_sp;
  const uint32_t MODE = A.mode;
  BTree & t = s_t;

  // ---- 1. settle the step whose evaluation just finished
  if (lane == 0 && A.pend)
  {
    bool back = false;
    if (A.pend == 1)
    {
      if (A.active[i])
      {
        const double lnl = A.lnl_new[i], lp_new = A.logpr_new[i];
        const double lnacc = (lp_new - t.logpr) + (lnl - t.lnl) + A.hast[i];
        const double u = rndu(&t.rng);
        t.proposals++;
        if (lnacc >= 0 || u < exp(lnacc)) { t.lnl = lnl; t.logpr = lp_new; t.accepted++; }
        else back = true;
      }
    }
    else if (A.pend == 2)
    {
      if (*A.flag == A.epoch) back = true;
      else { t.logpr = A.logpr_new[i]; if (A.active[i]) t.lnl = A.lnl_new[i]; }
    }
    else { t.lnl = A.lnl_new[i]; t.logpr = A.logpr_new[i]; }
    s_back = back ? 1 : 0;
  }
  __syncthreads();
  if (s_back) copy_nodes(t, A.undo[i], lane);
  __syncthreads();
  // ---- 2. THETA moved the thetas since this density was stored
  if (lane == 0 && A.refresh_logpr) t.logpr = tree_logpr(t, sp, s_tau, nullptr, nullptr, 0, s_w);
  __syncthreads();

  // ---- 3. the proposed species tree of an all-loci step is this workgroup's copy of the taus
  double lminf = 0, lmaxf = 0, tq_old = 0, tq_lo = 0, tq_hi = 0, minf = 1, maxf = 1;
  if (MODE == 2)
  {
    const int q = (int)A.tau_q, pq = sp.parent[q];
    tq_old = s_tau[q]; tq_lo = fmax(s_tau[sp.left[q]], s_tau[sp.right[q]]); tq_hi = pq >= 0 ? s_tau[pq] : 999.0;
    const double tnew = reflect(tq_old + sp.ft_tau*(A.tau_u - 0.5), tq_lo, tq_hi);
    minf = (tnew - tq_lo)/(tq_old - tq_lo); maxf = (tnew - tq_hi)/(tq_old - tq_hi);
    lminf = log(minf); lmaxf = log(maxf);
    __syncthreads();
    if (lane == 0) s_tau[q] = tnew;
    __syncthreads();
  }
  else if (MODE == 3)
  {
    if (lane < (uint32_t)sp.npop) s_tau[lane] *= A.mix_c;
    __syncthreads();
  }
  // the state a rejection comes back to (every proposing mode; a step that proposes nothing for this locus never reads it)
  if (MODE <= 3) copy_nodes(A.undo[i], t, lane);
  __syncthreads();
  if (MODE == 1)
  {
    if (lane == 0) gspr_pre(A, t, sp, s_tau, s_w);
    __syncthreads();
    if (s_w.g_ok) gspr_scan(t, sp, s_w, lane);
    __syncthreads();
  }
  if (lane == 0) big_propose(A, i, t, sp, s_tau, MODE, lminf, lmaxf, tq_old, tq_lo, tq_hi, minf, maxf, s_w);
  __syncthreads();
  {
    const uint32_t * tsrc = reinterpret_cast<const uint32_t *>(&s_t);
    uint32_t * tdst = reinterpret_cast<uint32_t *>(A.trees + i);
    for (uint32_t q = lane; q < sizeof(BTree)/4; q += BBS) tdst[q] = tsrc[q];
  }
}

} // namespace gbig
