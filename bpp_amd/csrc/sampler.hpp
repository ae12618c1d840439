// sampler.hpp — device-resident per-locus proposal control.
//
// The C host driver (csrc/host/a00_driver.c) sweeps every locus with GAGE / GSPR proposals,
// one batched launch + one host round trip per proposal.  Here the same state machine runs
// on the device: the gene tree of a locus (the gnode_t fields), its CLV buffers and its P-matrix
// (a,b) table are loaded into LDS once, ALL per-locus proposals of an iteration (tips-1 age
// moves, 2 tips-2 prune/regraft moves) are proposed, evaluated, accepted or rolled back there —
// same arithmetic, same per-locus random stream, same buffer toggling as the host driver — and
// the state goes back to HBM once.  The all-loci mixing step is a second launch whose single
// decision is taken on the device from the summed log-likelihood difference.  An iteration is
// 4 launches instead of 3 tips - 2 + 1 host round trips.
//
// Scope (round 1): JC69, one rate category, no scalers, up to 8 tips / 8 species, loci of <= 64 patterns.
// Parity: identical trajectory to the host driver on libbpp_amd.so, which is identical to the
// host driver on the real reference (tests/test_gpu_sampler.py).
//
// Included at the end of engine.hip (same translation unit: uses its engine/locus structs).
#pragma once
#include "bpp_amd_host.h"

namespace smp {

constexpr int MAXTIPS = 8;
constexpr int MAXN    = 16;               // 2*MAXTIPS-1 nodes, padded
constexpr int MAXBUF  = 2*(MAXTIPS - 1);  // inner CLV buffers
constexpr int MAXPM   = 2*(2*MAXTIPS - 2);// P-matrix buffers
constexpr int MAXPOP  = 16;               // A00_MAXPOP populations, padded
constexpr int TPB     = 16;               // loci per workgroup (64 lanes)
constexpr int BS      = 64;

struct Tree                                // one per locus, in HBM and (while sweeping) in LDS
{
  int8_t   left[MAXN], right[MAXN], parent[MAXN], clv[MAXN], pmat[MAXN], pop[MAXN];
  double   time[MAXN];
  double   lnl, logpr;                     // everything from here on survives a rejected proposal
  a00_rng_t rng;
  int32_t  root, tips;
  uint32_t proposals, accepted;
  uint32_t sw_nupd, sw_nbr;                // work of the sweep launches: node updates run / fresh branches (bpa_sampler_work)
  uint32_t al_nupd, al_nbr, al_neval, pad_; // the same of the all-loci steps + their evaluations (counted by the persistent kernel, sweep2.hpp)
};
static_assert(sizeof(Tree) % 16 == 0 && offsetof(Tree, lnl) % 16 == 0, "Tree is copied as uint4");

// buffer indices of one node update, one byte each in a 64-bit word: parent | left clv | left pmat | right clv | right pmat
typedef uint64_t Op;
__device__ __forceinline__ Op make_op(int parent, int lc, int lp, int rc, int rp)
{
  return (uint64_t)(uint32_t)parent | (uint64_t)(uint32_t)lc << 8 | (uint64_t)(uint32_t)lp << 16 | (uint64_t)(uint32_t)rc << 24 | (uint64_t)(uint32_t)rp << 32;
}

// the species tree (a00_set_species_tree): stree->nodes order, children before parents; the taus live in
// device memory (the TAU and MIX decisions move them), everything else is constant
struct Species
{
  int32_t  S, npop;
  int8_t   parent[MAXPOP], left[MAXPOP], right[MAXPOP];
  uint16_t anc[MAXPOP];                    // bit q: q is p or an ancestor of p
  double   ft_gage, ft_gspr, ft_tau, ft_mix, tau_alpha, tau_beta;
  double   ft_theta, theta_alpha, theta_beta;
  int32_t  program_moves, pad_;            // persistent kernel with BPP's proposal kernel: THETA / TAU / MIX as the program runs them (bpa_sampler_set_program_moves)
  double   theta_slide_prob;               //   share of sliding-window THETA proposals, the rest Gibbs draws
};

struct TaskLDS
{
  Tree   tr;                                      // what the lanes of the locus read; the leader's registers (LTree) are the master copy
  double ab[MAXPM][2];
  Op     ops[MAXBUF];
  int32_t nops, active;                           // active: 0 nothing to evaluate, 1 evaluate + decide, 2 rejected, 3 density only
  double hast;                                    // an all-loci step: this locus's term of the acceptance ratio
  uint32_t brm;                                   // branches whose (a,b) the proposal changes (bit = node below the branch)
  uint32_t wclv;                                  // inner CLV buffers written since the load (bit = buffer): what the store brings back
  uint16_t pnodes[MAXPOP];                        // inner nodes of each population of the proposed tree
  alignas(16) int8_t nin_new[MAXPOP];             // lineages entering / coalescences per population of the proposed tree (the lanes'
  alignas(16) int8_t nc_new[MAXPOP];              // density terms read them; the leader's own copy is in registers)
  double contrib[MAXPOP], contrib_new[MAXPOP];    // per-population terms of the MSC density: current / proposed
  uint32_t chain;                                 // populations whose term the proposal changes
};

// what a lane / a locus needs at every launch and never changes, flattened at upload: one 16-byte load per lane and
// one record per locus instead of the chain lane -> task -> locus table -> parameter block -> weights / tip codes
struct LaneRec                          // task 0xffffffff: idle lane; n | np << 8 | tips << 16
{
  uint32_t task, wgt, tipcodes, n_np_tips;
  double * clv;                         // this lane's pattern in inner CLV buffer 0 of its locus (buffer c: + c*np*4 doubles)
  double * pmat;                        // the locus's (a,b) table — both here so that the loads of the buffers go out one hop
};                                      // after the lane record, not three (lane -> task record -> buffer addresses)
struct TaskRec
{
  double * clv, * pmat;
  double rate, rw, f0, f1, f2, f3;
  alignas(16) int8_t gl[MAXPOP];       // gene tips below each population
  alignas(16) int8_t nin[MAXPOP];      // gene tips of each species (0 for the inner populations)
};

struct Args
{
  const LaneRec * lane_rec;    // [blocks*BS]
  const TaskRec * task_rec;    // [T]
  const uint32_t * blk_task_off;
  Tree * trees, * snap;        // [T] current / pre-step snapshot of an all-loci step
  double * mix_delta;          // [T] this locus's term of the all-loci acceptance ratio
  const uint32_t * mix_flag;   // epoch of the last REJECTED all-loci step
  uint32_t epoch;              // restore from snap when *mix_flag == epoch
  uint32_t mode;               // 0 sweep (GAGE+GSPR), 1 mix, 2 only settle a pending decision, 3 start-up evaluation, 4 tau
  uint32_t nsteps_gage, nsteps_gspr;
  double   mix_c, mix_lnc;
  const double * taus;         // device-resident species-tree parameters: [MAXPOP] tau | [MAXPOP] theta | [MAXPOP] log(2/theta)
  uint32_t tau_q;              // population of the TAU (mode 4) step
  double   tau_u;              // its window uniform
  double * part_out;           // [blocks] an all-loci step's terms summed per workgroup (task order): what the decision kernel adds up
  const double * lograt;       // [MAXN][MAXN] log(i/j): the Hastings ratios of GSPR (lograt_kernel)
  int8_t * pop_nc; double * pop_t2h;      // [MAXPOP][T] sufficient statistics of every locus's density, written by the sweep (population-major:
                                          // the THETA kernels read one population's run of loci)
  uint32_t ntasks;
  uint32_t refresh_logpr;      // the thetas moved since the trees' densities were stored: recompute them from the statistics first
  uint32_t dbg;                // timing experiments only (BPA_SMP_DBG): 1 skip the node updates, 4 skip the proposal, 16 phase cycles of workgroup 0
  double   bfbeta;             // 0 with opt_usedata == 0 (locus.c:2581): the sampler then draws from the MSC prior
  // the decision of an all-loci step inside its own launch (one GPU): every workgroup leaves its sum and a stamp in
  // uncached memory, the LAST workgroup of the grid waits for all stamps, adds the sums up in a fixed order and decides
  // (no other workgroup ever waits: nothing can dead-lock, and the wait is bounded anyway)
  uint32_t dec_on, dec_epoch;  // dec_epoch: this step's epoch = the stamp, and what *dec_flag becomes on rejection
  double   dec_uacc;
  uint32_t * dec_flag, * dec_counters;
  double *   dec_taus;
  unsigned long long * dec_stamps;   // [blocks], uncached like part_out
  int *    dec_err;
  Species  sp;
};

// a00_rndu of bpp_amd_host.h, callable on the device (same integer recurrence, same conversion)
__host__ __device__ inline double rndu(a00_rng_t * r)
{
  *r = *r*6364136223846793005ULL + 1442695040888963407ULL;
  return (double)((*r >> 11) + 0.5)*(1.0/9007199254740992.0);
}
// a00_reflect of bpp_amd_host.h
__host__ __device__ inline double reflect(double x, double a, double b)
{
  const double w = b - a;
  if (!(w > 0)) return a;
  if (x >= a && x <= b) return x;
  double e = x < a ? a - x : x - b;
  const long n = (long)(e/w);
  e -= (double)n*w;
  if ((x < a) == ((n & 1) == 0)) return a + e;
  return b - e;
}

// ---- the leader lane's copy of its locus's tree: REGISTERS ------------------------------------------------------------
// A proposal is a chain of a few hundred dependent look-ups in arrays of at most 15 small integers (children, parent,
// buffer indices, populations).  Out of LDS every one of them is a round trip of ~100 cycles for the ONE wave a SIMD
// runs here — the sweep was waiting on the LDS 70 % of its time.  Packed one byte per node into 2 (<= 4 tips) or 4
// (<= 8 tips) registers per array, a look-up is one v_perm_b32 (a byte select with a run-time selector), an update a
// shift and a v_bfi_b32: the tree lives in the leader's registers for the whole launch, rolls back from a register
// copy, and only what the OTHER lanes of the locus read (ages, parents, buffer indices) is mirrored to LDS — stores,
// nobody waits for them.
template <int W> struct ByteArr
{
  uint32_t w[W];
  __device__ __forceinline__ int get(int i) const
  {
    uint32_t v = __builtin_amdgcn_perm(w[W > 1 ? 1 : 0], w[0], (uint32_t)i);               // selector 0-3: second operand, 4-7: first
#pragma unroll
    for (int q = 1; q < W/2; ++q)                                                            // entries 8q .. 8q+7 (a selector out of range yields a byte nobody keeps)
    {
      const uint32_t h = __builtin_amdgcn_perm(w[2*q + 1], w[2*q], (uint32_t)i - 8u*(uint32_t)q);
      v = (i >> 3) == q ? h : v;
    }
    return (int)(int8_t)v;
  }
  __device__ __forceinline__ void set(int i, int v)
  {
    const uint32_t sh = ((uint32_t)i & 3u) << 3, m = 0xffu << sh, val = ((uint32_t)v & 0xffu) << sh, wi = (uint32_t)i >> 2;
#pragma unroll
    for (int k = 0; k < W; ++k) w[k] = wi == (uint32_t)k ? ((w[k] & ~m) | val) : w[k];
  }
  struct Ref
  {
    ByteArr & a; int i;
    __device__ __forceinline__ operator int() const { return a.get(i); }
    __device__ __forceinline__ Ref & operator=(int v) { a.set(i, v); return *this; }
    __device__ __forceinline__ Ref & operator=(const Ref & o) { a.set(i, (int)o); return *this; }
  };
  __device__ __forceinline__ int operator[](int i) const { return get(i); }
  __device__ __forceinline__ Ref operator[](int i) { return Ref{*this, i}; }
  __device__ __forceinline__ void load(const int8_t * lds) { for (int k = 0; k < W; ++k) w[k] = reinterpret_cast<const uint32_t *>(lds)[k]; }
  __device__ __forceinline__ void store(int8_t * lds) const { for (int k = 0; k < W; ++k) reinterpret_cast<uint32_t *>(lds)[k] = w[k]; }
};

template <int NT> struct LTree
{
  static constexpr int W = NT <= 4 ? 2 : NT <= 8 ? 4 : 8;
  ByteArr<W> left, right, parent, clv, pmat, pop;
  double *   time;                                // the ages stay in LDS (S.tr.time): doubles, read a few times per proposal
  a00_rng_t  rng;
  int32_t    root, tips;
  template <class TR> __device__ __forceinline__ void load(TR & t)
  {
    left.load(t.left); right.load(t.right); parent.load(t.parent); clv.load(t.clv); pmat.load(t.pmat); pop.load(t.pop);
    time = t.time; rng = t.rng; root = t.root; tips = t.tips;
  }
  // what the other lanes read, and what goes back to HBM at the end
  template <class TR> __device__ __forceinline__ void mirror(TR & t) const
  {
    left.store(t.left); right.store(t.right); parent.store(t.parent); clv.store(t.clv); pmat.store(t.pmat); pop.store(t.pop);
    t.root = root;
  }
};
// the integer part of a tree, for the roll-back of a rejected proposal
template <int NT> struct LUndo { ByteArr<LTree<NT>::W> left, right, parent, clv, pmat, pop; int32_t root; };
template <int NT> __device__ __forceinline__ void save(LUndo<NT> & u, const LTree<NT> & t)
{ u.left = t.left; u.right = t.right; u.parent = t.parent; u.clv = t.clv; u.pmat = t.pmat; u.pop = t.pop; u.root = t.root; }
template <int NT> __device__ __forceinline__ void restore(LTree<NT> & t, const LUndo<NT> & u)
{ t.left = u.left; t.right = u.right; t.parent = u.parent; t.clv = u.clv; t.pmat = u.pmat; t.pop = u.pop; t.root = u.root; }

// the species tree's topology (uniform over the launch) and the leader's per-population counts, the same way
struct LSpecies { ByteArr<4> parent, left, right; int32_t S, npop; };
struct LCounts  { ByteArr<4> nin, nc, nin_new, nc_new, gl; };

template <class TT> __device__ __forceinline__ void swap_clv(TT & t, int i)
{
  const int inner = t.tips - 1, c = t.clv[i] + inner;                     // the other of the node's two buffers
  t.clv[i] = c >= t.tips + 2*inner ? c - 2*inner : c;
}
template <class TT> __device__ __forceinline__ void swap_pmat(TT & t, int i)
{
  const int edges = 2*t.tips - 2, c = t.pmat[i] + edges;
  t.pmat[i] = c >= 2*edges ? c - 2*edges : c;
}
// node sets are 16-bit masks (n <= 15): no private arrays, nothing spills to scratch
template <class TT> __device__ __forceinline__ uint32_t path_mask(const TT & t, int v)
{
  uint32_t m = 0;
  for (; v >= 0; v = t.parent[v]) m |= 1u << v;
  return m;
}
// the nodes of the subtree below (and including) v
template <class TT> __device__ __forceinline__ uint32_t subtree_mask(const TT & t, int v)
{
  uint32_t m = 1u << v;
  for (;;)
  {
    uint32_t add = 0;
    for (uint32_t q = m; q; q &= q - 1) { const int x = __ffs(q) - 1; const int l = t.left[x]; if (l >= 0) add |= (1u << l) | (1u << (int)t.right[x]); }
    if (!(add & ~m)) return m;
    m |= add;
  }
}
// exchange the tree positions of node ids a and b, in place (buffer indices stay with the ids)
template <int NT> __device__ void swap_ids(LTree<NT> & t, int a, int b)
{
#pragma unroll
  for (int i = 0; i < 2*NT - 1; ++i)                 // (entries past the tree are -1: they match neither id)
  {
    const int l = t.left[i], r = t.right[i], p = t.parent[i];
    t.left[i]   = l == a ? b : l == b ? a : l;
    t.right[i]  = r == a ? b : r == b ? a : r;
    t.parent[i] = p == a ? b : p == b ? a : p;
  }
  const int l = t.left[a], r = t.right[a], p = t.parent[a], q = t.pop[a]; const double tm = t.time[a];
  t.left[a] = t.left[b]; t.right[a] = t.right[b]; t.parent[a] = t.parent[b]; t.time[a] = t.time[b]; t.pop[a] = t.pop[b];
  t.left[b] = l; t.right[b] = r; t.parent[b] = p; t.time[b] = tm; t.pop[b] = q;
  t.root = t.root == a ? b : t.root == b ? a : t.root;
}

// ---- species tree helpers (lca_pop / climb of a00_driver.c); tau / anc = this workgroup's LDS copies
__device__ __forceinline__ int lca_pop(const LSpecies & sp, const uint16_t * anc, int p, int q)
{
  const uint32_t aq = anc[q];
  while (!((aq >> p) & 1u)) p = sp.parent[p];
  return p;
}
__device__ __forceinline__ int climb(const LSpecies & sp, const double * tau, int p, double t)
{
  for (;;)
  {
    const int pp = sp.parent[p];
    if (pp < 0 || !(tau[pp] <= t)) return p;
    p = pp;
  }
}

// MSC density of the tree in S.tr: tree_logpr of a00_driver.c = gtree_logprob (gtree.c:3957), the
// populations in order, each one's coalescent times visited in ascending order (selection instead
// of a sort buffer: same intervals, same order of additions as a00_msc_contrib).  Only the
// populations in `mask` are recomputed (into the *_new fields); the others keep their term — the
// terms are pure functions of the tree, so the sum equals the host's from-scratch one bit for bit.
// Three parts: the leader lane counts (density_prepare), the lanes of the locus take one population
// each for the floating-point part (lanes_density: the wave runs ONE term's instructions instead of the
// union of every leader's population loop), the leader adds the terms up in population order (density_sum).
template <int NT, class ST>                           // ST: what a proposal leaves for its evaluation (TaskLDS here, gsm::GState in gsampler.hpp)
__device__ void density_prepare(ST & S, const LTree<NT> & t, LCounts & cn, const LSpecies & sp, uint32_t mask)
{
  constexpr int NN = 2*NT - 1;
  const int n = 2*t.tips - 1;
  S.chain = mask;
  for (uint32_t m = mask; m; m &= m - 1)             // ascending = children before parents
  {
    const int p = __ffs(m) - 1;
    int nin;
    if (p >= sp.S)
    {
      const int l = sp.left[p], r = sp.right[p];
      const int nl = ((mask >> l) & 1u) ? cn.nin_new[l] - cn.nc_new[l] : cn.nin[l] - cn.nc[l];
      const int nr = ((mask >> r) & 1u) ? cn.nin_new[r] - cn.nc_new[r] : cn.nin[r] - cn.nc[r];
      nin = nl + nr;
    }
    else nin = cn.nin[p];                         // gene tips of the species: fixed
    uint32_t nodes = 0;
#pragma unroll
    for (int k = 0; k < NN; ++k) if (k >= t.tips && k < n && t.pop[k] == p) nodes |= 1u << k;
    const int nc = __popc(nodes);
    cn.nin_new[p] = nin; cn.nc_new[p] = nc;
    S.nin_new[p] = (int8_t)nin; S.nc_new[p] = (int8_t)nc; S.pnodes[p] = (decltype(S.pnodes[p] + 0u))nodes;
  }
}
// the term of population p (any lane of the locus)
__device__ void density_term(TaskLDS & S, const Species & sp, const double * tau, int p, double * t2h_out, uint32_t t2h_stride)
{
  const Tree & t = S.tr;
  uint32_t nodes = S.pnodes[p];
  const int ncoal = S.nc_new[p], nin = S.nin_new[p];
  const double ptau = sp.parent[p] >= 0 ? tau[sp.parent[p]] : -1.0;
  int steps = ncoal + (ptau >= 0 ? 1 : 0);
  if (nin == steps) --steps;
  double T2h = 0, prev = tau[p];
  int nn = nin;
  for (int k = 0; k < steps; ++k, --nn)
  {
    double tk = ptau;
    if (k < ncoal)
    {
      int best = -1;
      for (uint32_t m = nodes; m; m &= m - 1) { const int x = __ffs(m) - 1; if (best < 0 || t.time[x] < t.time[best]) best = x; }
      tk = t.time[best]; nodes &= ~(1u << best);
    }
    T2h += nn*(nn - 1)*(tk - prev);
    prev = tk;
  }
  double c = 0;
  if (ncoal) c += ncoal*tau[2*MAXPOP + p];
  if (T2h) c -= T2h/(tau[MAXPOP + p]*1.0);
  S.contrib_new[p] = c;
  if (t2h_out) t2h_out[(size_t)p*t2h_stride] = T2h;
}
__device__ __forceinline__ int nth_bit(uint32_t m, int j)
{
  for (int i = 0; i < j; ++i) m &= m - 1;
  return __ffs(m) - 1;
}
// lane n of the locus's np lanes: the terms of the n-th, (n+np)-th ... population of the mask
__device__ __forceinline__ void lanes_density(TaskLDS & S, const Species & sp, const double * tau, uint32_t n, uint32_t np,
                                              double * t2h_out = nullptr, uint32_t t2h_stride = 0)
{
  const uint32_t mask = S.chain;
  const int cnt = __popc(mask);
  for (int j = (int)n; j < cnt; j += (int)np) density_term(S, sp, tau, nth_bit(mask, j), t2h_out, t2h_stride);
}
__device__ __forceinline__ double density_sum(const TaskLDS & S, int npop)
{
  double logpr = 0;
  for (int p = 0; p < npop; ++p) logpr += ((S.chain >> p) & 1u) ? S.contrib_new[p] : S.contrib[p];
  return logpr;
}
// the proposal stands: its terms become the current ones
__device__ __forceinline__ void commit_logpr(TaskLDS & S, LCounts & cn)
{
  for (uint32_t m = S.chain; m; m &= m - 1)
  {
    const int p = __ffs(m) - 1;
    S.contrib[p] = S.contrib_new[p]; cn.nin[p] = (int)cn.nin_new[p]; cn.nc[p] = (int)cn.nc_new[p];
  }
}
// populations on the path between two populations one of which is an ancestor (or self) of the other
__device__ __forceinline__ uint32_t pop_chain(const uint16_t * anc, int a, int b)
{
  const uint32_t aa = anc[a], ab = anc[b];
  const bool a_lower = (aa >> b) & 1u;
  const uint32_t lo = a_lower ? aa : ab, hi = a_lower ? ab : aa;
  const int higher = a_lower ? b : a;
  return lo & ~(hi & ~(1u << higher));
}

// install a proposal: toggle buffers, node-update list of the nodes in mask ndm in children-first order
// (= by age) — step_add of a00_driver.c; the fresh (a,b) of the changed branches (mask brm) are left to the
// lanes of the locus (lanes_branches)
template <int NT, class ST>
__device__ void install(ST & S, LTree<NT> & t, uint32_t brm, uint32_t ndm)
{
  S.brm = brm;
  for (; brm; brm &= brm - 1) swap_pmat(t, __ffs(brm) - 1);
  // the ages of the inner nodes, all at once (independent LDS reads): the youngest remaining node comes next
  // (a parent is always older than its children)
  double tk[NT - 1];
#pragma unroll
  for (int j = 0; j < NT - 1; ++j) tk[j] = t.time[t.tips + j < 2*NT ? t.tips + j : 0];
  int nn = 0;
  uint32_t wclv = S.wclv;
  while (ndm)
  {
    int best = -1; double tb = 0;
#pragma unroll
    for (int j = 0; j < NT - 1; ++j)
    {
      const int x = t.tips + j;
      if (((ndm >> x) & 1u) && (best < 0 || tk[j] < tb)) { best = x; tb = tk[j]; }
    }
    ndm &= ~(1u << best);
    swap_clv(t, best);
    const int l = t.left[best], r = t.right[best];
    const int pc = t.clv[best];
    // children first: a child of this node that is itself recomputed was toggled in an earlier round
    S.ops[nn] = make_op(pc, t.clv[l], t.pmat[l], t.clv[r], t.pmat[r]);
    wclv |= 1u << (pc - t.tips);
    ++nn;
  }
  S.nops = nn; S.wclv = wclv;
}

// lane n of the locus's np lanes: (a,b) of the n-th, (n+np)-th ... changed branch, into its (already toggled) buffer
__device__ __forceinline__ void lanes_branches(TaskLDS & S, double rate, uint32_t n, uint32_t np)
{
  const Tree & t = S.tr;
  const uint32_t brm = S.brm;
  const int cnt = __popc(brm);
  for (int j = (int)n; j < cnt; j += (int)np)
  {
    const int x = nth_bit(brm, j);
    const double len = (t.time[t.parent[x]] - t.time[x])*1.0;                // rate_mui = 1 (locus.c:2350)
    double A, B;
    jc69_ab(len, rate, A, B);
    S.ab[t.pmat[x]][0] = A; S.ab[t.pmat[x]][1] = B;
  }
}

// what a rejected per-locus proposal puts back besides the integer arrays: the ages it moved (at most three nodes)
struct TimeUndo { double t0, ta, tb; int32_t n0, na, nb; };
// section cycles of ONE leader lane (BPA_SMP_DBG & 16: lane 0 of workgroup 0)
struct Prof
{
  long long t, acc[12]; bool on;
  __device__ __forceinline__ void tick(int i) { if (on) { const long long t1 = clock64(); acc[i] += t1 - t; t = t1; } }
};

// GAGE on the k-th inner node (gage_step of a00_driver.c; propose_ages, gtree.c:4585)
template <int NT, class ST>
__device__ bool propose_gage(ST & S, LTree<NT> & t, LCounts & cn, TimeUndo & tu, double & hast, int k, const LSpecies & sp,
                             const Species & spl, const double * tau, Prof & pf)
{
  if (pf.on) pf.t = clock64();
  const int n = 2*t.tips - 1;
  const int v = t.tips + k;                          // the k-th inner node: tips are nodes 0..tips-1 (bpa_sampler_set_tree checks)
  if (v >= n) return false;
  const double u = rndu(&t.rng);
  const int l = t.left[v], r = t.right[v], p = t.parent[v];
  const double tl = t.time[l], tr = t.time[r], told = t.time[v], tpar = t.time[p < 0 ? 0 : p];
  const int pl = t.pop[l], pr = t.pop[r];
  double lo = fmax(tl, tr);
  if (pl != pr) lo = fmax(lo, tau[lca_pop(sp, spl.anc, pl, pr)]);
  const double hi = p >= 0 ? tpar : 999.0;
  if (!(hi > lo)) { (void)rndu(&t.rng); return false; }
  const double tnew = reflect(told + spl.ft_gage*(u - 0.5), lo, hi);
  const int oldpop = t.pop[v];
  tu.n0 = v; tu.t0 = told; tu.na = -1;
  t.time[v] = tnew;
  const int newpop = climb(sp, tau, pl, tnew);
  t.pop[v] = newpop;
  hast = 0;
  pf.tick(8);
  density_prepare<NT>(S, t, cn, sp, pop_chain(spl.anc, oldpop, newpop));
  pf.tick(9);
  uint32_t brm = (1u << l) | (1u << r);
  if (p >= 0) brm |= 1u << v;
  install<NT>(S, t, brm, path_mask(t, v));
  pf.tick(10);
  return true;
}

// GSPR on the k-th non-root node (gspr_step of a00_driver.c; propose_spr, gtree.c:6531)
// log(targets/sources) for every pair of counts, once per sampler: a GSPR proposal reads its Hastings ratio instead of
// spending ~2.5 k cycles of the leader lane on a division and a logarithm
__global__ void lograt_kernel(double * tab)
{
  const int i = threadIdx.x / MAXN, j = threadIdx.x % MAXN;
  tab[threadIdx.x] = (i && j) ? log((double)i/(double)j) : 0.0;
}

template <int NT, class ST>
__device__ bool propose_gspr(ST & S, LTree<NT> & t, LCounts & cn, TimeUndo & tu, double & hast, int k, const LSpecies & sp,
                             const Species & spl, const double * tau, const double * lograt, Prof & pf)
{
  if (pf.on) pf.t = clock64();
  const int n = 2*t.tips - 1;
  const int a = k < t.root ? k : k + 1;              // the k-th node that is not the root
  if (a >= n) return false;
  const double u1 = rndu(&t.rng), u2 = rndu(&t.rng);
  const int root_before = t.root;
  const int p = t.parent[a], lp = t.left[p], s = lp == a ? (int)t.right[p] : lp, g = t.parent[p];
  pf.tick(0);
  // youngest population from a's upwards that holds gene tips outside a's subtree
  const int leaves = __popc(subtree_mask(t, a) & ((1u << t.tips) - 1u));
  int pop0 = t.pop[a];
  const int popa = pop0;
  while (cn.gl[pop0] <= leaves && sp.parent[pop0] >= 0) pop0 = sp.parent[pop0];
  pf.tick(1);
  const double ta = t.time[a], tpo = t.time[p], troot = t.time[root_before];       // (independent LDS reads)
  const double lo = fmax(ta, tau[pop0]);
  const double tnew = reflect(tpo + spl.ft_gspr*(u1 - 0.5), lo, 999.0);
  const int popt = climb(sp, tau, popa, tnew);
  // targets (bit j = branch above node j; the father's own branch stands for the sibling's) and sources, in ONE
  // fully unrolled scan: the LDS reads of all nodes' ages are independent and go out together
  pf.tick(2);
  uint32_t tmask = 0; int nsrc = 1;
  {
    const int pp = t.pop[p];
    const bool above_root = tnew >= troot, src_on = p != root_before;
#pragma unroll
    for (int j = 0; j < 2*NT - 1; ++j)
    {
      const int pj = t.parent[j];
      const double tj = t.time[j], tpj = t.time[pj < 0 ? 0 : pj];
      const uint32_t aj = spl.anc[t.pop[j] & (MAXPOP - 1)];
      const bool in = j < n && j != a && j != root_before;
      if (in && !above_root && tj <= tnew && tpj > tnew && ((aj >> popt) & 1u)) tmask |= 1u << j;
      if (in && src_on && j != s && j != p && tj <= tpo && tpj > tpo && ((aj >> pp) & 1u)) ++nsrc;
    }
    if (above_root) tmask = 1u << root_before;
  }
  pf.tick(3);
  const int ntg = __popc(tmask);
  if (!ntg) { (void)rndu(&t.rng); return false; }
  int pick = (int)(u2*ntg);
  if (pick == ntg) pick = 0;                          // (int)(u2*ntg) % ntg of the host driver
  int tgt = nth_bit(tmask, pick);
  if (tgt == p) tgt = s;
  // prune: the sibling takes p's place; regraft p (with a below it) above tgt at tnew in popt
  t.parent[s] = g;
  if (g >= 0) { if (t.left[g] == p) t.left[g] = s; else t.right[g] = s; } else t.root = s;
  const int pc = t.parent[tgt];
  const uint32_t chain = pop_chain(spl.anc, t.pop[p], popt);
  tu.n0 = p; tu.t0 = tpo; tu.na = -1;
  t.time[p] = tnew; t.pop[p] = popt;
  t.left[p] = a; t.right[p] = tgt; t.parent[a] = p; t.parent[tgt] = p;
  t.parent[p] = pc;
  if (pc >= 0) { if (t.left[pc] == tgt) t.left[pc] = p; else t.right[pc] = p; } else t.root = p;
  pf.tick(4);
  uint32_t ndm = path_mask(t, p);
  if (g >= 0) ndm |= path_mask(t, g);
  uint32_t bset = (1u << a) | (1u << tgt) | (1u << p) | (1u << s);
  if (t.root != root_before)
  {
    // the root node object keeps its identity (gtree.c:6129-6175): rename the two ids in the sets
    const int newtop = t.root;
    tu.na = newtop; tu.nb = root_before; tu.ta = t.time[newtop]; tu.tb = t.time[root_before];
    swap_ids<NT>(t, newtop, root_before);
    const uint32_t bn = 1u << newtop, br_ = 1u << root_before;
    auto ren = [&](uint32_t m) { const uint32_t hn = m & bn, hr = m & br_; m &= ~(bn | br_); if (hn) m |= br_; if (hr) m |= bn; return m; };
    ndm = ren(ndm) | path_mask(t, newtop);
    bset = ren(bset);
  }
  uint32_t brm = 0;
  for (uint32_t m = bset; m; m &= m - 1) { const int x = __ffs(m) - 1; if (t.parent[x] >= 0) brm |= 1u << x; }
  hast = lograt[ntg*(2*NT) + nsrc];
  pf.tick(5);
  density_prepare<NT>(S, t, cn, sp, chain);
  pf.tick(6);
  install<NT>(S, t, brm, ndm);
  pf.tick(7);
  return true;
}

// a population's term of the MSC density from its sufficient statistics (a00_msc_term of bpp_amd_host.h)
__device__ __forceinline__ double msc_term(int ncoal, double T2h, double theta, double l2t)
{
  double c = 0;
  if (ncoal) c += ncoal*l2t;
  if (T2h) c -= T2h/(theta*1.0);
  return c;
}

__device__ void decide(double lnacc, double u, uint32_t epoch, uint32_t * flag, uint32_t * counters, double * taus,
                       const Species & sp, int tau_q, int theta_p, double win_u, double mix_c, double mix_lnc);

template<int NT>
__global__ void __launch_bounds__(BS) sweep_kernel(const Args A)
{
  __shared__ TaskLDS s_task[TPB];
  // CLV buffers of the workgroup's 64 lanes: 2*(tips-1) of them, sized at launch — with the fixed 8-tip size a
  // workgroup needed 52 KB of LDS, three fitted a CU, and config 2's 839 workgroups ran in two rounds on 768 slots
  extern __shared__ __attribute__((aligned(16))) double s_clv_raw[];
  double (*s_clv)[BS][4] = reinterpret_cast<double (*)[BS][4]>(s_clv_raw);
  __shared__ double  s_term[BS];
  __shared__ double  s_tau[3*MAXPOP];                    // tau | theta | log(2/theta) of this launch's (proposed) species tree
  const long long ph_start = clock64();
  const uint32_t b = blockIdx.x, lane = threadIdx.x, gl = b*BS + lane;
  const uint32_t t0 = A.blk_task_off[b], ntask = A.blk_task_off[b+1] - t0;
  const LaneRec lrec = A.lane_rec[gl];
  const uint32_t task = lrec.task;
  const bool active = task != 0xffffffffu;
  const uint32_t ts = active ? task - t0 : 0u;
  const bool leader = active && (lrec.n_np_tips & 255u) == 0;
  const bool restore_mix = A.epoch != 0 && *A.mix_flag == A.epoch;      // epoch 0: nothing pending
  // the species tree is indexed with run-time population numbers by every lane (density terms) and by the leaders
  // (ancestor sets): out of LDS, not out of the kernel-argument segment (a dynamic index into a by-value argument is
  // a global load each time); its topology also sits in the leaders' registers (LSpecies)
  __shared__ Species s_sp;
  {
    const uint32_t * src = reinterpret_cast<const uint32_t *>(&A.sp);
    uint32_t * dst = reinterpret_cast<uint32_t *>(&s_sp);
    for (uint32_t i = lane; i < sizeof(Species)/4; i += BS) dst[i] = src[i];
  }
  const Species & spl = s_sp;
  LSpecies sp;
  {
    const uint32_t * q = reinterpret_cast<const uint32_t *>(A.sp.parent);
    for (int k = 0; k < 4; ++k) sp.parent.w[k] = q[k];
    q = reinterpret_cast<const uint32_t *>(A.sp.left);
    for (int k = 0; k < 4; ++k) sp.left.w[k] = q[k];
    q = reinterpret_cast<const uint32_t *>(A.sp.right);
    for (int k = 0; k < 4; ++k) sp.right.w[k] = q[k];
    sp.S = A.sp.S; sp.npop = A.sp.npop;
  }
  const int npop = sp.npop;

  // ---- load: tree (or its pre-step snapshot), (a,b) table, this lane's CLV buffers and constants
  uint32_t np = 0, tips = 0, n = 0, tipcodes = 0, wgt = 0;
  double f0 = 0, f1 = 0, f2 = 0, f3 = 0, rw = 0, rate = 1;
  double * g_clv = nullptr, * g_pmat = nullptr;
  if (lane < (uint32_t)(3*MAXPOP)) s_tau[lane] = A.taus[lane];
  __shared__ double s_lograt[(2*NT)*(2*NT)];
  if (A.mode == 0) for (uint32_t i = lane; i < (uint32_t)((2*NT)*(2*NT)); i += BS) s_lograt[i] = A.lograt[(i/(2*NT))*MAXN + i % (2*NT)];
  LCounts cn;
  for (int k = 0; k < 4; ++k) cn.nin.w[k] = cn.nc.w[k] = cn.nin_new.w[k] = cn.nc_new.w[k] = cn.gl.w[k] = 0u;
  // an all-loci step that recomputes every inner node (mixing, start-up) reads no inner CLV: nothing to load
  const bool load_clv = A.mode == 0 || A.mode == 4;
  if (active)
  {
    const TaskRec & R = A.task_rec[task];
    n = lrec.n_np_tips & 255u; np = (lrec.n_np_tips >> 8) & 255u; tips = lrec.n_np_tips >> 16;
    wgt = lrec.wgt; tipcodes = lrec.tipcodes;
    g_clv = lrec.clv; g_pmat = lrec.pmat;
    // the (a,b) table of the locus: its lanes share the copy
    for (uint32_t i = n; i < 4*(2*tips - 2); i += np) (&s_task[ts].ab[0][0])[i] = g_pmat[i];
    const uint32_t nbuf = load_clv ? 2*(tips - 1) : 0u;
    for (uint32_t c = 0; c < nbuf; ++c)
    {
      const double2 * p = reinterpret_cast<const double2 *>(g_clv + (size_t)c*np*4);
      const double2 u = p[0], w = p[1];
      s_clv[c][lane][0] = u.x; s_clv[c][lane][1] = u.y; s_clv[c][lane][2] = w.x; s_clv[c][lane][3] = w.y;
    }
    rate = R.rate; rw = R.rw; f0 = R.f0; f1 = R.f1; f2 = R.f2; f3 = R.f3;
    if (n == 0)
    {
      const uint4 g4 = *reinterpret_cast<const uint4 *>(R.gl), n4 = *reinterpret_cast<const uint4 *>(R.nin);
      cn.gl.w[0] = g4.x; cn.gl.w[1] = g4.y; cn.gl.w[2] = g4.z; cn.gl.w[3] = g4.w;
      cn.nin.w[0] = n4.x; cn.nin.w[1] = n4.y; cn.nin.w[2] = n4.z; cn.nin.w[3] = n4.w;
    }
  }
  // trees of this workgroup's loci: all lanes copy, 16 B at a time
  {
    constexpr uint32_t U = sizeof(Tree)/16;
    const Tree * src = (restore_mix ? A.snap : A.trees) + t0;
    for (uint32_t i = lane; i < ntask*U; i += BS)
      reinterpret_cast<uint4 *>(&s_task[i/U].tr)[i % U] = reinterpret_cast<const uint4 *>(src + i/U)[i % U];
  }
  __syncthreads();
  if (A.refresh_logpr)
  {
    // THETA moved thetas after these densities were stored: every tree's density again, from its statistics — one
    // population's term per lane, added up in population order (what theta_step_all of a00_driver.c recomputes)
    if (active)
      for (int p = (int)n; p < npop; p += (int)np)
        s_task[ts].contrib_new[p] = msc_term(A.pop_nc[(size_t)p*A.ntasks + task], A.pop_t2h[(size_t)p*A.ntasks + task], s_tau[MAXPOP + p], s_tau[2*MAXPOP + p]);
    __syncthreads();
    if (leader)
    {
      double lp = 0;
      for (int p = 0; p < npop; ++p) lp += s_task[ts].contrib_new[p];
      s_task[ts].tr.logpr = lp;
    }
    __syncthreads();
  }
  // the proposed species tree of an all-loci step is this workgroup's copy of the taus
  double lminf = 0, lmaxf = 0, tq_old = 0, tq_lo = 0, tq_hi = 0, minf = 1, maxf = 1;
  if (A.mode == 4)
  {
    const int q = (int)A.tau_q, pq = spl.parent[q];
    tq_old = s_tau[q]; tq_lo = fmax(s_tau[spl.left[q]], s_tau[spl.right[q]]); tq_hi = pq >= 0 ? s_tau[pq] : 999.0;
    const double tnew = reflect(tq_old + spl.ft_tau*(A.tau_u - 0.5), tq_lo, tq_hi);
    minf = (tnew - tq_lo)/(tq_old - tq_lo); maxf = (tnew - tq_hi)/(tq_old - tq_hi);
    lminf = log(minf); lmaxf = log(maxf);
    __syncthreads();
    if (lane == 0) s_tau[q] = tnew;
  }
  else if (A.mode == 1)
  {
    __syncthreads();
    if (lane < (uint32_t)npop) s_tau[lane] *= A.mix_c;
  }
  // ---- the leader takes its tree into registers
  LTree<NT> T;
  LUndo<NT> U;
  TimeUndo tu{0, 0, 0, -1, -1, -1};
  double lnl_cur = 0, logpr_cur = 0, hast = 0, hast2 = 0;
  uint32_t nprop_done = 0, nacc = 0, w_nupd = 0, w_nbr = 0;
  if (leader)
  {
    TaskLDS & S = s_task[ts];
    if (restore_mix) { S.tr.rng = A.trees[task].rng; S.tr.proposals = A.trees[task].proposals; S.tr.accepted = A.trees[task].accepted;
                       S.tr.sw_nupd = A.trees[task].sw_nupd; S.tr.sw_nbr = A.trees[task].sw_nbr; }
    T.load(S.tr);
    lnl_cur = S.tr.lnl; logpr_cur = S.tr.logpr;
    if (A.mode == 0) density_prepare<NT>(S, T, cn, sp, (1u << npop) - 1u);      // the current terms: counts here, terms by the lanes below
    S.nops = 0; S.active = 0; S.wclv = 0; S.brm = 0;
  }
  else { T.time = nullptr; T.rng = 0; T.root = 0; T.tips = 0; }
  __syncthreads();
  if (A.mode == 0)
  {
    if (active) lanes_density(s_task[ts], spl, s_tau, n, np);
    __syncthreads();
    if (leader) commit_logpr(s_task[ts], cn);
    __syncthreads();
  }

  const uint32_t nprop = A.mode == 0 ? A.nsteps_gage + A.nsteps_gspr : (A.mode == 2 ? 0u : 1u);
  // phase timing of workgroup 0 (BPA_SMP_DBG & 16): proposal | lanes' share | node updates | decision
  const bool ph_on = (A.dbg & 16u) && b == 0 && lane == 0;
  Prof pf; pf.on = ph_on; pf.t = 0; for (int i = 0; i < 12; ++i) pf.acc[i] = 0;
  long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_t = ph_on ? clock64() : 0;
  if (ph_on) A.mix_delta[14] = (double)(ph_t - ph_start);
#define SMP_PHASE(i_) do { if (ph_on) { const long long t1_ = clock64(); ph[i_] += t1_ - ph_t; ph_t = t1_; } } while (0)
  for (uint32_t step = 0; step < nprop; ++step)
  {
    // ---- phase 1: the locus's leader lane proposes (registers), then mirrors the tree for the other lanes
    if (leader)
    {
      TaskLDS & S = s_task[ts];
      bool ok; int act = 1;
      if (A.mode == 0) save<NT>(U, T);
      if (A.mode == 0 && (A.dbg & 4u)) ok = false;
      else if (A.mode == 0)
        ok = step < A.nsteps_gage ? propose_gage<NT>(S, T, cn, tu, hast, (int)step, sp, spl, s_tau, pf)
                                  : propose_gspr<NT>(S, T, cn, tu, hast, (int)(step - A.nsteps_gage), sp, spl, s_tau, s_lograt, pf);
      else if (A.mode == 4)
      {
        // TAU q (tau_step of a00_driver.c): the gene nodes of q and its children between the bounds move
        const int nn_ = 2*T.tips - 1, q = (int)A.tau_q, cl = sp.left[q], cr = sp.right[q];
        A.snap[task] = S.tr;
        uint32_t brm = 0, ndm = 0; int above = 0, below = 0;
        for (int k = T.tips; k < nn_; ++k)
        {
          const int pk = T.pop[k]; const double tk = T.time[k];
          if ((pk != q && pk != cl && pk != cr) || tk < tq_lo || tk > tq_hi) continue;
          if (tk >= tq_old) { T.time[k] = tq_hi + maxf*(tk - tq_hi); ++above; } else { T.time[k] = tq_lo + minf*(tk - tq_lo); ++below; }
          brm |= (1u << (int)T.left[k]) | (1u << (int)T.right[k]);
          if (T.parent[k] >= 0) brm |= 1u << k;
          ndm |= path_mask(T, k);
        }
        density_prepare<NT>(S, T, cn, sp, (1u << npop) - 1u);
        hast = below*lminf; hast2 = above*lmaxf;       // p_delta of the host driver = (density difference + hast) + hast2, below
        ok = true;
        if (ndm) install<NT>(S, T, brm, ndm);
        else { S.brm = 0; S.nops = 0; act = 3; }         // no gene node moves here: only the density changes
      }
      else
      {
        // mixing (mix_step of a00_driver.c) or start-up: every branch, every inner node
        const int nn_ = 2*T.tips - 1;
        if (A.mode == 1) A.snap[task] = S.tr;
        uint32_t brm = 0, ndm = 0; int ninner = 0;
        for (int k = 0; k < nn_; ++k)
        {
          if (T.left[k] >= 0) { if (A.mode == 1) T.time[k] *= A.mix_c; ndm |= 1u << k; ++ninner; }
          if (T.parent[k] >= 0) brm |= 1u << k;
        }
        if (A.mode == 3)
        {
          for (uint32_t m = brm; m; m &= m - 1) swap_pmat(T, __ffs(m) - 1);       // start-up evaluates in place:
          for (uint32_t m = ndm; m; m &= m - 1) swap_clv(T, __ffs(m) - 1);        // toggle twice = no toggle
        }
        density_prepare<NT>(S, T, cn, sp, (1u << npop) - 1u);
        hast = (double)ninner*A.mix_lnc;
        install<NT>(S, T, brm, ndm);
        ok = true;
      }
      S.active = ok ? act : 0;
      if (!ok) S.nops = 0;
      else T.mirror(S.tr);
    }
    __syncthreads();
    SMP_PHASE(1);
    // ---- phase 1b: the lanes of the locus share the floating-point part of the proposal — one density term, one
    // branch's exponential each
    if (active && s_task[ts].active)
    {
      TaskLDS & S = s_task[ts];
      lanes_density(S, spl, s_tau, n, np);
      lanes_branches(S, rate, n, np);
    }
    __syncthreads();
    SMP_PHASE(2);
    // ---- phase 2: one lane per pattern runs the node updates out of LDS
    double term = 0;
    if (active && s_task[ts].active == 1 && !(A.dbg & 1u))
    {
      const TaskLDS & S = s_task[ts];
      const int nops = S.nops;
      double last[4] = {0, 0, 0, 0}; uint32_t last_c = 0xffffffffu;
      for (int o = 0; o < nops; ++o)
      {
        const Op opw = S.ops[o];
        const struct { uint32_t parent, lc, lp, rc, rp; } op = {(uint32_t)opw & 255u, (uint32_t)(opw >> 8) & 255u, (uint32_t)(opw >> 16) & 255u,
                                                                (uint32_t)(opw >> 24) & 255u, (uint32_t)(opw >> 32) & 255u};
        double lv[4], rv[4], x[4], y[4];
        if (op.lc < tips) expand_code((tipcodes >> (4*op.lc)) & 15u, lv);
        else if (op.lc == last_c) { lv[0] = last[0]; lv[1] = last[1]; lv[2] = last[2]; lv[3] = last[3]; }
        else { const double * c = s_clv[op.lc - tips][lane]; lv[0] = c[0]; lv[1] = c[1]; lv[2] = c[2]; lv[3] = c[3]; }
        if (op.rc < tips) expand_code((tipcodes >> (4*op.rc)) & 15u, rv);
        else if (op.rc == last_c) { rv[0] = last[0]; rv[1] = last[1]; rv[2] = last[2]; rv[3] = last[3]; }
        else { const double * c = s_clv[op.rc - tips][lane]; rv[0] = c[0]; rv[1] = c[1]; rv[2] = c[2]; rv[3] = c[3]; }
        matvec4_ab(S.ab[op.lp][0], S.ab[op.lp][1], lv, x);
        matvec4_ab(S.ab[op.rp][0], S.ab[op.rp][1], rv, y);
        last[0] = x[0]*y[0]; last[1] = x[1]*y[1]; last[2] = x[2]*y[2]; last[3] = x[3]*y[3]; last_c = op.parent;
        double * out = s_clv[op.parent - tips][lane];
        out[0] = last[0]; out[1] = last[1]; out[2] = last[2]; out[3] = last[3];
      }
      // the last update is the root's (children first, the root is the oldest node of every update list)
      const double tr_ = dot4_pair(f0, f1, f2, f3, last);
      term = log(0 + tr_*rw)*wgt;
    }
    s_term[lane] = term;
    __syncthreads();
    SMP_PHASE(3);
    // ---- phase 3: the leader sums in pattern order and decides
    if (leader && s_task[ts].active == 3)
    {
      // TAU without a moving gene node in this locus
      TaskLDS & S = s_task[ts];
      const double lp_new = density_sum(S, npop);
      const double dl = ((lp_new - logpr_cur) + hast) + hast2;
      A.mix_delta[task] = dl; S.hast = dl;
      logpr_cur = lp_new;
      commit_logpr(S, cn);
    }
    else if (leader && s_task[ts].active)
    {
      TaskLDS & S = s_task[ts];
      double lnl = 0;
      for (uint32_t q = 0; q < np; ++q) lnl += s_term[lane + q];
      lnl = A.bfbeta == 1.0 ? lnl : A.bfbeta == 0.0 ? 0.0 : A.bfbeta*lnl;
      const double lp_new = density_sum(S, npop);
      if (A.mode == 0)
      {
        w_nupd += (uint32_t)S.nops; w_nbr += (uint32_t)__popc(S.brm);
        const double lnacc = (lp_new - logpr_cur) + (lnl - lnl_cur) + hast;
        const double u = rndu(&T.rng);
        ++nprop_done;
        if (lnacc >= 0 || u < exp(lnacc)) { lnl_cur = lnl; logpr_cur = lp_new; ++nacc; commit_logpr(S, cn); }
        else
        {
          // rejected: topology, populations and buffer indices come back from the register copy, the moved ages from tu
          restore<NT>(T, U);
          if (tu.na >= 0) { T.time[tu.na] = tu.ta; T.time[tu.nb] = tu.tb; }
          T.time[tu.n0] = tu.t0;
          T.mirror(S.tr);
        }
      }
      else
      {
        const double dpr = lp_new - logpr_cur;
        const double h = A.mode == 4 ? (dpr + hast) + hast2 : dpr + hast;
        const double dl = A.mode == 3 ? 0.0 : (lnl - lnl_cur) + h;
        A.mix_delta[task] = dl; S.hast = dl;
        lnl_cur = lnl; logpr_cur = lp_new;
        commit_logpr(S, cn);
      }
    }
    __syncthreads();
    SMP_PHASE(4);
  }
  if (ph_on) for (int i = 0; i < 6; ++i) A.mix_delta[8 + i] = (double)ph[i];
  if (ph_on) for (int i = 0; i < 12; ++i) A.mix_delta[16 + i] = (double)pf.acc[i];
  const long long ph_store = clock64();
#undef SMP_PHASE

  // ---- an all-loci step leaves the sum of its loci's terms (task order): 839 values for the decision instead of 10 000
  if ((A.mode == 1 || A.mode == 4) && lane == 0)
  {
    double part = 0;
    for (uint32_t k = 0; k < ntask; ++k) part += s_task[k].hast;
    if (!A.dec_on) A.part_out[b] = part;
    else
    {
      __hip_atomic_store(A.part_out + b, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the sum is out before the stamp says so
      __hip_atomic_store(A.dec_stamps + b, (unsigned long long)A.dec_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- store
  if (leader)
  {
    // the scalars the leader kept in registers go back into the LDS tree that is copied out below
    TaskLDS & S = s_task[ts];
    S.tr.rng = T.rng; S.tr.lnl = lnl_cur; S.tr.logpr = logpr_cur;
    S.tr.proposals += nprop_done; S.tr.accepted += nacc; S.tr.sw_nupd += w_nupd; S.tr.sw_nbr += w_nbr;
  }
  if (A.mode == 0)
  {
    // the sufficient statistics of the final state, for the THETA kernels (not kept in LDS along the way: a workgroup
    // must stay under 40 KB)
    if (leader)
    {
      TaskLDS & S = s_task[ts];
      density_prepare<NT>(S, T, cn, sp, (1u << npop) - 1u);
      for (int p = 0; p < npop; ++p) A.pop_nc[(size_t)p*A.ntasks + task] = (int8_t)(int)cn.nc_new[p];
    }
    __syncthreads();
    if (active) lanes_density(s_task[ts], spl, s_tau, n, np, A.pop_t2h + task, A.ntasks);
  }
  __syncthreads();
  {
    constexpr uint32_t U4 = sizeof(Tree)/16;
    for (uint32_t i = lane; i < ntask*U4; i += BS)
      reinterpret_cast<uint4 *>(A.trees + t0 + i/U4)[i % U4] = reinterpret_cast<const uint4 *>(&s_task[i/U4].tr)[i % U4];
  }
  if (active && nprop)
  {
    for (uint32_t i = n; i < 4*(2*tips - 2); i += np) g_pmat[i] = (&s_task[ts].ab[0][0])[i];
    // only the CLV buffers a node update wrote go back
    const uint32_t wclv = s_task[ts].wclv;
    for (uint32_t m = wclv; m; m &= m - 1)
    {
      const uint32_t c = (uint32_t)__ffs(m) - 1u;
      double2 * p = reinterpret_cast<double2 *>(g_clv + (size_t)c*np*4);
      double2 u, w; u.x = s_clv[c][lane][0]; u.y = s_clv[c][lane][1]; w.x = s_clv[c][lane][2]; w.y = s_clv[c][lane][3];
      p[0] = u; p[1] = w;
    }
  }
  if (ph_on) A.mix_delta[15] = (double)(clock64() - ph_store);

  // ---- the decision, by the last workgroup of the grid (its own state is stored: it has nothing else to do)
  if (A.dec_on && (A.mode == 1 || A.mode == 4) && b == gridDim.x - 1)
  {
    const uint32_t nb = gridDim.x;
    const unsigned long long stamp = A.dec_epoch, t_wait = wall_clock64();
    bool timed_out = false;
    for (;;)
    {
      bool all = true;
      for (uint32_t i = lane; i < nb; i += BS)
        all = all && __hip_atomic_load(A.dec_stamps + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == stamp;
      if (__all(all)) break;
      if (wall_clock64() - t_wait > 20000000ull) { timed_out = true; break; }       // 0.2 s at 100 MHz
      __builtin_amdgcn_s_sleep(2);
    }
    double acc = 0;
    for (uint32_t i = lane; i < nb; i += BS) acc += __hip_atomic_load(A.part_out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    s_term[lane] = acc;
    __syncthreads();
    if (lane == 0)
    {
      double tot = 0;
      for (int q = 0; q < BS; ++q) tot += s_term[q];
      if (timed_out) { *A.dec_err = 1; tot = -1e300; }              // a lost stamp: reject (the trees roll back) and tell the host
      decide(tot, A.dec_uacc, A.dec_epoch, A.dec_flag, A.dec_counters, A.dec_taus, spl, A.mode == 4 ? (int)A.tau_q : -1, -1,
             A.tau_u, A.mix_c, A.mix_lnc);
    }
  }
}

// the single decision of an all-loci step (tau_step / mix_step of a00_driver.c; stree.c:6280,
// prop_mixing.c:203-205): flag := epoch when REJECTED; on acceptance the device-resident taus follow
__device__ void decide(double lnacc, double u, uint32_t epoch, uint32_t * flag,
                       uint32_t * counters, double * taus, const Species & sp, int tau_q, int theta_p, double win_u,
                       double mix_c, double mix_lnc)
{
  double tnew = 0;
  const int root = sp.npop - 1;
  bool valid = true;
  if (theta_p >= 0)
  {
    const double old = taus[MAXPOP + theta_p];
    tnew = reflect(old + sp.ft_theta*(win_u - 0.5), 0.0, 999.0);
    lnacc += (sp.theta_alpha - 1)*log(tnew/old) - sp.theta_beta*(tnew - old);
    valid = tnew > 0;
  }
  else if (tau_q >= 0)
  {
    const int pq = sp.parent[tau_q];
    const double old = taus[tau_q], lo = fmax(taus[sp.left[tau_q]], taus[sp.right[tau_q]]), hi = pq >= 0 ? taus[pq] : 999.0;
    tnew = reflect(old + sp.ft_tau*(win_u - 0.5), lo, hi);
    if (pq < 0 && sp.tau_alpha > 0) lnacc += (sp.tau_alpha - 1 - (sp.S - 1) + 1)*log(tnew/old) - sp.tau_beta*(tnew - old);
  }
  else
  {
    lnacc += (double)(sp.S - 1)*mix_lnc;
    if (sp.tau_alpha > 0)
      lnacc += (sp.tau_alpha - 1)*mix_lnc - sp.tau_beta*(taus[root]*mix_c - taus[root]) - (double)(sp.S - 2)*mix_lnc;
  }
  const bool accept = valid && (lnacc >= 0 || u < exp(lnacc));
  counters[0] += 1; counters[1] += accept ? 1u : 0u;
  if (!accept) { *flag = epoch; return; }
  if (theta_p >= 0) { taus[MAXPOP + theta_p] = tnew; taus[2*MAXPOP + theta_p] = log(2.0/(1.0*tnew)); }
  else if (tau_q >= 0) taus[tau_q] = tnew;
  else for (int p = 0; p < sp.npop; ++p) taus[p] *= mix_c;
}

__global__ void decide_kernel(const double * __restrict__ sum, double u, uint32_t epoch, uint32_t * flag,
                              uint32_t * counters, double * taus, Species sp, int tau_q, int theta_p, double win_u,
                              double mix_c, double mix_lnc)
{
  if (threadIdx.x || blockIdx.x) return;
  decide(sum[0], u, epoch, flag, counters, taus, sp, tau_q, theta_p, win_u, mix_c, mix_lnc);
}

// one GPU: the sum of lnl_sum_kernel (same order of additions) and the decision in ONE launch
__global__ void __launch_bounds__(1024) sum_decide_kernel(const double * __restrict__ term, uint32_t n, double u,
                                                          uint32_t epoch, uint32_t * flag, uint32_t * counters, double * taus,
                                                          Species sp, int tau_q, int theta_p, double win_u, double mix_c,
                                                          double mix_lnc)
{
  __shared__ double sh[1024];
  double acc = 0;
  for (uint32_t i = threadIdx.x; i < n; i += 1024) acc += term[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t w = 512; w > 0; w >>= 1)
  {
    if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) decide(sh[0], u, epoch, flag, counters, taus, sp, tau_q, theta_p, win_u, mix_c, mix_lnc);
}

// THETA (theta_step_all of a00_driver.c): one workgroup per population.  The per-locus terms come from the sufficient
// statistics the sweep left behind (no tree is loaded), are summed in a fixed order and the population's own decision
// is taken right here — the thetas are conditionally independent given the gene trees.  ext_sum != NULL: the sums were
// made (and all-reduced over the ranks) beforehand, only decide.
struct ThetaArgs { double win_u[MAXPOP], uacc[MAXPOP]; uint32_t on[MAXPOP]; };

__global__ void __launch_bounds__(1024) theta_sum_decide_kernel(const int8_t * __restrict__ pop_nc, const double * __restrict__ pop_t2h,
                                                                uint32_t T, double * taus, Species sp, ThetaArgs ta,
                                                                uint32_t * counters, double * sums_out, const double * ext_sum,
                                                                int decide_on)
{
  __shared__ double sh[1024];
  const int p = (int)blockIdx.x;
  if (!ta.on[p]) { if (threadIdx.x == 0 && sums_out) sums_out[p] = 0.0; return; }
  const double told = taus[MAXPOP + p], l2t_old = taus[2*MAXPOP + p];
  const double tnew = reflect(told + sp.ft_theta*(ta.win_u[p] - 0.5), 0.0, 999.0);
  const double l2t_new = log(2.0/(1.0*tnew));
  double total;
  if (ext_sum) total = ext_sum[p];
  else
  {
    double acc = 0;
    for (uint32_t i = threadIdx.x; i < T; i += 1024)
    {
      const int nc = pop_nc[(size_t)p*T + i]; const double t2h = pop_t2h[(size_t)p*T + i];
      acc += msc_term(nc, t2h, tnew, l2t_new) - msc_term(nc, t2h, told, l2t_old);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t w = 512; w > 0; w >>= 1)
    {
      if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
      __syncthreads();
    }
    total = sh[0];
    if (threadIdx.x == 0 && sums_out) sums_out[p] = total;
  }
  if (threadIdx.x || !decide_on) return;
  const double lnacc = total + ((sp.theta_alpha - 1)*log(tnew/told) - sp.theta_beta*(tnew - told));
  const bool accept = tnew > 0 && (lnacc >= 0 || ta.uacc[p] < exp(lnacc));
  atomicAdd(&counters[0], 1u); if (accept) atomicAdd(&counters[1], 1u);
  if (accept) { taus[MAXPOP + p] = tnew; taus[2*MAXPOP + p] = l2t_new; }
}


} // namespace smp

#include "sweep2.hpp"
#include "gsampler.hpp"
#include "gsampler2.hpp"
#include "bigsampler.hpp"

// ------------------------------------------------------------------------------------ host ---
// which device sampler takes the loci, when a test wants another than the one that fits (read when a sampler is made, not cached:
// a test process makes samplers of several kinds)
static bool smp_env_generic() { return getenv("BPA_SMP_GENERIC") != nullptr; }
static bool smp_env_big() { return getenv("BPA_SMP_BIG") != nullptr; }

struct bpa_sampler
{
  bpa_engine * eng = nullptr;
  unsigned nloci = 0, maxtips = 0;
  std::vector<bpa_locus *> loci;
  DevBuf<uint32_t> blk_task_off, flag, counters;
  DevBuf<smp::LaneRec> lane_rec;
  DevBuf<smp::TaskRec> task_rec;
  DevBuf<smp::Tree> trees, snap;
  DevBuf<double> mix_delta, mix_sum, taus, pop_t2h, theta_sums, lograt;
  // per-workgroup sums of an all-loci step and their stamps: UNCACHED device memory — the deciding workgroup reads what
  // workgroups on other XCDs (other L2s) wrote during the same launch (Args::dec_*)
  struct Uncached
  {
    void * p = nullptr;
    bool reserve(size_t bytes)
    {
      if (p) return true;
      if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess)
      {
        (void)hipGetLastError(); p = nullptr;
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
      }
      return p && hipMemset(p, 0, bytes) == hipSuccess;
    }
    void free() { if (p) (void)hipFree(p); p = nullptr; }
  } uc;
  double * wg_part = nullptr; unsigned long long * stamps = nullptr; int * dec_err = nullptr;
  bool fuse_decision = false;           // BPA_SMP_FUSE=1: the decision inside the step launch (measured: slower)
  DevBuf<int8_t> pop_nc;
  smp::Species sp{};                    // species tree (host copy; the taus below are only the start values)
  bool has_theta[smp::MAXPOP] = {};     // populations that can hold a coalescence (a00_initialize)
  // kernel timing (bpa_sampler_enable_timing): start / stop events attached to every stride-th launch, by kind
  struct Timed { hipEvent_t e0, e1; int kind; };
  std::vector<Timed> timed;
  unsigned timing_stride = 0, timing_phase = 0;      // 0: off
  double timed_ms[2] = {0, 0}; unsigned long timed_n[2] = {0, 0};     // 0 sweep (GAGE + GSPR of every locus), 1 all-loci step (TAU / MIX)
  bpa_allreduce_fn allreduce = nullptr; // several GPUs: sum the all-loci steps' device scalar over the ranks
  void * allreduce_ctx = nullptr;
  double * sum_ext = nullptr;           // caller-owned device scalar for that sum (NULL: internal)
  unsigned locus_offset = 0;            // global index of this rank's first locus (random streams)
  std::vector<unsigned> stream_index;   // a part of a composite: every locus's index in the whole set (its random stream)
  struct bpa_composite * comp = nullptr; // loci of several kinds: the parts (composite.hpp); this object then only dispatches
  std::vector<double> h_taus;
  std::vector<smp::Tree> h_trees;
  // ---- the generic path (gsampler.hpp): loci outside the sweep kernel's scope, evaluated by the engine's step kernels
  bool generic = false;
  std::vector<gsm::GTree> g_trees;      // host copies (instead of h_trees)
  DevBuf<gsm::GTree> g_dev, g_undo;
  DevBuf<gsm::GLocus> g_loc;
  DevBuf<double> g_lnl, g_lnlcur, g_hast, g_logpr, g_delta, g_site, g_len, g_lograt;
  // the program's THETA / TAU / MIX on a generic sampler (decided on the host: gsampler_host.hpp gs_prog_*)
  DevBuf<double> g_t2h3, g_progout;
  DevBuf<gsm::GDecState> g_dst; DevBuf<double> g_dsum; bool gp_dev = false;
  hipEvent_t gp_pace[4] = {nullptr, nullptr, nullptr, nullptr}; unsigned long gp_pace_n = 0;      // the end of each of the last iterations (gs_iterate: the host stays <= 2 iterations ahead)      // the program's all-loci decisions on the device (gdec_kernel): state, the sums' buffer
  unsigned long long gp_seq = 0;        // ... and the number of the launch whose arrival words the host polls (gs_prog_fetch)
  double * gp_pin = nullptr, * gp_pin_dev = nullptr;   // 64 doubles of pinned host memory the program's sum kernels write straight into (gs_prog_out)
  DevBuf<uint32_t> g_arrive;            // 20-state loci: tiles arrived per locus (the per-locus sum inside partials_lnl_wave20_kernel)
  long long gp_k[smp::MAXPOP] = {}; double gp_T[smp::MAXPOP] = {}; bool gp_ok = false, gp_pre_valid = false; double gp_pre_window = 0;
  double gp_tau[smp::MAXPOP] = {}, gp_theta[smp::MAXPOP] = {}; bool gp_mirror = false;
  unsigned long long gp_pj[10] = {}, gp_pj_base[4] = {};     // by move type since the last bpa_sampler_adapt_finetune: tau / mix / theta window on the host, the loci's age and prune-regraft moves from the trees (base: their totals at that call)      // the host's copy of the species tree (it takes every decision)
  // two half-batches of the per-locus steps on two streams (gsampler_host.hpp: gs_fork / gs_join): loci [0, g_isplit) are the
  // slots [0, g_ssplit) = workgroups [0, g_bsplit) of the engine's packing, the rest the other half
  bool g_split = false, g_forked = false;
  // (round 6: g_np part-batches, 2 by default; part p = loci [g_pi[p], g_pi[p+1]) = slots [g_ps[p], ..) = workgroups [g_pb[p], ..) of the
  //  packing — 20-state sets: tiles [g_pt[p], ..) —, launched on g_st[p]; g_st[0] is the engine's stream)
  static constexpr int GPARTS = 4;
  unsigned g_np = 1, g_pi[GPARTS + 1] = {}, g_ps[GPARTS + 1] = {}, g_pb[GPARTS + 1] = {}, g_pt[GPARTS + 1] = {};
  hipStream_t g_st[GPARTS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t g_ev_fork = nullptr, g_ev_join[GPARTS] = {nullptr, nullptr, nullptr, nullptr};
  int env_parts = 0;                    // BPA_GS_PARTS (experimental build): part-batches of the per-locus steps (default 2)
  DevBuf<uint8_t> g_active;
  DevBuf<uint4> g_recs;
  DevBuf<MatRec2> g_mat2;
  DevBuf<uint32_t> g_bmo;
  unsigned g_units = 0, g_maxmat = 0, g_pend = 0, g_npat = 0, g_rmax = 1, g_pack_epoch = 0;
  bool g_alljc = true;
  // more than 16 tips, scalers, diploid loci: the big-tree sampler (bigsampler.hpp), trees in HBM, the engine's general kernels
  bool big = false;
  std::vector<gbig::BTree> b_trees;
  DevBuf<gbig::BTree> b_dev, b_undo;
  DevBuf<uint32_t> b_thr;
  // 20-state loci: the generic sampler's steps as the records of the tiled kernels (gsampler.hpp: fmt20)
  bool g_s20 = false;
  unsigned g_ntiles = 0, g_maxops = 0;
  DevBuf<OpDev> g_ops20;
  DevBuf<uint32_t> g_oprng, g_root20, g_mtask, g_mpm, g_tlocus, g_tpat, g_ttask, g_tn0;
  DevBuf<int32_t> g_rscaler;
  // substitution-parameter moves (bpa_sampler_set_subst_moves): freqs | exchangeabilities | alpha per locus
  std::vector<double> g_sm_host;
  DevBuf<double> g_sm, g_sm_old;
  DevBuf<uint32_t> g_ids;
  double g_ft[3] = {0, 0, 0}, g_alpha_a = 1, g_alpha_b = 1;
  unsigned g_pend_mode = 0, g_pend_k = 0;
  bool g_eigen_dirty = false;
  bool g_level_eval = false;            // gs_level_roots: this evaluation stores every parent
  bool g_root_stale = false;            // a step's evaluation left the root's CLV unstored (flags bit 11): gs_download brings the buffers level
  bool g_pm_fused = false;              // the proposal launch just issued filled the step's P-matrices itself (gs_step -> gs_eval)
  unsigned long g_evals = 0;            // launches of the likelihood step kernel (bpa_sampler_work reports them as `sweeps`)
  unsigned nblocks = 0, epoch = 0;
  bool logpr_stale = false;     // thetas moved since the trees' densities were stored (Args::refresh_logpr)
  // diagnostic switches, read once at creation: BPA_SMP_DBG (bit mask, see Args::dbg), BPA_SMP_STEPS=g,q (proposal counts),
  // BPA_SMP_TRACE (per-launch times on stderr), BPA_SMP_NOMIX (sweeps only)
  uint32_t env_dbg = 0; int env_gage = -1, env_gspr = -1; bool env_trace = false, env_nomix = false;
  // the generic sampler's switches (A/B and kept paths; read ONCE, at creation: nothing reads the environment on the launch path,
  // and a test can still make samplers of either kind in one process)
  int env_fusea = -1; bool env_noeigfuse = false, env_fusepm = true, env_pmgroup = true, env_hostdec = false, env_rootstore = false; int env_chain = -1, env_pinout = 2;
  long env_inject = 0; uint32_t env_inject_bit = 1024u; long v2_launch_no = 0;    // BPA_SMP_INJECT=k[,w]: the k-th persistent launch with all-loci steps gives up at its first wait (w: workgroup 0 alone) — tests of the run-again path
  bool mix_pending = false;             // a mixing decision taken on the device has not been applied yet
  a00_rng_t grng = 0;
  unsigned long seed = 0, launches = 0, sweeps = 0;
  bool uploaded = false;
  uint32_t h_counters[4] = {0, 0, 0, 0};   // all-loci proposals / accepted (+ of those: THETA Gibbs draws / accepted) at the last invalidation (re-uploaded)
  bool host_current = false;            // the host copies (h_trees / g_trees, g_sm_host) hold the device's state: set by a download, cleared by iterate
  // ---- the persistent iteration kernel (sweep2.hpp): one GPU, loci that fit the sweep kernel, root = the last node
  bool v2_ok = false, env_v1 = false;
  int v2_nt = 0;                        // its instance: 4 or 8 tips
  int v2_retries = 0;                   // persistent launches that timed out in a row (sampler_download runs their iterations again)
  bool v2_prog = false;                 // ... with the program's moves: wave 0 of every workgroup is the control wave (no loci)
  unsigned v2_nwaves = 0, v2_nwg = 0, v2_lwaves = 0;
  size_t v2_lds = 0;
  DevBuf<uint32_t> v2_wave_off;
  DevBuf<smp2::Loc> v2_loc;
  DevBuf<uint2> v2_pat;
  DevBuf<unsigned long long> v2_xbuf;
  DevBuf<a00_rng_t> v2_grng;
  DevBuf<int> v2_err;
  DevBuf<unsigned long long> v2_pj;      // proposals / accepted by move type since the last bpa_sampler_adapt_finetune (sweep2.hpp Args::pj)
  DevBuf<double> v2_prof, v2_declog;
  DevBuf<smp::Species> v2_sp;
  smp::Species v2_sp_sent{};            // what v2_sp holds
  // the persistent launches since the last download, oldest first: what a launch that gave up (a shared device) is run again from
  struct V2Launch { a00_rng_t grng_before; unsigned chunk; bool mix_pending, logpr_stale; };
  std::vector<V2Launch> v2_log;
  bool kernel_bpp = false, v2_grng_sent = false;   // BPP's own generator + Bactrian-Laplace windows (bpa_sampler_set_proposal_kernel); the global stream then lives on the device
  bpa_p2p * p2p = nullptr;              // several GPUs, the sums exchanged INSIDE the persistent kernel over xGMI mailboxes (bpa_sampler_set_p2p)
  unsigned long v2_iters = 0;           // iterations run by persistent launches (bpa_sampler_timing)
};

static bpa_sampler * comp_create(bpa_engine_t * e, bpa_locus_t * const * loci, unsigned nloci, unsigned long seed, bool * plain);
static void comp_destroy(bpa_sampler * s);
static int comp_invalidate(bpa_sampler * s);
static int comp_upload(bpa_sampler * s);
static int comp_set_allreduce(bpa_sampler * s, bpa_allreduce_fn fn, void * ctx, double * device_sum, unsigned first_locus);
static int comp_run(bpa_sampler * s, int what, unsigned n);
static int comp_summary(bpa_sampler * s, double * total_lnl, unsigned long * proposals, unsigned long * accepted, unsigned long * launches);
static int comp_timing(bpa_sampler * s, double * sweep_ms, unsigned long * sweep_launches, double * allloci_ms, unsigned long * allloci_launches);
static int comp_work(bpa_sampler * s, double * bytes, unsigned long * node_updates, unsigned long * pattern_updates, unsigned long * sweeps);
static bpa_sampler * comp_part(bpa_sampler * s, unsigned i, unsigned * j);
static bpa_sampler * comp_part0(bpa_sampler * s);
template <class F> static int comp_each(bpa_sampler * s, F f);
#define COMP_FAIL(msg) do { if (s->comp) return fail(msg); } while (0)

static bpa_sampler * sampler_create_plain(bpa_engine_t * e, bpa_locus_t * const * loci, unsigned nloci, unsigned long seed);
extern "C" bpa_sampler_t * bpa_sampler_create(bpa_engine_t * e, bpa_locus_t * const * loci, unsigned nloci,
                                              unsigned long seed)
{
  if (!e || !loci || !nloci) { fail("bpa_sampler_create: null argument"); return nullptr; }
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  // loci of more than one kind: a part per kind, stepped together (composite.hpp)
  bool plain = true;
  bpa_sampler * c = comp_create(e, loci, nloci, seed, &plain);
  if (c || !plain) return c;
  return sampler_create_plain(e, loci, nloci, seed);
}

static bpa_sampler * sampler_create_plain(bpa_engine_t * e, bpa_locus_t * const * loci, unsigned nloci, unsigned long seed)
{
  bpa_sampler * s = new bpa_sampler();
  s->eng = e; s->nloci = nloci; s->seed = seed; s->grng = a00_rng_seed(seed, A00_GLOBAL_STREAM);
  s->sp.theta_slide_prob = 0.1;
  s->loci.assign(loci, loci + nloci);
  // the LDS sweep kernel (JC69, one rate category, <= 8 tips, <= 64 patterns) where every locus fits it, else the generic
  // path over the engine's step kernels (any 4-state model on the engine's packing, <= 16 tips; BPA_SMP_GENERIC=1 forces it)
  bool fits_sweep = !smp_env_generic(), fits_generic = true, all_jc = true, all_kl = true;
  unsigned n20 = 0, nbig = 0;
  const bool force_big = smp_env_big();
  for (unsigned i = 0; i < nloci; ++i)
  {
    const bpa_locus * l = loci[i];
    const bool counts = l && l->eng == e && l->alive && l->tips >= 2 && l->clv_buffers == 2*(l->tips - 1) && l->prob_matrices == 2*(2*l->tips - 2);
    // beyond 16 tips, with scalers or as an unphased diploid a 4-state locus takes the big-tree sampler (bigsampler.hpp)
    const bool wants_big = counts && l->states == 4 && (force_big || l->tips > (unsigned)gsm::NT || l->scale_buffers != 0 || l->dev.unphased_length);
    if (wants_big)
    {
      if (l->tips > (unsigned)gbig::BT || (l->scale_buffers != 0 && l->scale_buffers != 2*(l->tips - 1)))
      {
        fail("bpa_sampler_create: a locus of the big-tree sampler has <= 64 tips and no or 2 (tips - 1) scale buffers (method.c:4110-4146)");
        delete s; return nullptr;
      }
      ++nbig; s->maxtips = std::max(s->maxtips, l->tips); s->g_rmax = std::max(s->g_rmax, l->rate_cats);
      continue;
    }
    const bool base = counts && (l->states == 4 || l->states == 20) && l->scale_buffers == 0 && !l->dev.unphased_length;
    if (!base)
    {
      fail("bpa_sampler_create: loci must be 4- or 20-state (20-state: without scalers, not diploid), with the buffer counts of method.c:4110-4146");
      delete s; return nullptr;
    }
    if (l->states == 20) { ++n20; fits_sweep = false; fits_generic = fits_generic && l->tips <= (unsigned)gsm::NT && l->rate_cats <= 4; all_jc = false;
                           s->maxtips = std::max(s->maxtips, l->tips); s->g_rmax = std::max(s->g_rmax, l->rate_cats); continue; }
    fits_sweep = fits_sweep && l->rate_cats == 1 && l->dev.model == 0 && l->tips <= (unsigned)smp::MAXTIPS && l->sites <= (unsigned)smp::BS;
    fits_generic = fits_generic && l->tips <= (unsigned)gsm::NT && l->rate_cats <= 8 && l->sites*l->rate_cats < PACK_BS;
    all_jc = all_jc && l->rate_cats == 1 && l->dev.model == 0;
    all_kl = all_kl && l->rate_cats > 1;
    s->maxtips = std::max(s->maxtips, l->tips);
    s->g_rmax = std::max(s->g_rmax, l->rate_cats);
  }
  if (nbig)
  {
    // one big locus makes the whole sampler the big-tree one (its kernels take any 4-state locus)
    for (unsigned i = 0; i < nloci; ++i)
      if (loci[i]->states != 4 || loci[i]->tips > (unsigned)gbig::BT || (loci[i]->scale_buffers != 0 && loci[i]->scale_buffers != 2*(loci[i]->tips - 1)))
      {
        fail("bpa_sampler_create: the loci of a big-tree sampler are all 4-state with <= 64 tips");
        delete s; return nullptr;
      }
    for (unsigned i = 0; i < nloci; ++i) { s->maxtips = std::max(s->maxtips, loci[i]->tips); s->g_rmax = std::max(s->g_rmax, loci[i]->rate_cats); }
    s->big = true;
    s->b_trees.assign(nloci, gbig::BTree{});
  }
  else if (n20)
  {
    // amino-acid loci: the generic sampler's proposal control, the likelihood by the tiled 20-state kernels
    if (n20 != nloci || !fits_generic)
    {
      fail("bpa_sampler_create: 20-state loci go together (no 4-state locus in the same sampler), with <= 16 tips and <= 4 rate categories");
      delete s; return nullptr;
    }
    s->generic = true; s->g_alljc = false; s->g_s20 = true;
    s->g_trees.assign(nloci, gsm::GTree{});
  }
  else if (!fits_sweep)
  {
    if (!fits_generic || !(all_jc || all_kl))
    {
      fail("bpa_sampler_create: beyond the sweep kernel (JC69, 1 rate category, <= 8 tips, <= 64 patterns) the loci must all be JC69 with "
           "one rate category or all have several categories, with <= 16 tips and < 256 patterns x categories");
      delete s; return nullptr;
    }
    s->generic = true; s->g_alljc = all_jc;
    s->g_trees.assign(nloci, gsm::GTree{});
  }
  s->h_trees.assign(nloci, smp::Tree{});
  if (const char * dv = getenv("BPA_SMP_DBG")) s->env_dbg = (uint32_t)atoi(dv);
  if (const char * st = BPA_EXP_SWITCH("BPA_SMP_STEPS")) { unsigned g = 0, q = 0; if (sscanf(st, "%u,%u", &g, &q) == 2) { s->env_gage = (int)g; s->env_gspr = (int)q; } }
  s->env_trace = getenv("BPA_SMP_TRACE") != nullptr; s->env_nomix = BPA_EXP_SWITCH("BPA_SMP_NOMIX") != nullptr;
  if (const char * inj = getenv("BPA_SMP_INJECT")) { s->env_inject = atol(inj); s->env_inject_bit = strstr(inj, ",w") ? 2048u : 1024u; }
  s->fuse_decision = BPA_EXP_SWITCH("BPA_SMP_FUSE") != nullptr;
  { const char * v;
    v = BPA_EXP_SWITCH("BPA_GS_FUSEA");     s->env_fusea = v ? (v[0] == '1' ? 1 : 0) : -1;
    v = BPA_EXP_SWITCH("BPA_GS_FUSEPM");    s->env_fusepm = !(v && v[0] == '0');
    v = BPA_EXP_SWITCH("BPA_S20_PMGROUP");  s->env_pmgroup = !(v && v[0] == '0');
    v = BPA_EXP_SWITCH("BPA_GS_PARTS");    s->env_parts = v ? atoi(v) : 0;
    s->env_noeigfuse = BPA_EXP_SWITCH("BPA_GS_NOEIGFUSE") != nullptr;        // (A/B: the eigensystem refresh as a launch of its own)
    v = getenv("BPA_GS_CHAIN");     s->env_chain = v ? (v[0] != '0' ? 1 : 0) : -1;
    v = BPA_EXP_SWITCH("BPA_GS_PINOUT");    s->env_pinout = v ? (v[0] == '0' ? 0 : v[0] == '1' ? 1 : 2) : 2;
    s->env_hostdec = getenv("BPA_GS_HOSTDEC") != nullptr;
    s->env_rootstore = BPA_EXP_SWITCH("BPA_GS_ROOTSTORE") != nullptr; }
  s->env_v1 = getenv("BPA_SMP_V1") != nullptr;
  s->sp.ft_gage = 0.004; s->sp.ft_gspr = 0.004; s->sp.ft_tau = 0.001; s->sp.ft_mix = 0.3;      // a00_create's defaults
  return s;
}

static int sampler_download(bpa_sampler * s);
extern "C" void bpa_sampler_destroy(bpa_sampler_t * s)
{
  if (!s) return;
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) comp_destroy(s);
  (void)set_device(s->eng);
  // (the loci outlive the sampler: their buffers are left as a step-by-step caller would have left them, gs_level_roots)
  if (s->generic && s->uploaded && s->g_root_stale && s->g_pack_epoch == s->eng->pack_epoch && !sampler_download(s)) fprintf(stderr, "[bpp_amd] bpa_sampler_destroy: the loci's root buffers could not be brought level (%s)\n", bpa_last_error());
  (void)hipStreamSynchronize(s->eng->stream);
  for (auto & t : s->timed) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
  s->blk_task_off.free(); s->lane_rec.free(); s->task_rec.free(); s->flag.free();
  s->counters.free(); s->trees.free(); s->snap.free(); s->mix_delta.free(); s->mix_sum.free(); s->taus.free(); s->pop_t2h.free(); s->theta_sums.free(); s->pop_nc.free(); s->lograt.free(); s->uc.free();
  for (int p_ = 1; p_ < bpa_sampler::GPARTS; ++p_)
  {
    if (s->g_st[p_]) { (void)hipStreamSynchronize(s->g_st[p_]); (void)hipStreamDestroy(s->g_st[p_]); s->g_st[p_] = nullptr; }
    if (s->g_ev_join[p_]) { (void)hipEventDestroy(s->g_ev_join[p_]); s->g_ev_join[p_] = nullptr; }
  }
  if (s->g_ev_fork) { (void)hipEventDestroy(s->g_ev_fork); s->g_ev_fork = nullptr; }
  s->g_t2h3.free(); s->g_progout.free(); s->g_arrive.free(); s->gp_mirror = false; s->g_dst.free(); s->g_dsum.free();
  if (s->gp_pin) { (void)hipHostFree(s->gp_pin); s->gp_pin = s->gp_pin_dev = nullptr; }
  for (auto & ev : s->gp_pace) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
  s->g_dev.free(); s->g_undo.free(); s->g_loc.free(); s->g_lnl.free(); s->g_lnlcur.free(); s->g_hast.free(); s->g_logpr.free(); s->g_delta.free(); s->g_site.free();
  s->g_len.free(); s->g_lograt.free(); s->g_active.free(); s->g_recs.free(); s->g_mat2.free(); s->g_bmo.free();
  s->g_sm.free(); s->g_sm_old.free(); s->g_ids.free();
  s->b_dev.free(); s->b_undo.free(); s->b_thr.free();
  s->g_ops20.free(); s->g_oprng.free(); s->g_root20.free(); s->g_mtask.free(); s->g_mpm.free(); s->g_tlocus.free(); s->g_tpat.free(); s->g_ttask.free(); s->g_tn0.free(); s->g_rscaler.free();
  s->v2_wave_off.free(); s->v2_loc.free(); s->v2_pat.free(); s->v2_xbuf.free(); s->v2_grng.free(); s->v2_err.free(); s->v2_prof.free(); s->v2_declog.free(); s->v2_sp.free();
  delete s;
}

// A setter that makes the next call upload the host copies again (a new tree, species tree, tip assignment, stream
// offset) first brings them level with the device: trees with their random streams and counters, taus and thetas, the
// all-loci counters — a run that is reconfigured half-way continues from where it was, not from the last download
static int sampler_download(bpa_sampler * s);
static bool v2_wants_prog(const bpa_sampler * s);
static int sampler_invalidate(bpa_sampler * s)
{
  if (s->uploaded)
  {
    bpa_engine * e = s->eng;
    if (!set_device(e) || !sampler_download(s)) return 0;
    HIPCHK(hipMemcpy(s->h_taus.data(), s->taus.p, s->h_taus.size()*sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(s->h_counters, s->counters.p, sizeof s->h_counters, hipMemcpyDeviceToHost));
  }
  s->uploaded = false; s->host_current = false;
  return 1;
}

template <class TR, int MAXNODES>
static int set_tree_fields(TR & t, int tips, const int * left, const int * right, const double * times, int root, a00_rng_t rng)
{
  const int n = 2*tips - 1;
  std::memset(&t, 0, sizeof(t));
  for (int k = 0; k < MAXNODES; ++k) { t.left[k] = t.right[k] = t.parent[k] = -1; t.clv[k] = t.pmat[k] = (int8_t)k; t.pop[k] = (int8_t)(k < tips ? k : -1); }
  for (int k = 0; k < n; ++k)
  {
    t.left[k] = (int8_t)left[k]; t.right[k] = (int8_t)right[k]; t.time[k] = times[k];
    if ((left[k] < 0) != (k < tips) || (right[k] < 0) != (k < tips) || left[k] >= n || right[k] >= n)
      return fail("bpa_sampler_set_tree: nodes 0..tips-1 are the tips (no children), the others have two");
    if (left[k] >= 0) { t.parent[left[k]] = (int8_t)k; t.parent[right[k]] = (int8_t)k; }
  }
  t.root = root; t.tips = tips; t.rng = rng; t.lnl = 0;
  return 1;
}

// start state of a stream (index = global locus index, or A00_GLOBAL_STREAM): our 64-bit one, or — BPP's kernel — the
// 32-bit legacy_rndu state the host driver derives from the same seeding function (a00_create: rng >> 16)
static unsigned stream_of(const bpa_sampler * s, unsigned i) { return s->stream_index.empty() ? s->locus_offset + i : s->stream_index[i]; }
static a00_rng_t stream_seed(const bpa_sampler * s, unsigned stream)
{
  const a00_rng_t z = a00_rng_seed(s->seed, stream);
  return s->kernel_bpp ? (a00_rng_t)(unsigned int)(z >> 16) : z;
}

extern "C" int bpa_sampler_set_proposal_kernel(bpa_sampler_t * s, int kind)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp && kind != BPA_KERNEL_UNIFORM) return fail("bpa_sampler_set_proposal_kernel: loci of several kinds run the library's own proposal kernel (BPP's lives inside the persistent kernel's single launch)");
  if (s->comp) return 1;
  if (kind != BPA_KERNEL_UNIFORM && kind != BPA_KERNEL_BPP) return fail("bpa_sampler_set_proposal_kernel: BPA_KERNEL_UNIFORM or BPA_KERNEL_BPP");
  if (s->uploaded) return fail("bpa_sampler_set_proposal_kernel: before bpa_sampler_initialize (as a00_set_proposal_kernel)");
  if (s->big && kind == BPA_KERNEL_BPP) return fail("bpa_sampler_set_proposal_kernel: BPP's kernel runs in the persistent iteration kernel and in the generic sampler (loci of <= 16 tips)");
  s->kernel_bpp = kind == BPA_KERNEL_BPP;
  // (the generic sampler: together with bpa_sampler_set_program_moves and a theta prior — checked when the run starts)
  for (unsigned i = 0; i < s->nloci; ++i) (s->generic ? s->g_trees[i].rng : s->h_trees[i].rng) = stream_seed(s, stream_of(s, i));
  s->grng = stream_seed(s, A00_GLOBAL_STREAM);
  s->v2_grng_sent = false;
  return 1;
}

extern "C" int bpa_sampler_set_program_moves(bpa_sampler_t * s, int on, double slide_prob)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) return on ? fail("bpa_sampler_set_program_moves: loci of several kinds run the library's own moves") : 1;
  if (on && !(slide_prob >= 0 && slide_prob <= 1)) return fail("bpa_sampler_set_program_moves: slide_prob is a probability");
  if (on && !s->kernel_bpp) return fail("bpa_sampler_set_program_moves: the program's moves draw from BPP's proposal kernel (bpa_sampler_set_proposal_kernel(s, BPA_KERNEL_BPP) first)");
  s->sp.program_moves = on ? 1 : 0;
  if (on) s->sp.theta_slide_prob = slide_prob;
  // (the persistent kernel's form — with or without a control wave — is chosen at upload)
  if (s->uploaded && s->v2_ok && v2_wants_prog(s) != s->v2_prog) return sampler_invalidate(s);
  return 1;
}

extern "C" int bpa_sampler_gibbs_counters(bpa_sampler_t * s, unsigned long * proposals, unsigned long * accepted)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) { if (proposals) *proposals = 0; if (accepted) *accepted = 0; return 1; }
  if (!sampler_download(s)) return 0;
  uint32_t c[4];
  HIPCHK(hipMemcpy(c, s->counters.p, sizeof c, hipMemcpyDeviceToHost));
  if (proposals) *proposals = c[2];
  if (accepted) *accepted = c[3];
  return 1;
}

extern "C" int bpa_sampler_set_tree(bpa_sampler_t * s, unsigned i, const int * left, const int * right,
                                    const double * times, int root)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) { unsigned j; bpa_sampler * p = comp_part(s, i, &j); if (!p) return fail("bpa_sampler_set_tree: locus index out of range"); return comp_invalidate(s) && bpa_sampler_set_tree(p, j, left, right, times, root); }
  if (i >= s->nloci) return fail("bpa_sampler_set_tree: locus index out of range");
  const int tips = (int)s->loci[i]->tips;
  const a00_rng_t rng = stream_seed(s, stream_of(s, i));
  if (!sampler_invalidate(s)) return 0;
  if (s->big) return set_tree_fields<gbig::BTree, gbig::BN>(s->b_trees[i], tips, left, right, times, root, rng);
  if (s->generic) return set_tree_fields<gsm::GTree, gsm::NN>(s->g_trees[i], tips, left, right, times, root, rng);
  return set_tree_fields<smp::Tree, smp::MAXN>(s->h_trees[i], tips, left, right, times, root, rng);
}

// populations of the inner nodes (assign_pops of a00_driver.c): common population of the children, then up to the age
template <class TR>
static int assign_pops_host(const bpa_sampler * s, TR & t)
{
  const int n = 2*t.tips - 1;
  std::vector<int> order;
  for (int k = t.tips; k < n; ++k) order.push_back(k);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return t.time[a] < t.time[b]; });
  for (int k = 0; k < t.tips; ++k) if (t.pop[k] < 0 || t.pop[k] >= s->sp.S) return fail("bpa_sampler: tip species out of range");
  for (int v : order)
  {
    int c = t.pop[t.left[v]];
    while (!((s->sp.anc[t.pop[t.right[v]]] >> c) & 1u)) c = s->sp.parent[c];
    if (t.time[v] < s->h_taus[c]) return fail("bpa_sampler: a gene-tree node is younger than the divergence of its descendants' species");
    while (s->sp.parent[c] >= 0 && s->h_taus[s->sp.parent[c]] <= t.time[v]) c = s->sp.parent[c];
    t.pop[v] = (int8_t)c;
  }
  return 1;
}
static int gs_upload(bpa_sampler * s);
static int sampler_launch(bpa_sampler * s, unsigned mode, double mix_c, double mix_lnc, unsigned tau_q, double tau_u, double dec_uacc);
static int gs_initialize(bpa_sampler * s);
static int gs_iterate(bpa_sampler * s, unsigned iterations);
static int gs_download(bpa_sampler * s);
static int gb_upload(bpa_sampler * s);
static int gb_initialize(bpa_sampler * s);
static int gb_iterate(bpa_sampler * s, unsigned iterations);
static int gb_download(bpa_sampler * s);

// ---- the persistent iteration kernel (sweep2.hpp): tables, eligibility, launch
template <int NT> static size_t v2_lds_base(bool prog)
{
  return ((sizeof(smp2::WgLDS<NT>) + 15) & ~(size_t)15) + sizeof(smp2::WaveLDS<NT>)*(size_t)(smp2::Cfg<NT>::WAVES - (prog ? 1 : 0));
}
// the program's moves (BPP's kernel + bpa_sampler_set_program_moves + a theta prior + a theta to move): the persistent kernel's
// form with a control wave per workgroup
static bool v2_wants_prog(const bpa_sampler * s)
{
  if (!(s->kernel_bpp && s->sp.program_moves && s->sp.theta_alpha > 0 && s->sp.npop > s->sp.S)) return false;
  for (int p = 0; p < s->sp.npop; ++p) if (s->has_theta[p]) return true;
  return false;
}
// v2_ok stays false where the kernel does not apply (the one-launch-per-step path of this file then runs): more loci than
// stay resident on the device at once, a tree whose root is not its last node or whose buffer indices are not the
// reference's start values toggled (gtree.c:2398, 2433: clv_index = pmatrix_index = node index), BPA_SMP_V1=1
static int sampler_upload_v2(bpa_sampler * s, const std::vector<smp::TaskRec> & task_rec)
{
  s->v2_ok = false;
  if (s->env_v1 || s->generic || s->big) return 1;
  const unsigned T = s->nloci;
  const int npop = s->sp.npop;
  if (s->maxtips > 8 || npop > 15) return 1;
  const int NT = (s->maxtips <= 4 && npop <= 7) ? 4 : 8;
  for (unsigned i = 0; i < T; ++i)
  {
    const smp::Tree & t = s->h_trees[i];
    const int n = 2*t.tips - 1, inner = t.tips - 1, edges = 2*t.tips - 2;
    if (t.root != n - 1) return 1;
    for (int k = 0; k < n; ++k)
    {
      if (k < t.tips ? t.clv[k] != k : (t.clv[k] != k && t.clv[k] != k + inner)) return 1;
      if (k != t.root && t.pmat[k] != k && t.pmat[k] != k + edges) return 1;
    }
  }
  const unsigned G = NT == 4 ? 8 : 16, LPW = 64/G, WAVES = NT == 4 ? (unsigned)smp2::Cfg<4>::WAVES : (unsigned)smp2::Cfg<8>::WAVES;
  std::vector<uint32_t> woff{0};
  std::vector<smp2::Loc> loc(T);
  std::vector<uint2> pat;
  unsigned cnt = 0, used = 0;
  for (unsigned t = 0; t < T; ++t)
  {
    const bpa_locus * l = s->loci[t];
    const unsigned np = l->sites, tips = l->tips;
    if (cnt == LPW || used + np > 64u) { woff.push_back(t); cnt = 0; used = 0; }
    ++cnt; used += np;
    smp2::Loc & L = loc[t];
    const smp::TaskRec & r = task_rec[t];
    L.clv = r.clv; L.pmat = r.pmat; L.rate = r.rate; L.rw = r.rw; L.f0 = r.f0; L.f1 = r.f1; L.f2 = r.f2; L.f3 = r.f3;
    L.np = np; L.tips = tips; L.pat_off = (uint32_t)pat.size(); L.pad = 0;
    for (int p = 0; p < 16; ++p) L.gl[p] = r.gl[p];
    for (unsigned n = 0; n < np; ++n)
    {
      uint32_t codes = 0;
      for (unsigned tip = 0; tip < tips; ++tip) codes |= (uint32_t)(l->tipcodes[(size_t)tip*np + n] & 15u) << (4*tip);
      pat.push_back(make_uint2(l->weights[n], codes));
    }
  }
  woff.push_back(T);
  const bool prog = v2_wants_prog(s);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, s->eng->device));
  // waves with loci per workgroup: as few as put every wave's workgroup on a CU of its own (two waves on one SIMD share its
  // issue slots: the pair takes a third longer than a wave alone) — 4 where the loci allow, i.e. up to 4 x 8 x CUs loci
  const unsigned nwaves = (unsigned)woff.size() - 1;
  const unsigned LMAX = WAVES - (prog ? 1u : 0u), ncu = (unsigned)std::max(prop.multiProcessorCount, 1);
  unsigned LWAVES = (LMAX > 4u && nwaves <= 4u*ncu) ? 4u : LMAX;
  if (const char * ev = BPA_EXP_SWITCH("BPA_SMP_LWAVES")) { const unsigned v = (unsigned)std::atoi(ev); if (v >= 1 && v <= LMAX) LWAVES = v; }     // (experiments)       // (a pair in every workgroup anyway beyond that: then as few workgroups as possible)
  const unsigned nwg = (nwaves + LWAVES - 1)/LWAVES;
  // every workgroup must be resident (they wait for each other's sums): one per CU — a workgroup takes most of a CU's LDS
  const size_t base = NT == 4 ? v2_lds_base<4>(prog) : v2_lds_base<8>(prog);
  const size_t lds_max = std::min<size_t>((size_t)prop.sharedMemPerBlock > 65536 ? (size_t)prop.sharedMemPerBlock : 160*1024, 160*1024) - 256;
  if (base > lds_max) return 1;
  // (the LDS decides how many workgroups a CU holds; two waves per SIMD at most are counted on)
  const unsigned per_cu = (unsigned)std::min<size_t>(std::max<size_t>(lds_max/base, 1), 8/WAVES ? 8/WAVES : 1);
  if (nwg > per_cu*(unsigned)prop.multiProcessorCount) return 1;
  s->v2_lds = base;
  s->v2_nt = NT; s->v2_nwaves = nwaves; s->v2_nwg = nwg; s->v2_prog = prog; s->v2_lwaves = LWAVES;
  const int zero2v[3] = {0, 0, 0};
  if (!upload(s->v2_wave_off, woff.data(), woff.size()) || !upload(s->v2_loc, loc.data(), loc.size()) ||
      !upload(s->v2_pat, pat.data(), pat.size()) || !s->v2_xbuf.reserve((size_t)2*smp2::XN) || !s->v2_grng.reserve(1) ||
      !upload(s->v2_err, zero2v, 3) || !s->v2_prof.reserve(40 + (size_t)nwg) || !s->v2_declog.reserve(4*2048) || !s->v2_sp.reserve(1) || !s->v2_pj.reserve(16))
    return 0;
  HIPCHK(hipMemset(s->v2_pj.p, 0, 16*sizeof(unsigned long long)));
  HIPCHK(hipMemset(s->v2_prof.p, 0, (40 + (size_t)nwg)*sizeof(double)));
  {
    void (*k0)(const smp2::Args) = NT == 4 ? smp2::iter_kernel<4, false> : smp2::iter_kernel<8, false>;
    void (*k1)(const smp2::Args) = NT == 4 ? smp2::iter_kernel<4, true> : smp2::iter_kernel<8, true>;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->v2_lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->v2_lds));
    void (*k2)(const smp2::Args) = NT == 4 ? smp2::iter_kernel<4, true, true> : smp2::iter_kernel<8, true, true>;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->v2_lds));
  }
  std::memset(&s->v2_sp_sent, 0xff, sizeof s->v2_sp_sent);       // (nothing sent yet)
  s->v2_ok = true;
  return 1;
}

static int sampler_upload(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  if (s->uploaded) return 1;
  if (!set_device(e) || !flush(e)) return 0;
  const unsigned T = s->nloci;
  if (!s->sp.npop) return fail("bpa_sampler: set the species tree first (bpa_sampler_set_species_tree)");
  if (s->big) return gb_upload(s);
  if (s->generic) return gs_upload(s);
  for (unsigned i = 0; i < T; ++i) if (!assign_pops_host(s, s->h_trees[i])) return 0;
  for (int p = 0; p < smp::MAXPOP; ++p) s->has_theta[p] = p >= s->sp.S && p < s->sp.npop;
  for (unsigned i = 0; i < T; ++i)
  {
    int cnt[smp::MAXPOP] = {0};
    const smp::Tree & t = s->h_trees[i];
    for (int k = 0; k < t.tips; ++k) if (++cnt[t.pop[k]] >= 2) s->has_theta[t.pop[k]] = true;
  }
  std::vector<uint32_t> blk_off{0};
  std::vector<smp::LaneRec> lane_rec;
  std::vector<smp::TaskRec> task_rec(T);
  const smp::LaneRec idle{0xffffffffu, 0, 0, 0, nullptr, nullptr};
  unsigned used = 0, ntask = 0;
  for (unsigned t = 0; t < T; ++t)
  {
    const bpa_locus * l = s->loci[t];
    const unsigned np = l->sites, tips = l->tips;
    if (used + np > (unsigned)smp::BS || ntask == (unsigned)smp::TPB)
    { lane_rec.resize(blk_off.size()*smp::BS, idle); blk_off.push_back(t); used = 0; ntask = 0; }
    for (unsigned n = 0; n < np; ++n)
    {
      uint32_t codes = 0;
      for (unsigned tip = 0; tip < tips; ++tip) codes |= (uint32_t)(l->tipcodes[(size_t)tip*np + n] & 15u) << (4*tip);
      lane_rec.push_back(smp::LaneRec{t, l->weights[n], codes, n | np << 8 | tips << 16, l->dev.clv + (size_t)n*4, l->dev.pmat});
    }
    used += np; ++ntask;
    smp::TaskRec & r = task_rec[t];
    const double * f = l->par.data() + par_matrix(1, 4, 0) + pm_freqs(4);
    r.clv = l->dev.clv; r.pmat = l->dev.pmat; r.rate = l->par[par_rates(1)]; r.rw = l->par[par_rate_weights(1)];
    r.f0 = f[0]; r.f1 = f[1]; r.f2 = f[2]; r.f3 = f[3];
    const smp::Tree & tr = s->h_trees[t];
    for (int p = 0; p < smp::MAXPOP; ++p) r.gl[p] = r.nin[p] = 0;
    for (unsigned k = 0; k < tips; ++k)
    {
      r.nin[tr.pop[k]]++;
      for (int q = tr.pop[k]; q >= 0; q = s->sp.parent[q]) r.gl[q]++;
    }
  }
  lane_rec.resize(blk_off.size()*smp::BS, idle);
  blk_off.push_back(T);
  s->nblocks = (unsigned)blk_off.size() - 1;
  uint32_t zero2[2] = {0, 0};
  if (!upload(s->blk_task_off, blk_off.data(), blk_off.size()) ||
      !upload(s->lane_rec, lane_rec.data(), lane_rec.size()) || !upload(s->task_rec, task_rec.data(), T) ||
      !upload(s->trees, s->h_trees.data(), T) || !upload(s->snap, s->h_trees.data(), T) ||
      !upload(s->flag, zero2, 1) || !upload(s->counters, s->h_counters, 4) || !s->mix_delta.reserve(T) || !s->mix_sum.reserve(1) ||
      !upload(s->taus, s->h_taus.data(), s->h_taus.size()) || !s->pop_t2h.reserve((size_t)T*smp::MAXPOP) ||
      !s->pop_nc.reserve((size_t)T*smp::MAXPOP) || !s->theta_sums.reserve(smp::MAXPOP) || !s->lograt.reserve(smp::MAXN*smp::MAXN))
    return 0;
  {
    // [nblocks] sums | [nblocks] stamps | error flag
    const size_t nb = ((size_t)s->nblocks + 1)*8;
    s->uc.free();
    if (!s->uc.reserve(2*nb + 64)) return fail("bpa_sampler: out of (uncached) device memory");
    s->wg_part = reinterpret_cast<double *>(s->uc.p);
    s->stamps = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(s->uc.p) + nb);
    s->dec_err = reinterpret_cast<int *>(reinterpret_cast<char *>(s->uc.p) + 2*nb);
  }
  if (s->allreduce)
  {
    // Sharded run: which populations can hold a coalescence is a property of ALL loci, not of this rank's — every rank
    // draws two numbers of the shared global stream per such population (bpa_sampler_iterate), so the mask must be the
    // same everywhere or the ranks' streams drift apart.  OR over the ranks = sum of 0/1 through the caller's collective.
    double m[smp::MAXPOP];
    for (int p = 0; p < smp::MAXPOP; ++p) m[p] = s->has_theta[p] ? 1.0 : 0.0;
    double * ar = s->sum_ext ? s->sum_ext : s->theta_sums.p;
    HIPCHK(hipMemcpyAsync(ar, m, sizeof m, hipMemcpyHostToDevice, e->stream));
    if (!s->allreduce(s->allreduce_ctx, ar, (unsigned)smp::MAXPOP, (void *)e->stream)) return fail("bpa_sampler: the all-reduce callback failed");
    HIPCHK(hipMemcpyAsync(m, ar, sizeof m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int p = 0; p < smp::MAXPOP; ++p) s->has_theta[p] = m[p] > 0.5;
  }
  if (s->p2p && !s->allreduce)
  {
    // (the same OR over the ranks through the mailboxes)
    double m[smp::MAXPOP];
    for (int p = 0; p < smp::MAXPOP; ++p) m[p] = s->has_theta[p] ? 1.0 : 0.0;
    HIPCHK(hipMemcpyAsync(s->theta_sums.p, m, sizeof m, hipMemcpyHostToDevice, e->stream));
    if (!bpa_p2p_allreduce(s->p2p, s->theta_sums.p, (unsigned)smp::MAXPOP)) return 0;
    HIPCHK(hipMemcpyAsync(m, s->theta_sums.p, sizeof m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (bpa_p2p_status(s->p2p) != 0) return fail("bpa_sampler: the p2p exchange timed out while the ranks agreed on the THETA mask");
    for (int p = 0; p < smp::MAXPOP; ++p) s->has_theta[p] = m[p] > 0.5;
  }
  hipLaunchKernelGGL(smp::lograt_kernel, dim3(1), dim3(smp::MAXN*smp::MAXN), 0, e->stream, s->lograt.p);
  HIPCHK(hipGetLastError());
  s->epoch = 0; s->mix_pending = false;
  if (!sampler_upload_v2(s, task_rec)) return 0;
  if (s->p2p && !s->allreduce && !s->v2_ok)
    return fail("bpa_sampler_set_p2p: the in-kernel exchange needs the persistent iteration kernel (JC69 loci of <= 8 tips and <= 64 patterns, "
                "root = last node, every workgroup resident); install an all-reduce callback instead");
  s->uploaded = true;
  return 1;
}

static int sampler_timing_drain(bpa_sampler * s)
{
  if (s->timed.empty()) return 1;
  HIPCHK(hipStreamSynchronize(s->eng->stream));
  for (auto & t : s->timed)
  {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, t.e0, t.e1));
    s->timed_ms[t.kind] += ms; s->timed_n[t.kind]++;
    (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1);
  }
  s->timed.clear();
  return 1;
}

// dec_uacc >= 0: an all-loci step (mode 1 / 4) that takes its own decision (Args::dec_*)
static int sampler_launch(bpa_sampler * s, unsigned mode, double mix_c, double mix_lnc = 0, unsigned tau_q = 0, double tau_u = 0,
                          double dec_uacc = -1.0)
{
  bpa_engine * e = s->eng;
  smp::Args a{};
  if (dec_uacc >= 0)
  {
    a.dec_on = 1u; a.dec_epoch = s->epoch + 1; a.dec_uacc = dec_uacc;
    a.dec_flag = s->flag.p; a.dec_counters = s->counters.p; a.dec_taus = s->taus.p; a.dec_stamps = s->stamps; a.dec_err = s->dec_err;
  }
  a.lane_rec = s->lane_rec.p; a.task_rec = s->task_rec.p; a.blk_task_off = s->blk_task_off.p;
  a.trees = s->trees.p; a.snap = s->snap.p;
  a.mix_delta = s->mix_delta.p; a.mix_flag = s->flag.p; a.mode = mode;
  // the first launch after an all-loci decision applies it (restore from the snapshot when it was a
  // rejection); every other launch passes epoch 0 = nothing pending
  a.epoch = s->mix_pending ? s->epoch : 0u;
  s->mix_pending = false;
  a.bfbeta = e->usedata ? e->bfbeta : 0.0;
  a.refresh_logpr = s->logpr_stale ? 1u : 0u; s->logpr_stale = false;
  a.pop_nc = s->pop_nc.p; a.pop_t2h = s->pop_t2h.p; a.lograt = s->lograt.p; a.ntasks = s->nloci; a.part_out = s->wg_part;
  a.dbg = s->env_dbg;
  a.taus = s->taus.p; a.tau_q = tau_q; a.tau_u = tau_u; a.sp = s->sp; a.mix_lnc = mix_lnc;
  a.nsteps_gage = s->maxtips - 1; a.nsteps_gspr = 2*s->maxtips - 2; a.mix_c = mix_c;
  if (s->env_gage >= 0) { a.nsteps_gage = (uint32_t)s->env_gage; a.nsteps_gspr = (uint32_t)s->env_gspr; }
  // the unrolled node passes of the leader's code are sized by the largest tree: 4-tip loci get their own instance
  void (*kern)(const smp::Args) = s->maxtips <= 4 ? smp::sweep_kernel<4> : smp::sweep_kernel<smp::MAXTIPS>;
  const size_t lds = (size_t)2*(s->maxtips - 1)*smp::BS*4*sizeof(double);
  if (s->env_trace)
  {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipExtLaunchKernelGGL(kern, dim3(s->nblocks), dim3(smp::BS), lds, e->stream, e0, e1, 0, a);
    (void)hipStreamSynchronize(e->stream);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    fprintf(stderr, "[smp] mode %u gage %u gspr %u epoch %u blocks %u: %.1f us\n", mode, a.nsteps_gage, a.nsteps_gspr, a.epoch, s->nblocks, ms*1e3);
    if (a.dbg & 16u)
    {
      double ph[8]; (void)hipMemcpy(ph, s->mix_delta.p + 8, sizeof ph, hipMemcpyDeviceToHost);
      fprintf(stderr, "[smp] cycles of workgroup 0: load %.0f proposal %.0f lanes %.0f node updates %.0f decision %.0f store %.0f\n",
              ph[6], ph[1], ph[2], ph[3], ph[4], ph[7]);
      if (mode == 0)
      {
        double pr[12]; (void)hipMemcpy(pr, s->mix_delta.p + 16, sizeof pr, hipMemcpyDeviceToHost);
        fprintf(stderr, "[smp] leader of locus 0, all proposals: gspr links %.0f subtree/pop0 %.0f bounds %.0f scan %.0f surgery %.0f masks %.0f density %.0f install %.0f | gage pre %.0f density %.0f install %.0f\n",
                pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7], pr[8], pr[9], pr[10]);
      }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    s->launches++;
    return 1;
  }
  if (s->timing_stride && (mode == 0 || mode == 1 || mode == 4) && (s->timing_phase++ % s->timing_stride) == 0)
  {
    if (s->timed.size() >= 4096 && !sampler_timing_drain(s)) return 0;
    bpa_sampler::Timed t{nullptr, nullptr, mode == 0 ? 0 : 1};
    HIPCHK(hipEventCreate(&t.e0)); HIPCHK(hipEventCreate(&t.e1));
    hipExtLaunchKernelGGL(kern, dim3(s->nblocks), dim3(smp::BS), lds, e->stream, t.e0, t.e1, 0, a);
    s->timed.push_back(t);
  }
  else
    hipLaunchKernelGGL(kern, dim3(s->nblocks), dim3(smp::BS), lds, e->stream, a);
  HIPCHK(hipGetLastError());
  s->launches++;
  return 1;
}

extern "C" int bpa_sampler_set_species_tree(bpa_sampler_t * s, int species, const int * parent, const double * tau,
                                            const double * theta)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) { const int ok = comp_invalidate(s) && comp_each(s, [&](bpa_sampler * p) { return bpa_sampler_set_species_tree(p, species, parent, tau, theta); }); if (ok) s->sp = comp_part0(s)->sp; return ok; }
  const int np = 2*species - 1;
  if (species < 1 || species > smp::MAXTIPS) return fail("bpa_sampler_set_species_tree: 1..8 species");
  if (!sampler_invalidate(s)) return 0;           // (the trees keep their state; taus and thetas are replaced below)
  smp::Species & sp = s->sp;
  for (int p = 0; p < np; ++p)
  {
    const bool okp = p == np - 1 ? parent[p] == -1 : (parent[p] > p && parent[p] < np && parent[p] >= species);
    const bool okt = theta[p] > 0 && (p < species ? tau[p] == 0 : tau[p] > 0) && (parent[p] < 0 || !okp || tau[parent[p]] > tau[p]);
    if (!okp || !okt) return fail("bpa_sampler_set_species_tree: populations must come tips first, children before parents, "
                                  "with theta > 0 and tau increasing towards the root");
  }
  sp.S = species; sp.npop = np;
  for (int p = 0; p < smp::MAXPOP; ++p) { sp.parent[p] = sp.left[p] = sp.right[p] = -1; sp.anc[p] = 0; }
  s->h_taus.assign(3*smp::MAXPOP, 0.0);
  for (int p = 0; p < smp::MAXPOP; ++p) s->h_taus[smp::MAXPOP + p] = 1.0;
  for (int p = 0; p < np; ++p)
  {
    sp.parent[p] = (int8_t)parent[p];
    s->h_taus[p] = tau[p];
    s->h_taus[smp::MAXPOP + p] = theta[p];
    s->h_taus[2*smp::MAXPOP + p] = std::log(2.0/(1.0*theta[p]));      // the host driver's libm value (a00_msc_contrib)
  }
  for (int p = 0; p < np - 1; ++p)
  {
    const int q = parent[p];
    if (sp.left[q] < 0) sp.left[q] = (int8_t)p; else if (sp.right[q] < 0) sp.right[q] = (int8_t)p;
    else return fail("bpa_sampler_set_species_tree: a population has more than two children");
  }
  for (int p = species; p < np; ++p) if (sp.right[p] < 0) return fail("bpa_sampler_set_species_tree: an inner population needs two children");
  for (int p = 0; p < np; ++p) for (int q = p; q >= 0; q = parent[q]) sp.anc[p] |= (uint16_t)(1u << q);
  s->uploaded = false; s->host_current = false;
  return 1;
}

extern "C" int bpa_sampler_set_tip_species(bpa_sampler_t * s, unsigned i, const int * species)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) { unsigned j; bpa_sampler * p = comp_part(s, i, &j); if (!p) return fail("bpa_sampler_set_tip_species: locus index out of range"); return comp_invalidate(s) && bpa_sampler_set_tip_species(p, j, species); }
  if (i >= s->nloci) return fail("bpa_sampler_set_tip_species: locus index out of range");
  if (!sampler_invalidate(s)) return 0;
  if (s->big)
  {
    gbig::BTree & t = s->b_trees[i];
    if (!t.tips) return fail("bpa_sampler_set_tip_species: set the tree first");
    for (int k = 0; k < t.tips; ++k) t.pop[k] = (int16_t)species[k];
    return 1;
  }
  if (s->generic)
  {
    gsm::GTree & t = s->g_trees[i];
    if (!t.tips) return fail("bpa_sampler_set_tip_species: set the tree first");
    for (int k = 0; k < t.tips; ++k) t.pop[k] = (int8_t)species[k];
    return 1;
  }
  smp::Tree & t = s->h_trees[i];
  if (!t.tips) return fail("bpa_sampler_set_tip_species: set the tree first");
  for (int k = 0; k < t.tips; ++k) t.pop[k] = (int8_t)species[k];
  return 1;
}

extern "C" void bpa_sampler_set_finetune(bpa_sampler_t * s, double gage, double gspr, double tau, double mix)
{
  if (s->comp) (void)comp_each(s, [&](bpa_sampler * p) { bpa_sampler_set_finetune(p, gage, gspr, tau, mix); return 1; });
  s->sp.ft_gage = gage; s->sp.ft_gspr = gspr; s->sp.ft_tau = tau; s->sp.ft_mix = mix;
}

// ---- the burn-in's step-length rule (reset_finetune_onestep, method.c:1122-1136; called by reset_finetune, method.c:1508-1516,
// four times during the burn-in and once at its end: method.c:5364-5377) on the persistent kernel's move-type counters
static double finetune_onestep(const double pjump, const double ft)
{
  const double maxstep = 99, optimum = 0.3, half_pi = 1.5707963267948966;          // (pj_optimum, method.c:45)
  if (pjump < 0.001) return ft/100;
  if (pjump > 0.999) return std::min(maxstep, ft*100);
  return std::min(maxstep, ft*std::tan(half_pi*pjump)/std::tan(half_pi*optimum));
}

extern "C" double bpa_finetune_onestep(double pjump, double finetune) { return finetune_onestep(pjump, finetune); }

// several ranks: the program's rule sees the acceptance proportions over ALL loci (one finetune per move for the whole data
// set), so the per-locus moves' counts are pooled over the ranks first — through the callback, or through the mailboxes' one-shot
// exchange when the sums live inside the persistent kernel.  Every rank calls bpa_sampler_adapt_finetune at the same point.
static int sampler_pool_counts(bpa_sampler * s, unsigned long long * c, unsigned n)
{
  if ((!s->allreduce && !s->p2p) || !n) return 1;
  // (several ranks: a rank that went on with its own counts would end at step lengths of its own)
  if (n > (unsigned)smp::MAXPOP || !s->theta_sums.p) return fail("bpa_sampler_adapt_finetune: the ranks' counts cannot be pooled (no exchange buffer)");
  bpa_engine * e = s->eng;
  double v[smp::MAXPOP];
  for (unsigned i = 0; i < n; ++i) v[i] = (double)c[i];
  double * ar = (s->allreduce && s->sum_ext) ? s->sum_ext : s->theta_sums.p;
  HIPCHK(hipMemcpyAsync(ar, v, n*sizeof(double), hipMemcpyHostToDevice, e->stream));
  if (s->allreduce) { if (!s->allreduce(s->allreduce_ctx, ar, n, (void *)e->stream)) return fail("bpa_sampler: the all-reduce callback failed"); }
  else if (!bpa_p2p_allreduce(s->p2p, ar, n)) return 0;
  HIPCHK(hipMemcpyAsync(v, ar, n*sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (unsigned i = 0; i < n; ++i) c[i] = (unsigned long long)(v[i] + 0.5);
  return 1;
}

extern "C" int bpa_sampler_adapt_finetune(bpa_sampler_t * s, double * pjump, double * finetune)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp || s->big) return fail("bpa_sampler_adapt_finetune: the move-type counters are the persistent iteration kernel's and the generic sampler's (with the program's moves)");
  if (!sampler_download(s)) return 0;                 // (settles the launches in flight; the counters are then final)
  if (s->generic)
  {
    if (!(s->kernel_bpp && s->sp.program_moves)) return fail("bpa_sampler_adapt_finetune: on a generic sampler the step-length rule runs with the program's moves (bpa_sampler_set_program_moves)");
    unsigned long long tot[4] = {0, 0, 0, 0};          // gage proposed / accepted, gspr proposed / accepted over all loci
    for (const auto & t : s->g_trees) { tot[0] += t.pj_gage; tot[1] += t.pj_gage_acc; tot[2] += t.pj_gspr; tot[3] += t.pj_gspr_acc; }
    unsigned long long c[10];
    for (int k = 0; k < 4; ++k) { if (tot[k] < s->gp_pj_base[k]) s->gp_pj_base[k] = 0;      /* (the trees were set again: their counts start over) */
                                  c[k] = tot[k] - s->gp_pj_base[k]; s->gp_pj_base[k] = tot[k]; }
    for (int k = 4; k < 10; ++k) { c[k] = s->gp_pj[k]; s->gp_pj[k] = 0; }
    if (!sampler_pool_counts(s, c, 4)) return 0;          // (the all-loci moves' counts are every rank's own copy of the same decisions)
    double * ft[5] = { &s->sp.ft_gage, &s->sp.ft_gspr, &s->sp.ft_tau, &s->sp.ft_mix, &s->sp.ft_theta };
    for (int m = 0; m < 5; ++m)
    {
      const double pj = c[2*m] ? (double)c[2*m + 1]/(double)c[2*m] : -1.0;
      if (pj >= 0 && *ft[m] > 0) *ft[m] = finetune_onestep(pj, *ft[m]);
      if (pjump) pjump[m] = pj;
      if (finetune) finetune[m] = *ft[m];
    }
    return 1;
  }
  if (!s->v2_ok || !s->v2_pj.p) return fail("bpa_sampler_adapt_finetune: the sampler does not run the persistent iteration kernel");
  // the all-reduce callback's form of several ranks runs the all-loci steps as launches of their own (smp::decide_kernel), which keep
  // no counts by move type: the rule would silently leave the tau / mixing / theta windows where they are
  if (s->allreduce && !s->p2p) return fail("bpa_sampler_adapt_finetune: with an all-reduce callback the persistent kernel's all-loci steps are launches of their own, which do not count proposals by move type — use the mailboxes (bpa_sampler_set_p2p) or the generic sampler");
  unsigned long long c[16];
  HIPCHK(hipMemcpy(c, s->v2_pj.p, sizeof c, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(s->v2_pj.p, 0, sizeof c));          // pjump_reset (method.c:5377)
  if (!sampler_pool_counts(s, c, 4)) return 0;
  double * ft[5] = { &s->sp.ft_gage, &s->sp.ft_gspr, &s->sp.ft_tau, &s->sp.ft_mix, &s->sp.ft_theta };
  for (int m = 0; m < 5; ++m)
  {
    const double pj = c[2*m] ? (double)c[2*m + 1]/(double)c[2*m] : -1.0;
    // (a move that was never proposed keeps its step length: the program has no such case — every active one runs every iteration;
    //  the theta window is proposed once in ten, opt_theta_slide_prob)
    if (pj >= 0 && *ft[m] > 0) *ft[m] = finetune_onestep(pj, *ft[m]);
    if (pjump) pjump[m] = pj;
    if (finetune) finetune[m] = *ft[m];
  }
  return 1;                                            // (the species record goes to the device with the next launch: v2_sp_sent)
}

// the program's burn-in: `iterations` iterations with the step lengths reset from the acceptance proportions exactly where the
// program's loop resets them (method.c:5364-5417): its counter i runs from -burnin, a reset happens at the TOP of iteration i
// when i % (burnin/4) == 0 and at least 100 iterations have run since the last one (ft_round, method.c:5417), and once more at
// i == 0 — burnin 400: after 100 / 200 / 300 / 400 iterations; 300: after 150 / 300; 402: after 102 / 202 / 302 / 402; below 200:
// the program resets nothing (opt_burnin >= 200), nor does this.  bpa_burnin_schedule lists the points (tests/test_finetune_adaptation.py).
extern "C" unsigned bpa_burnin_schedule(unsigned burnin, unsigned * after, unsigned cap)
{
  unsigned n = 0;
  if (burnin < 200) return 0;
  const long q = (long)(burnin/4);
  long ft_round = 0;
  for (long i = -(long)burnin; i < 0; ++i)
  {
    if (ft_round >= 100 && i % q == 0) { if (after && n < cap) after[n] = (unsigned)(i + (long)burnin); ++n; ft_round = 0; }
    ++ft_round;
  }
  if (after && n < cap) after[n] = burnin;
  return n + 1;
}

extern "C" int bpa_sampler_burnin(bpa_sampler_t * s, unsigned iterations, double * finetune)
{
  // what the rule needs is checked BEFORE the chain moves (a sampler that cannot adapt must not be left half-way through)
  if (iterations >= 200)
  {
    if (s->comp || s->big) return fail("bpa_sampler_burnin: the step-length rule runs on the persistent iteration kernel and on the generic sampler with the program's moves (not on loci of several kinds or of more than 16 tips)");
    if (s->generic && !(s->kernel_bpp && s->sp.program_moves)) return fail("bpa_sampler_burnin: on a generic sampler the step-length rule runs with the program's moves (bpa_sampler_set_program_moves)");
    if (!s->generic)
    {
      std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
      if (!set_device(s->eng) || !sampler_upload(s)) return 0;           // (which kernel runs the loci is settled at upload)
      if (!s->v2_ok || !s->v2_pj.p) return fail("bpa_sampler_burnin: the sampler does not run the persistent iteration kernel (its move-type counters feed the rule)");
      if (s->allreduce && !s->p2p) return fail("bpa_sampler_burnin: with an all-reduce callback the persistent kernel's all-loci steps keep no counts by move type — use the mailboxes (bpa_sampler_set_p2p) or the generic sampler");
    }
  }
  unsigned pts[16];
  const unsigned npts = bpa_burnin_schedule(iterations, pts, 16);
  unsigned done = 0;
  for (unsigned r = 0; r < npts && r < 16u; ++r)
  {
    if (pts[r] > done && !bpa_sampler_iterate(s, pts[r] - done)) return 0;
    done = pts[r];
    if (!bpa_sampler_adapt_finetune(s, nullptr, r + 1 == npts ? finetune : nullptr)) return 0;
  }
  if (iterations > done && !bpa_sampler_iterate(s, iterations - done)) return 0;
  if (!npts && finetune) { finetune[0] = s->sp.ft_gage; finetune[1] = s->sp.ft_gspr; finetune[2] = s->sp.ft_tau; finetune[3] = s->sp.ft_mix; finetune[4] = s->sp.ft_theta; }
  return 1;
}

extern "C" void bpa_sampler_set_tau_prior(bpa_sampler_t * s, double alpha, double beta)
{
  if (s->comp) (void)comp_each(s, [&](bpa_sampler * p) { bpa_sampler_set_tau_prior(p, alpha, beta); return 1; });
  s->sp.tau_alpha = alpha; s->sp.tau_beta = beta;
}

extern "C" void bpa_sampler_set_theta_prior(bpa_sampler_t * s, double alpha, double beta, double finetune)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  // (a composite: whether there is a THETA step at all is part of what the parts agree on at upload)
  if (s->comp) { const bool had = s->sp.theta_alpha > 0; if (had != (alpha > 0)) (void)comp_invalidate(s); (void)comp_each(s, [&](bpa_sampler * p) { bpa_sampler_set_theta_prior(p, alpha, beta, finetune); return 1; }); }
  s->sp.theta_alpha = alpha; s->sp.theta_beta = beta; s->sp.ft_theta = finetune;
  if (s->uploaded && s->v2_ok && v2_wants_prog(s) != s->v2_prog) (void)sampler_invalidate(s);      // (the kernel's form is chosen at upload)
}

extern "C" int bpa_sampler_get_thetas(bpa_sampler_t * s, double * theta)
{
  bpa_engine * e = s->eng;
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (s->comp) return comp_upload(s) && bpa_sampler_get_thetas(comp_part0(s), theta);
  if (!set_device(e)) return 0;
  const size_t np = (size_t)s->sp.npop;
  if (!s->uploaded) { for (size_t i = 0; i < np; ++i) theta[i] = s->h_taus[smp::MAXPOP + i]; return (int)np; }
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(theta, s->taus.p + smp::MAXPOP, np*sizeof(double), hipMemcpyDeviceToHost));
  return (int)np;
}

extern "C" int bpa_sampler_get_taus(bpa_sampler_t * s, double * taus)
{
  bpa_engine * e = s->eng;
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (s->comp) return comp_upload(s) && bpa_sampler_get_taus(comp_part0(s), taus);
  if (!set_device(e)) return 0;
  const size_t np = (size_t)s->sp.npop;
  if (!s->uploaded) { for (size_t i = 0; i < np; ++i) taus[i] = s->h_taus[i]; return (int)np; }
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(taus, s->taus.p, np*sizeof(double), hipMemcpyDeviceToHost));
  return (int)np;
}

extern "C" int bpa_sampler_initialize(bpa_sampler_t * s)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) return comp_run(s, 0, 0);
  if (!sampler_upload(s)) return 0;
  s->host_current = false;
  if (s->big) return gb_initialize(s);
  if (s->generic) return gs_initialize(s);
  return sampler_launch(s, 3, 1.0);
}

// the all-loci steps' acceptance term: summed over this rank's loci on the device, then over the ranks
static int sampler_sum(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  double * out = s->sum_ext ? s->sum_ext : s->mix_sum.p;
  hipLaunchKernelGGL(lnl_sum_kernel, dim3(1), dim3(1024), 0, e->stream, s->wg_part, s->nblocks, out);
  HIPCHK(hipGetLastError());
  if (s->allreduce && !s->allreduce(s->allreduce_ctx, out, 1u, (void *)e->stream)) return fail("bpa_sampler: the all-reduce callback failed");
  return 1;
}

// sum this step's per-locus terms (over the ranks too when an all-reduce is installed) and decide
static int sampler_decide(bpa_sampler * s, double uacc, int tau_q, int theta_p, double win_u, double mix_c, double mix_lnc)
{
  bpa_engine * e = s->eng;
  s->epoch++;
  if (!s->allreduce)
    hipLaunchKernelGGL(smp::sum_decide_kernel, dim3(1), dim3(1024), 0, e->stream, s->wg_part, s->nblocks, uacc, s->epoch,
                       s->flag.p, s->counters.p, s->taus.p, s->sp, tau_q, theta_p, win_u, mix_c, mix_lnc);
  else
  {
    if (!sampler_sum(s)) return 0;
    hipLaunchKernelGGL(smp::decide_kernel, dim3(1), dim3(1), 0, e->stream, s->sum_ext ? s->sum_ext : s->mix_sum.p, uacc, s->epoch,
                       s->flag.p, s->counters.p, s->taus.p, s->sp, tau_q, theta_p, win_u, mix_c, mix_lnc);
  }
  HIPCHK(hipGetLastError());
  s->launches += s->allreduce ? 2 : 1;
  s->mix_pending = true;
  return 1;
}

extern "C" int bpa_sampler_set_allreduce(bpa_sampler_t * s, bpa_allreduce_fn fn, void * ctx, double * device_sum,
                                         unsigned first_locus)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) return comp_set_allreduce(s, fn, ctx, device_sum, first_locus);
  s->allreduce = fn; s->allreduce_ctx = ctx; s->sum_ext = device_sum;
  if (first_locus != s->locus_offset)
  {
    if (!sampler_invalidate(s)) return 0;
    s->locus_offset = first_locus;
    for (unsigned i = 0; i < s->nloci; ++i)
      (s->big ? s->b_trees[i].rng : s->generic ? s->g_trees[i].rng : s->h_trees[i].rng) = stream_seed(s, first_locus + i);
    s->uploaded = false;
  }
  return 1;
}

// the whole iteration(s) as persistent launches: GAGE + GSPR of every locus, THETA, TAU per divergence, MIX — state in
// LDS from the first proposal to the last decision (sweep2.hpp)
// in_kernel_allloci = false (several ranks): only the per-locus sweep of ONE iteration; the all-loci steps then run as
// launches of sampler.hpp's kernels with the sums all-reduced in between
static int sampler_iterate_v2(bpa_sampler * s, unsigned iterations, bool in_kernel_allloci)
{
  bpa_engine * e = s->eng;
  const int npop = s->sp.npop, S = s->sp.S;
  uint32_t theta_mask = 0;
  if (s->sp.theta_alpha > 0) for (int p = 0; p < npop; ++p) if (s->has_theta[p]) theta_mask |= 1u << p;
  const bool allloci = in_kernel_allloci && !s->env_nomix;
  // exchanges of an iteration: THETA in blocks of 15 values, one per TAU, one for MIX (sweep2.hpp: exchange)
  const bool program = s->v2_prog;                                 // (k, T) per theta instead of one difference per population
  const unsigned XV = (unsigned)smp2::XV;
  // (the program's moves: the first TAU's five sums ride on the THETA step's exchange, sweep2.hpp)
  const bool merged = program;
  const unsigned x_theta = !theta_mask ? 0u : program ? (2u*(unsigned)__builtin_popcount(theta_mask) + (merged ? 5u : 0u) + XV - 1u)/XV : ((unsigned)npop + XV - 1u)/XV;
  const unsigned x_per_iter = allloci ? x_theta + (unsigned)(npop - S) - (merged ? 1u : 0u) + 1u : 0u;
  const unsigned draws_per_iter = allloci ? 2u*(unsigned)__builtin_popcount(theta_mask) + 2u*(unsigned)(npop - S) + 2u : 0u;
  void (*kern)(const smp2::Args) = s->v2_prog   ? (s->v2_nt == 4 ? smp2::iter_kernel<4, true, true> : smp2::iter_kernel<8, true, true>)
                                 : s->kernel_bpp ? (s->v2_nt == 4 ? smp2::iter_kernel<4, true> : smp2::iter_kernel<8, true>)
                                                 : (s->v2_nt == 4 ? smp2::iter_kernel<4, false> : smp2::iter_kernel<8, false>);
  const unsigned bs = s->v2_nt == 4 ? smp2::Cfg<4>::BS : smp2::Cfg<8>::BS;
  while (iterations)
  {
    const unsigned chunk = std::min(iterations, 4096u);          // (bounds one launch's duration)
    smp2::Args a{};
    a.wave_off = s->v2_wave_off.p; a.loc = s->v2_loc.p; a.pat = s->v2_pat.p; a.trees = s->trees.p; a.taus = s->taus.p;
    if (s->v2_log.size() >= 65536u && !sampler_download(s)) return 0;            // (bounds the log of a caller that never looks)
    s->v2_log.push_back({s->grng, chunk, s->mix_pending, s->logpr_stale});
    // what the one-launch-per-step path left pending is settled while this launch loads (Args::snap ...)
    a.snap = s->snap.p; a.mix_flag = s->flag.p; a.epoch = s->mix_pending ? s->epoch : 0u; a.refresh_logpr = s->logpr_stale ? 1u : 0u;
    s->mix_pending = false; s->logpr_stale = false;
    a.counters = s->counters.p; a.lograt = s->lograt.p; a.pop_nc = s->pop_nc.p; a.pop_t2h = s->pop_t2h.p;
    a.ntasks = s->nloci; a.nwaves = s->v2_nwaves; a.nwg = s->v2_nwg; a.lwaves = s->v2_lwaves; a.xbuf = s->v2_xbuf.p;
    a.err = s->v2_err.p; a.grng = s->v2_grng.p; a.niter = chunk; a.pj = s->v2_pj.p;
    a.nsteps_gage = s->maxtips - 1; a.nsteps_gspr = 2*s->maxtips - 2;
    if (s->env_gage >= 0) { a.nsteps_gage = (uint32_t)s->env_gage; a.nsteps_gspr = (uint32_t)s->env_gspr; }
    a.theta_mask = theta_mask; a.do_allloci = allloci ? 1u : 0u; a.dbg = s->env_dbg;
    if (allloci && ++s->v2_launch_no == s->env_inject) a.dbg |= s->env_inject_bit;
    if (s->p2p && allloci)
    {
      a.peers = s->p2p->d_peer.p; a.mail = s->p2p->mail; a.rank = s->p2p->rank; a.world = s->p2p->world; a.slot_bytes = s->p2p->slot_bytes;
      a.seq0 = s->p2p->seq; a.spin_limit = s->p2p->spin_limit; a.p2p_err = s->p2p->d_err.p;
      s->p2p->seq += (unsigned long long)chunk*x_per_iter;
    }
    else { a.world = 1; a.rank = 0; }
    a.bfbeta = e->usedata ? e->bfbeta : 0.0; a.prof = s->v2_prof.p; a.declog = s->v2_declog.p; a.sp = s->v2_sp.p;
    if (std::memcmp(&s->v2_sp_sent, &s->sp, sizeof(smp::Species)) != 0)          // (finetune / prior setters change it between calls)
    {
      HIPCHK(hipMemcpyAsync(s->v2_sp.p, &s->sp, sizeof(smp::Species), hipMemcpyHostToDevice, e->stream));
      s->v2_sp_sent = s->sp;
    }
    if (s->env_dbg & 256u) HIPCHK(hipMemsetAsync(s->v2_declog.p, 0, 4*2048*sizeof(double), e->stream));
    if (allloci)
    {
      // (a sweep-only launch draws nothing from the global stream and exchanges nothing)
      // (BPP's kernel draws a data-dependent count from the global stream — an acceptance number only when needed —, so
      //  the device's word is the master after the first launch; our kernel's fixed count is mirrored on the host below)
      if (!s->kernel_bpp || !s->v2_grng_sent)
        HIPCHK(hipMemcpyAsync(s->v2_grng.p, &s->grng, sizeof(a00_rng_t), hipMemcpyHostToDevice, e->stream));
      s->v2_grng_sent = true;
      HIPCHK(hipMemsetAsync(s->v2_xbuf.p, 0, (size_t)2*smp2::XN*sizeof(unsigned long long), e->stream));
    }
    if (s->timing_stride && (s->timing_phase++ % s->timing_stride) == 0)
    {
      if (s->timed.size() >= 4096 && !sampler_timing_drain(s)) return 0;
      bpa_sampler::Timed t{nullptr, nullptr, 0};
      HIPCHK(hipEventCreate(&t.e0)); HIPCHK(hipEventCreate(&t.e1));
      hipExtLaunchKernelGGL(kern, dim3(s->v2_nwg), dim3(bs), s->v2_lds, e->stream, t.e0, t.e1, 0, a);
      s->timed.push_back(t);
    }
    else
      hipLaunchKernelGGL(kern, dim3(s->v2_nwg), dim3(bs), s->v2_lds, e->stream, a);
    HIPCHK(hipGetLastError());
    // the host's copy of the global stream follows the kernel's draws
    if (!s->kernel_bpp) for (unsigned long k = 0; k < (unsigned long)chunk*draws_per_iter; ++k) (void)a00_rndu(&s->grng);
    s->launches++; s->sweeps += chunk; s->v2_iters += chunk;
    iterations -= chunk;
  }
  return 1;
}

extern "C" int bpa_sampler_set_p2p(bpa_sampler_t * s, bpa_p2p_t * p, unsigned first_locus)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  COMP_FAIL("bpa_sampler_set_p2p: loci of several kinds exchange through bpa_sampler_set_allreduce (the mailboxes are read by the persistent kernel's single launch)");
  if (p && (!p->connected || p->eng != s->eng)) return fail("bpa_sampler_set_p2p: connect the exchange first (same engine)");
  if (p && p->nmax < 16u) return fail("bpa_sampler_set_p2p: the mailboxes must hold at least 16 values");
  // only the persistent kernel reads the mailboxes: a generic or big-tree sampler would decide from its own shard's sums
  if (p && (s->generic || s->big)) return fail("bpa_sampler_set_p2p: the in-kernel exchange needs the persistent kernel (JC69 loci of <= 8 tips, <= 64 patterns); use bpa_sampler_set_allreduce");
  if (!sampler_invalidate(s)) return 0;
  s->p2p = p;
  if (first_locus != s->locus_offset)
  {
    s->locus_offset = first_locus;
    for (unsigned i = 0; i < s->nloci; ++i)
    {
      const a00_rng_t r = stream_seed(s, first_locus + i);
      if (s->big) s->b_trees[i].rng = r; else if (s->generic) s->g_trees[i].rng = r; else s->h_trees[i].rng = r;
    }
  }
  return 1;
}

extern "C" int bpa_sampler_iterate(bpa_sampler_t * s, unsigned iterations)
{
  bpa_engine * e = s->eng;
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (s->comp) return comp_run(s, 1, iterations);
  if (!sampler_upload(s)) return 0;
  s->host_current = false;
  if (s->big) return gb_iterate(s, iterations);
  if (s->generic) return gs_iterate(s, iterations);
  if (s->v2_ok && !s->allreduce && !s->env_trace) return sampler_iterate_v2(s, iterations, true);
  if (s->kernel_bpp) return fail("bpa_sampler: BPP's proposal kernel runs in the persistent iteration kernel only (one GPU, or several with bpa_sampler_set_p2p)");
  // several ranks: the per-locus sweep by the persistent kernel (one launch), the all-loci steps one launch each
  const bool v2_sweep = s->v2_ok && !s->env_trace;
  const bool fused = s->fuse_decision && !s->allreduce && !s->env_trace;      // (several GPUs: sum -> all-reduce -> decide)
  for (unsigned it = 0; it < iterations; ++it)
  {
    if (v2_sweep) { if (!sampler_iterate_v2(s, 1, false)) return 0; }
    else
    {
      if (!sampler_launch(s, 0, 1.0)) return 0;                  // GAGE + GSPR of every locus (settles a pending mix first)
      s->sweeps++;
    }
    if (s->env_nomix) continue;
    if (s->sp.theta_alpha > 0)
    {
      // THETA for every population that can hold a coalescence, from the statistics the sweep just stored: the sums and
      // the (independent) decisions in one launch, then every locus's density with the new thetas
      smp::ThetaArgs ta{};
      for (int p = 0; p < s->sp.npop; ++p)
        if (s->has_theta[p]) { ta.on[p] = 1u; ta.win_u[p] = a00_rndu(&s->grng); ta.uacc[p] = a00_rndu(&s->grng); }
      if (!s->allreduce)
        hipLaunchKernelGGL(smp::theta_sum_decide_kernel, dim3(s->sp.npop), dim3(1024), 0, e->stream, s->pop_nc.p, s->pop_t2h.p,
                           s->nloci, s->taus.p, s->sp, ta, s->counters.p, (double *)nullptr, (const double *)nullptr, 1);
      else
      {
        hipLaunchKernelGGL(smp::theta_sum_decide_kernel, dim3(s->sp.npop), dim3(1024), 0, e->stream, s->pop_nc.p, s->pop_t2h.p,
                           s->nloci, s->taus.p, s->sp, ta, s->counters.p, s->theta_sums.p, (const double *)nullptr, 0);
        // all populations' sums in ONE collective, in the memory the callback's collective addresses
        double * ar = s->sum_ext ? s->sum_ext : s->theta_sums.p;
        const size_t nb = (size_t)s->sp.npop*sizeof(double);
        if (ar != s->theta_sums.p) HIPCHK(hipMemcpyAsync(ar, s->theta_sums.p, nb, hipMemcpyDeviceToDevice, e->stream));
        if (!s->allreduce(s->allreduce_ctx, ar, (unsigned)s->sp.npop, (void *)e->stream)) return fail("bpa_sampler: the all-reduce callback failed");
        if (ar != s->theta_sums.p) HIPCHK(hipMemcpyAsync(s->theta_sums.p, ar, nb, hipMemcpyDeviceToDevice, e->stream));
        hipLaunchKernelGGL(smp::theta_sum_decide_kernel, dim3(s->sp.npop), dim3(1024), 0, e->stream, s->pop_nc.p, s->pop_t2h.p,
                           s->nloci, s->taus.p, s->sp, ta, s->counters.p, (double *)nullptr, (const double *)s->theta_sums.p, 1);
      }
      s->logpr_stale = true;        // the next launch (any mode) recomputes every tree's density with the new thetas
      HIPCHK(hipGetLastError());
      s->launches += 1;
    }
    for (int q = s->sp.S; q < s->sp.npop; ++q)                    // one rubber-band step per species divergence
    {
      const double uprop = a00_rndu(&s->grng), uacc_t = a00_rndu(&s->grng);
      if (fused)
      {
        // the step's launch decides for itself (its last workgroup): no kernel of its own for the decision
        if (!sampler_launch(s, 4, 1.0, 0.0, (unsigned)q, uprop, uacc_t)) return 0;
        s->epoch++; s->mix_pending = true;
        continue;
      }
      if (!sampler_launch(s, 4, 1.0, 0.0, (unsigned)q, uprop)) return 0;
      if (!sampler_decide(s, uacc_t, q, -1, uprop, 1.0, 0.0)) return 0;
    }
    const double lnc = s->sp.ft_mix*(a00_rndu(&s->grng) - 0.5), c = std::exp(lnc);
    const double uacc = a00_rndu(&s->grng);
    if (fused)
    {
      if (!sampler_launch(s, 1, c, lnc, 0, 0.0, uacc)) return 0;
      s->epoch++; s->mix_pending = true;
      continue;
    }
    if (!sampler_launch(s, 1, c, lnc)) return 0;                 // mixing proposal of every locus
    if (!sampler_decide(s, uacc, -1, -1, 0.0, c, lnc)) return 0;
  }
  return 1;
}

// settle a pending mixing decision and bring the trees to the host
static int sampler_download(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  if (!sampler_upload(s)) return 0;
  if (s->host_current) return 1;                  // nothing has run since the last download
  if (s->big) { if (!gb_download(s)) return 0; s->host_current = true; return 1; }
  if (s->generic) { if (!gs_download(s)) return 0; s->host_current = true; return 1; }
  if (!sampler_launch(s, 2, 1.0)) return 0;
  HIPCHK(hipMemcpyAsync(s->h_trees.data(), s->trees.p, s->nloci*sizeof(smp::Tree), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  int err = 0;
  HIPCHK(hipMemcpy(&err, s->dec_err, sizeof err, hipMemcpyDeviceToHost));
  if (err) return fail("bpa_sampler: an in-launch decision timed out waiting for a workgroup's sum (BPA_SMP_FUSE is set: unset it)");
  if (s->v2_ok)
  {
    if (s->env_dbg & 16u)
    {
      double pr[16]; HIPCHK(hipMemcpy(pr, s->v2_prof.p, sizeof pr, hipMemcpyDeviceToHost));
      fprintf(stderr, "[smp2] cycles of lane 0 of workgroup 0, last launch: propose %.0f evaluate %.0f decide %.0f theta %.0f tau %.0f mix %.0f | exchange: theta %.0f tau+mix %.0f | inside the exchanges: first barrier %.0f sums %.0f barrier %.0f arrival + poll %.0f last barrier %.0f | of theta / tau / mix: the decision's arithmetic after the totals %.0f %.0f, MIX's re-draws inside the wait %.0f\n",
              pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7], pr[8], pr[9], pr[10], pr[11], pr[12], pr[13], pr[14], pr[15]);
      double p2[24]; HIPCHK(hipMemcpy(p2, s->v2_prof.p + 16 + s->v2_nwg, sizeof p2, hipMemcpyDeviceToHost));
      fprintf(stderr, "[smp2] sweep cycles of the waves of workgroup 0:");
      for (int w = 0; w < 8; ++w) fprintf(stderr, " %.0f", p2[8 + w]);
      fprintf(stderr, "\n");
      if (s->v2_prog)
        fprintf(stderr, "[smp2] (program-moves kernel: the numbers above are the control wave's — before B1 | wait for the loci | push | poll | THETA decision | TAU decisions | MIX decision | next proposal)\n");
    }
    if (s->env_dbg & 256u)
    {
      std::vector<double> r(4*2048); HIPCHK(hipMemcpy(r.data(), s->v2_declog.p, r.size()*sizeof(double), hipMemcpyDeviceToHost));
      for (unsigned k = 1000; k < 2048; ++k) if (r[4*k] != 0) fprintf(stderr, "[smp2dbg] %u: %g %.17g %.17g %.17g\n", k, r[4*k], r[4*k+1], r[4*k+2], r[4*k+3]);
      for (unsigned k = 0; k < 1000 && r[4*k] != 0; ++k)
        fprintf(stderr, "[smp2] %s %d lnacc %.17g u %.17g -> %d\n", r[4*k] >= 300 ? "mix" : r[4*k] >= 200 ? "tau" : "theta",
                (int)r[4*k] % 100, r[4*k + 1], r[4*k + 2], (int)r[4*k + 3]);
    }
    if (s->env_dbg & 32u)
    {
      std::vector<double> w(s->v2_nwg); HIPCHK(hipMemcpy(w.data(), s->v2_prof.p + 16, w.size()*sizeof(double), hipMemcpyDeviceToHost));
      double mn = 1e300, mx = 0, sm = 0; unsigned imx = 0;
      for (unsigned i = 0; i < w.size(); ++i) { mn = std::min(mn, w[i]); if (w[i] > mx) { mx = w[i]; imx = i; } sm += w[i]; }
      fprintf(stderr, "[smp2] sweep cycles per workgroup, last launch: min %.0f mean %.0f max %.0f (workgroup %u of %u)\n", mn, sm/w.size(), mx, imx, s->v2_nwg);
    }
    int verr[3] = {0, 0, 0};
    HIPCHK(hipMemcpy(verr, s->v2_err.p, sizeof verr, hipMemcpyDeviceToHost));
    err = verr[0];
    if (!err && s->p2p) { HIPCHK(hipMemcpy(&err, s->p2p->d_err.p, sizeof err, hipMemcpyDeviceToHost)); if (err) return fail("bpa_sampler: the exchange between the GPUs timed out inside the persistent kernel (a rank is missing or slow; install an all-reduce callback instead)"); }
    if (err)
    {
      // A launch whose workgroups were not all resident together (the device is shared: another process, a partitioned GPU) gave
      // up after 0.5 s and left every locus as it found it; verr[1] = the iterations such launches did not run.  They are run
      // again: up to three times by the persistent kernel (the co-tenant may be gone), then — our own kernel's moves — by the
      // one-launch-per-step path, which needs no co-residency; this sampler keeps to that path from then on.
      const int zero2[2] = {0, 0};
      HIPCHK(hipMemcpy(s->v2_err.p, zero2, sizeof zero2, hipMemcpyHostToDevice));
      if (s->p2p || verr[1] <= 0) return fail("bpa_sampler: the persistent iteration kernel timed out waiting for a workgroup's sum (is the device shared? BPA_SMP_V1=1 selects one launch per step)");
      fprintf(stderr, "[bpp_amd] the persistent iteration kernel timed out waiting for a workgroup (is the device shared?): %d iterations are run again\n", verr[1]);
      // The launches that did not run are the LAST ones of the log (a launch that finds the error word set gives up at once,
      // sweep2.hpp): they are taken off the counters and the host's copy of the global stream goes back to where the first of
      // them started, so that the iterations run again with the random numbers they would have had — the chain is the one a
      // run without the time-out walks (our own kernel's stream is mirrored on the host; BPP's lives on the device, where a
      // launch that gives up leaves it alone)
      {
        long left = verr[1];
        while (left > 0 && !s->v2_log.empty())
        {
          const bpa_sampler::V2Launch l = s->v2_log.back(); s->v2_log.pop_back();
          left -= (long)l.chunk;
          s->launches--; s->sweeps -= l.chunk; s->v2_iters -= l.chunk;
          if (!s->kernel_bpp) s->grng = l.grng_before;
          s->mix_pending = l.mix_pending; s->logpr_stale = l.logpr_stale;
        }
        if (left != 0) return fail("bpa_sampler: the persistent kernel reports iterations it did not run that no launch of the log accounts for");
      }
      s->v2_log.clear();
      if (++s->v2_retries > 3)
      {
        if (s->kernel_bpp) return fail("bpa_sampler: the persistent iteration kernel keeps timing out (a shared device: its workgroups must all be resident at once) and BPP's proposal kernel has no other path");
        s->v2_ok = false;
      }
      if (!bpa_sampler_iterate(s, (unsigned)verr[1])) return 0;
      return sampler_download(s);
    }
    s->v2_retries = 0; s->v2_log.clear();
    if (verr[2])
    {
      const int z = 0;
      HIPCHK(hipMemcpy(s->v2_err.p + 2, &z, sizeof z, hipMemcpyHostToDevice));
      return fail("bpa_sampler: an all-loci step was ACCEPTED with a locus's term of 256 log units or more in its sum (the coarse companion accumulator: 2^-10 resolution); the chain is not to be trusted from there on");
    }
  }
  s->host_current = true;
  return 1;
}

extern "C" int bpa_sampler_get_tree(bpa_sampler_t * s, unsigned i, int * left, int * right, int * parent,
                                    double * times, int * clv, int * pmat, int * root, double * lnl)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) { unsigned j; bpa_sampler * p = comp_part(s, i, &j); if (!p) return fail("bpa_sampler_get_tree: locus index out of range"); return comp_upload(s) && bpa_sampler_get_tree(p, j, left, right, parent, times, clv, pmat, root, lnl); }
  if (i >= s->nloci) return fail("bpa_sampler_get_tree: locus index out of range");
  if (!sampler_download(s)) return 0;                   // (a no-op while the host copy is current)
  auto out = [&](const auto & t)
  {
    const int n = 2*t.tips - 1;
    for (int k = 0; k < n; ++k)
    {
      if (left) left[k] = t.left[k]; if (right) right[k] = t.right[k]; if (parent) parent[k] = t.parent[k];
      if (times) times[k] = t.time[k]; if (clv) clv[k] = t.clv[k]; if (pmat) pmat[k] = t.pmat[k];
    }
    if (root) *root = t.root;
    if (lnl) *lnl = t.lnl;
  };
  if (s->big) out(s->b_trees[i]); else if (s->generic) out(s->g_trees[i]); else out(s->h_trees[i]);
  return 1;
}

extern "C" int bpa_sampler_get_tree_msc(bpa_sampler_t * s, unsigned i, int * pop, double * logpr)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) { unsigned j; bpa_sampler * p = comp_part(s, i, &j); if (!p) return fail("bpa_sampler_get_tree_msc: locus index out of range"); return comp_upload(s) && bpa_sampler_get_tree_msc(p, j, pop, logpr); }
  if (i >= s->nloci) return fail("bpa_sampler_get_tree_msc: locus index out of range");
  if (!sampler_download(s)) return 0;
  auto out = [&](const auto & t)
  {
    if (pop) for (int k = 0; k < 2*t.tips - 1; ++k) pop[k] = t.pop[k];
    if (logpr) *logpr = t.logpr;
  };
  if (s->big) out(s->b_trees[i]); else if (s->generic) out(s->g_trees[i]); else out(s->h_trees[i]);
  return 1;
}

extern "C" int bpa_sampler_enable_timing(bpa_sampler_t * s, unsigned stride)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) return comp_each(s, [&](bpa_sampler * p) { return bpa_sampler_enable_timing(p, stride); });
  if (!set_device(s->eng) || !sampler_timing_drain(s)) return 0;
  s->timing_stride = stride; s->timing_phase = 0;
  s->timed_ms[0] = s->timed_ms[1] = 0; s->timed_n[0] = s->timed_n[1] = 0;
  return 1;
}

extern "C" int bpa_sampler_timing(bpa_sampler_t * s, double * sweep_ms, unsigned long * sweep_launches,
                                  double * allloci_ms, unsigned long * allloci_launches)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) return comp_timing(s, sweep_ms, sweep_launches, allloci_ms, allloci_launches);
  if (!set_device(s->eng) || !sampler_timing_drain(s)) return 0;
  if (sweep_ms) *sweep_ms = s->timed_ms[0];
  if (sweep_launches) *sweep_launches = s->timed_n[0];
  if (allloci_ms) *allloci_ms = s->timed_ms[1];
  if (allloci_launches) *allloci_launches = s->timed_n[1];
  return 1;
}

// algorithmic work of the sweep launches so far, by the formulas of SURVEY.md section 8d (JC69, one rate category:
// K1 96 Np + 256 B per node update, K2 36 Np B per evaluated proposal, K4 128 B per fresh P-matrix)
extern "C" int bpa_sampler_work(bpa_sampler_t * s, double * bytes, unsigned long * node_updates,
                                unsigned long * pattern_updates, unsigned long * sweeps)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) return comp_work(s, bytes, node_updates, pattern_updates, sweeps);
  if (!sampler_download(s)) return 0;
  double by = 0; unsigned long nu = 0, pu = 0;
  for (unsigned i = 0; i < s->nloci; ++i)
  {
    // K1 3 Np R S 8 + 2 R S^2 8 bytes per node update, K2 (Np R S 8 + 4 Np) per evaluated proposal, K4 R S^2 8 per fresh P-matrix
    const double np = s->loci[i]->sites, R = s->loci[i]->rate_cats;
    const double nupd = s->big ? s->b_trees[i].work_nupd : s->generic ? s->g_trees[i].work_nupd : (double)s->h_trees[i].sw_nupd + s->h_trees[i].al_nupd,
                 nbr = s->big ? s->b_trees[i].work_nbr : s->generic ? s->g_trees[i].work_nbr : (double)s->h_trees[i].sw_nbr + s->h_trees[i].al_nbr;
    const double nprop = s->big ? s->b_trees[i].work_neval : s->generic ? s->g_trees[i].work_neval : (double)s->h_trees[i].proposals + s->h_trees[i].al_neval;
    // (the generic path counts every evaluated step; where the P-matrix phase is a launch of its own — several rate
    //  categories — the K4 bytes are not the timed kernel's)
    const double S = s->loci[i]->states;
    by += nupd*(3.0*np*R*S*8.0 + 2.0*R*S*S*8.0) + nprop*(R*S*8.0 + 4.0)*np + ((s->generic && !s->g_alljc) ? 0.0 : nbr*R*S*S*8.0);
    nu += (unsigned long)nupd; pu += (unsigned long)(nupd*np);
  }
  if (bytes) *bytes = by;
  if (node_updates) *node_updates = nu;
  if (pattern_updates) *pattern_updates = pu;
  if (sweeps) *sweeps = (s->generic || s->big) ? s->g_evals : s->sweeps;
  return 1;
}

extern "C" int bpa_sampler_kind(bpa_sampler_t * s)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) return comp_upload(s) ? BPA_SAMPLER_COMPOSITE : -1;
  if (!sampler_upload(s)) return -1;
  if (s->big) return BPA_SAMPLER_BIG;
  if (s->generic) return BPA_SAMPLER_GENERIC;
  if (s->v2_ok && !s->env_trace) return s->allreduce ? BPA_SAMPLER_HYBRID : BPA_SAMPLER_PERSISTENT;
  return BPA_SAMPLER_SWEEP;
}

extern "C" int bpa_sampler_streams(bpa_sampler_t * s)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) return comp_upload(s) ? 1 : -1;
  if (!sampler_upload(s)) return -1;
  return s->generic && s->g_split ? (int)s->g_np : 1;
}

extern "C" int bpa_sampler_summary(bpa_sampler_t * s, double * total_lnl, unsigned long * proposals,
                                   unsigned long * accepted, unsigned long * launches)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) return comp_summary(s, total_lnl, proposals, accepted, launches);
  if (!sampler_download(s)) return 0;
  uint32_t c[2];
  HIPCHK(hipMemcpy(c, s->counters.p, 8, hipMemcpyDeviceToHost));
  double tot = 0; unsigned long pr = c[0], ac = c[1];
  if (s->big) for (const auto & t : s->b_trees) { tot += t.lnl; pr += t.proposals; ac += t.accepted; }
  else if (s->generic) for (const auto & t : s->g_trees) { tot += t.lnl; pr += t.proposals; ac += t.accepted; }
  else for (const auto & t : s->h_trees) { tot += t.lnl; pr += t.proposals; ac += t.accepted; }
  if (total_lnl) *total_lnl = tot;
  if (proposals) *proposals = pr;
  if (accepted) *accepted = ac;
  if (launches) *launches = s->launches;
  return 1;
}

#include "gsampler_host.hpp"
#include "bigsampler_host.hpp"
#include "composite.hpp"
