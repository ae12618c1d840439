// gsampler.hpp — the device-resident A00 sampler for loci outside the LDS sweep kernel's scope (sampler.hpp: JC69,
// one rate category, <= 8 tips, <= 64 patterns): any 4-state model of the engine's packing — several rate categories,
// GTR, up to 16 tips, up to 255 lanes (patterns x categories) per locus — i.e. BASELINE configs 3 and 5.
//
// Same moves, same random streams, same arithmetic as the C host driver (csrc/host/a00_driver.c) and as the sweep
// kernel — the proposal code IS the sweep kernel's (propose_gage / propose_gspr / density_prepare / install are
// templates over the tree size and over what a proposal leaves behind) — but the likelihood of a step is not computed
// in LDS by the sampler itself: the step is written, on the device, as the compact records of the engine's batched
// kernels (device_types.hpp: StepRec / StepOp / MatRec2) and evaluated by step_jc69_v2_kernel /
// step_s4_klane_v2_kernel, the kernels every other path uses.  One proposal step of all loci is
//
//     gstep_kernel     one LANE per locus: settle the previous step (accept / roll back), propose, MSC density, records
//     (P-matrix launch + ) the engine's step kernel over the records                  -> lnL per locus
//     all-loci steps only: gsum_decide_kernel (sum of the per-locus terms, ONE decision)
//
// with the trees in HBM between launches (496 B per locus) and nothing on the host but the launch loop.
// Reference: gtree.c:4585 (ages), 6531 (SPR), stree.c:5512 / 4338 (tau + rubber band), prop_mixing.c:52, gtree.c:3957.
#pragma once
#include "gamma_dev.hpp"

namespace gsm {
using smp::Op; using smp::make_op; using smp::Species; using smp::LSpecies; using smp::LCounts; using smp::LTree;
using smp::LUndo; using smp::TimeUndo; using smp::Prof; using smp::MAXPOP; using smp::rndu; using smp::reflect;

constexpr int NT = 16;                    // tips per locus (2 NT - 1 = 31 nodes: node sets are 32-bit masks)
constexpr int NN = 2*NT;
constexpr int GBS = 64;                   // loci per workgroup

struct GTree                              // one per locus, in HBM
{
  int8_t   left[NN], right[NN], parent[NN], clv[NN], pmat[NN], pop[NN];
  double   time[NN];
  double   lnl, logpr;
  a00_rng_t rng;
  int32_t  root, tips;
  uint32_t proposals, accepted;
  uint32_t work_nupd, work_nbr;           // node updates / fresh P-matrices / evaluations of all steps so far (bpa_sampler_work)
  uint32_t work_neval;
  uint32_t pj_gage, pj_gage_acc, pj_gspr, pj_gspr_acc, pad_[3];    // of the proposals / accepted: the gene-node age and the prune-regraft moves (the burn-in's step-length rule wants the move types apart)
};
static_assert(sizeof(GTree) % 16 == 0, "GTree is copied as uint4");

struct GLocus                             // what never changes, per locus
{
  uint32_t slot, pat_off;                 // its slot of the engine's packing, first pattern in the step's term array
  uint32_t R, pad;                        // rate categories
  double * par;                           // the locus's parameter block (device_types.hpp): the substitution-parameter moves write it
  alignas(16) int8_t gl[MAXPOP];          // gene tips below each population
  alignas(16) int8_t nin[MAXPOP];         // gene tips of each species
};

struct GState                             // one per lane, in LDS: what a proposal leaves for its evaluation
{
  double   time[NN];                      // the lane's ages (LTree::time points here)
  Op       ops[NT - 1];
  int32_t  nops;
  uint32_t brm, wclv, chain;
  uint32_t pnodes[MAXPOP];
  int8_t   nin_new[MAXPOP], nc_new[MAXPOP];
};

struct GArgs
{
  GTree * trees, * undo;                  // [T] current (or proposed, while a step is being evaluated) / the state before the pending step
  const GLocus * loc;                     // [T]
  uint32_t T;
  uint32_t i0, iend;                      // the launch's loci [i0, iend) (a half-batch launch; all loci: 0, T)
  uint32_t mode;                          // 0 GAGE k, 1 GSPR k, 2 TAU, 3 MIX, 4 settle (+ THETA statistics), 5 start-up evaluation,
                                          // 6 base frequency k, 7 exchangeability k, 8 alpha (locus.c:2782, 3168; prop_gamma.c:52)
  uint32_t k;
  uint32_t pend;                          // the step to settle first: 0 none, 1 per-locus decisions, 2 an all-loci decision (flag / epoch),
                                          // 3 commit (start-up), 4 per-locus decisions of a substitution-parameter step
  uint32_t pend_mode, pend_k;             // pend 4: which component that step proposed
  double * sm, * sm_old;                  // [T][11] freqs | exchangeabilities | alpha of every locus; [T][2] the pending step's old values
  double ft_freqs, ft_qrates, ft_alpha, alpha_a, alpha_b;
  const double * lnl_new;                 // [T] lnL of the pending step's evaluation (task = locus)
  double * hast, * logpr_new;             // [T] Hastings term / proposed MSC density of the step being proposed (read back when it is settled)
  double * delta;                         // [T] an all-loci step: this locus's density + Jacobian term
  double * lnl_cur;                       // [T] an all-loci step: the locus's current lnL, next to delta (the sum kernel reads compact arrays, not the trees)
  uint8_t * active;                       // [T] the locus has a likelihood evaluation pending
  const uint32_t * flag; uint32_t epoch;  // an all-loci step was REJECTED when *flag == epoch
  // the step's records for the engine's kernels
  uint4 * recs2; uint32_t units;          // [slots*units]
  MatRec2 * mat2; double * mat_length; uint32_t maxmat;     // [slots*maxmat]
  const double * taus; const double * lograt;
  uint32_t tau_q; double tau_u, mix_c, mix_lnc;
  int8_t * pop_nc; double * pop_t2h;      // [MAXPOP][T] mode 4: the statistics the THETA kernel reads
  uint32_t refresh_logpr;
  // 20-state loci (fmt20): the step as the records of the tiled kernels — partials_lnl_pipe20_kernel reads ops20[op_rng20[2 i]
  // .. op_rng20[2 i + 1]) and root20[i], pmatrix_wg2_kernel the entries mat_task20 / mat_pm20 / mat_length [i maxmat + j]
  // (task 0xffffffff: a hole) — one task per locus
  OpDev * ops20; uint32_t * op_rng20, * root20, * mat_task20, * mat_pm20;
  uint32_t fmt20, maxops20;
  double tau_w;                           // bpp: the TAU window itself (the host's Bactrian-Laplace variate; tau_u - 1/2 would round)
  uint32_t bpp, prog;                     // the reference's generator / windows / acceptance rule (bpa_sampler_set_proposal_kernel); the program's THETA / TAU / MIX, decided on the host
  double * t2h3;                          // [T][3] prog, TAU q: the T2h of q and of its two children after the move
  // 4-state loci on the engine's packing: the lane group that wrote the step's fresh branches also fills their P-matrices
  // (what pmatrix_s4_dense_kernel did as a launch of its own between the proposal and the node updates)
  const SlotStatic * slot_tab; uint32_t fuse_pm;
  const struct GDecState * dstep;         // the program's moves decided on the device: the TAU window / MIX factor of this step (else null: tau_w, mix_c, mix_lnc)
  uint32_t fuse_eigen, pad_fe;            // gstep_kernel: a lane that writes frequencies / exchangeabilities into its locus's parameter block refreshes the block's eigensystem itself
  Species sp;
};

// the one-lane kernel's draws under either proposal kernel (smp2::Stream, sweep2.hpp), chosen at run time: GArgs::bpp
__device__ __forceinline__ double g_window(a00_rng_t & r, const uint32_t bpp)
{
  if (bpp) { smp2::Stream<true> st{r}; const double w = st.window(); r = st.r; return w; }
  return rndu(&r) - 0.5;
}
__device__ __forceinline__ bool g_accept(a00_rng_t & r, const uint32_t bpp, const double lnacc)
{
  if (bpp) { smp2::Stream<true> st{r}; const bool a = st.accept(lnacc); r = st.r; return a; }
  const double u = rndu(&r);
  return lnacc >= 0 || u < exp(lnacc);
}

// the MSC density term of population p from the lane's own state (density_term of sampler.hpp, ages in S.time)
__device__ __forceinline__ double gdensity_term(const GState & S, const Species & sp, const double * tau, int p, double & T2h_out)
{
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
      int best = -1; double tb = 0;
      for (uint32_t m = nodes; m; m &= m - 1) { const int x = __ffs(m) - 1; const double tx = S.time[x]; if (best < 0 || tx < tb) { best = x; tb = tx; } }
      tk = tb; nodes &= ~(1u << best);
    }
    T2h += nn*(nn - 1)*(tk - prev);
    prev = tk;
  }
  double c = 0;
  if (ncoal) c += ncoal*tau[2*MAXPOP + p];
  if (T2h) c -= T2h/(tau[MAXPOP + p]*1.0);
  T2h_out = T2h;
  return c;
}

// the whole density, populations in order (tree_logpr of a00_driver.c); optionally leaves the THETA statistics
template <int TT>
__device__ double gdensity(GState & S, const LTree<TT> & T, LCounts & cn, const LSpecies & sp, const Species & spl, const double * tau,
                           int8_t * nc_out, double * t2h_out, uint32_t stride)
{
  smp::density_prepare<TT>(S, T, cn, sp, (1u << sp.npop) - 1u);
  double logpr = 0;
  for (int p = 0; p < sp.npop; ++p)
  {
    double t2h;
    logpr += gdensity_term(S, spl, tau, p, t2h);
    if (nc_out) { nc_out[(size_t)p*stride] = S.nc_new[p]; t2h_out[(size_t)p*stride] = t2h; }
  }
  return logpr;
}

__global__ void glograt_kernel(double * tab)         // log(i/j), i, j < NN
{
  const int i = threadIdx.x / NN, j = threadIdx.x % NN;
  tab[threadIdx.x] = (i && j) ? log((double)i/(double)j) : 0.0;
}

// component `mode` (6 frequencies, 7 exchangeabilities, 8 alpha -> category rates) of a locus's values into its parameter block
__device__ __forceinline__ void write_par(double * par, uint32_t R, uint32_t mode, const double * m)
{
  if (mode == 6)      for (int q = 0; q < 4; ++q) par[par_matrix(R, 4, 0) + pm_freqs(4) + q] = m[q];
  else if (mode == 7) for (int q = 0; q < 6; ++q) par[par_matrix(R, 4, 0) + pm_subst(4) + q] = m[4 + q];
  else
  {
    double rates[8];
    gdev::gamma_cats(m[10], R, rates);                   // prop_gamma.c:93-97
    for (uint32_t q = 0; q < R; ++q) par[par_rates(R) + q] = rates[q];
  }
}

// MODE: GArgs::mode as a compile-time constant, one instance per move (the whole kernel is 17 k instructions otherwise);
// TT: a bound on the tips of the sampler's loci (8 or 16) — the register trees and the unrolled node passes are sized by it
template <uint32_t MODE, int TT>
__global__ void __launch_bounds__(GBS) gstep_kernel(const GArgs A)
{
  __shared__ GState s_st[GBS];
  __shared__ double s_tau[3*MAXPOP];
  constexpr int TN = 2*TT;
  __shared__ double s_lograt[TN*TN];
  __shared__ Species s_sp;
  const uint32_t lane = threadIdx.x, i = A.i0 + blockIdx.x*GBS + lane;
  const bool valid = i < A.iend;
  {
    const uint32_t * src = reinterpret_cast<const uint32_t *>(&A.sp);
    uint32_t * dst = reinterpret_cast<uint32_t *>(&s_sp);
    for (uint32_t q = lane; q < sizeof(Species)/4; q += GBS) dst[q] = src[q];
  }
  if (lane < (uint32_t)(3*MAXPOP)) s_tau[lane] = A.taus[lane];
  if (MODE == 1) for (uint32_t q = lane; q < (uint32_t)(TN*TN); q += GBS) s_lograt[q] = A.lograt[(q/TN)*NN + q % TN];
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
  GState & S = s_st[lane];
  LTree<TT> T;
  LCounts cn;
  for (int k = 0; k < 4; ++k) cn.nin.w[k] = cn.nc.w[k] = cn.nin_new.w[k] = cn.nc_new.w[k] = cn.gl.w[k] = 0u;
  T.time = S.time; T.rng = 0; T.root = 0; T.tips = 2;
  double lnl_cur = 0, logpr_cur = 0;
  uint32_t nprop = 0, nacc = 0, w_nupd = 0, w_nbr = 0, w_nev = 0;
  GLocus L{};
  if (valid)
  {
    GTree & g = A.trees[i];
    L = A.loc[i];
    T.left.load(g.left); T.right.load(g.right); T.parent.load(g.parent); T.clv.load(g.clv); T.pmat.load(g.pmat); T.pop.load(g.pop);
    for (int k = 0; k < TN; ++k) S.time[k] = g.time[k];
    T.rng = g.rng; T.root = g.root; T.tips = g.tips;
    lnl_cur = g.lnl; logpr_cur = g.logpr; nprop = g.proposals; nacc = g.accepted; w_nupd = g.work_nupd; w_nbr = g.work_nbr; w_nev = g.work_neval;
    const uint4 g4 = *reinterpret_cast<const uint4 *>(L.gl), n4 = *reinterpret_cast<const uint4 *>(L.nin);
    cn.gl.w[0] = g4.x; cn.gl.w[1] = g4.y; cn.gl.w[2] = g4.z; cn.gl.w[3] = g4.w;
    cn.nin.w[0] = n4.x; cn.nin.w[1] = n4.y; cn.nin.w[2] = n4.z; cn.nin.w[3] = n4.w;
    S.nops = 0; S.brm = 0; S.wclv = 0; S.chain = 0;
  }
  __syncthreads();

  bool restore_par = false, propose_par = false;       // the locus's parameter block is written in ONE place, below (step 4b)
  // ---- 1. settle the step whose evaluation just finished
  if (valid && A.pend)
  {
    bool back = false;
    if (A.pend == 1)
    {
      if (A.active[i])
      {
        const double lnl = A.lnl_new[i], lp_new = A.logpr_new[i];
        const double lnacc = (lp_new - logpr_cur) + (lnl - lnl_cur) + A.hast[i];
        ++nprop;
        const bool acc_ = g_accept(T.rng, A.bpp, lnacc);
        if (A.pend_mode == 0) { A.trees[i].pj_gage += 1; A.trees[i].pj_gage_acc += acc_ ? 1u : 0u; }
        else if (A.pend_mode == 1) { A.trees[i].pj_gspr += 1; A.trees[i].pj_gspr_acc += acc_ ? 1u : 0u; }
        if (acc_) { lnl_cur = lnl; logpr_cur = lp_new; ++nacc; }
        else back = true;
      }
    }
    else if (A.pend == 2)
    {
      if (*A.flag == A.epoch) back = true;
      else { logpr_cur = A.logpr_new[i]; if (A.active[i]) lnl_cur = A.lnl_new[i]; }
    }
    else if (A.pend == 4)
    {
      if (A.active[i])
      {
        const double lnl = A.lnl_new[i];
        const double lnacc = (lnl - lnl_cur) + A.hast[i];
        ++nprop;
        if (g_accept(T.rng, A.bpp, lnacc)) { lnl_cur = lnl; ++nacc; }
        else
        {
          // the old values come back, in the sampler's copy and in the locus's parameter block
          back = true;
          double * m = A.sm + (size_t)i*11;
          if (A.pend_mode == 8) m[10] = A.sm_old[2*i];
          else { double * v = A.pend_mode == 6 ? m : m + 4; const int ref = A.pend_mode == 6 ? 3 : 1; v[A.pend_k] = A.sm_old[2*i]; v[ref] = A.sm_old[2*i + 1]; }
          restore_par = true;
        }
      }
    }
    else { lnl_cur = A.lnl_new[i]; logpr_cur = A.logpr_new[i]; }
    if (back)
    {
      const GTree & u = A.undo[i];
      T.left.load(u.left); T.right.load(u.right); T.parent.load(u.parent); T.clv.load(u.clv); T.pmat.load(u.pmat); T.pop.load(u.pop);
      for (int k = 0; k < TN; ++k) S.time[k] = u.time[k];
      T.root = u.root;
    }
  }
  // ---- 2. THETA moved the thetas since this density was stored
  if (valid && A.refresh_logpr) logpr_cur = gdensity(S, T, cn, sp, spl, s_tau, nullptr, nullptr, 0);
  __syncthreads();

  // ---- 3. the proposed species tree of an all-loci step is this workgroup's copy of the taus
  double lminf = 0, lmaxf = 0, tq_old = 0, tq_lo = 0, tq_hi = 0, minf = 1, maxf = 1;
  if (MODE == 2)
  {
    const int q = (int)A.tau_q, pq = spl.parent[q];
    tq_old = s_tau[q]; tq_lo = fmax(s_tau[spl.left[q]], s_tau[spl.right[q]]); tq_hi = pq >= 0 ? s_tau[pq] : 999.0;
    const double tnew = reflect(tq_old + spl.ft_tau*(A.tau_u - 0.5), tq_lo, tq_hi);
    minf = (tnew - tq_lo)/(tq_old - tq_lo); maxf = (tnew - tq_hi)/(tq_old - tq_hi);
    lminf = log(minf); lmaxf = log(maxf);
    __syncthreads();
    if (lane == 0) s_tau[q] = tnew;
    __syncthreads();
  }
  else if (MODE == 3)
  {
    if (lane < (uint32_t)npop) s_tau[lane] *= A.mix_c;
    __syncthreads();
  }

  // ---- 4. propose
  bool evaluate = false;
  if (valid && MODE != 4)
  {
    // the state a rejection comes back to
    {
      GTree & u = A.undo[i];
      T.left.store(u.left); T.right.store(u.right); T.parent.store(u.parent); T.clv.store(u.clv); T.pmat.store(u.pmat); T.pop.store(u.pop);
      for (int k = 0; k < TN; ++k) u.time[k] = S.time[k];
      u.root = T.root;
    }
    TimeUndo tu{0, 0, 0, -1, -1, -1};
    Prof pf; pf.on = false; pf.t = 0;
    double hast = 0;
    bool ok = true;
    if (MODE == 0)      ok = smp::propose_gage<TT>(S, T, cn, tu, hast, (int)A.k, sp, spl, s_tau, pf);
    else if (MODE == 1) ok = smp::propose_gspr<TT>(S, T, cn, tu, hast, (int)A.k, sp, spl, s_tau, s_lograt, pf);
    else if (MODE == 2)
    {
      // TAU q (tau_step of a00_driver.c): the gene nodes of q and its children between the bounds move
      const int nn_ = 2*T.tips - 1, q = (int)A.tau_q, cl = sp.left[q], cr = sp.right[q];
      uint32_t brm = 0, ndm = 0; int above = 0, below = 0;
      for (int k = T.tips; k < nn_; ++k)
      {
        const int pk = T.pop[k]; const double tk = S.time[k];
        if ((pk != q && pk != cl && pk != cr) || tk < tq_lo || tk > tq_hi) continue;
        if (tk >= tq_old) { S.time[k] = tq_hi + maxf*(tk - tq_hi); ++above; } else { S.time[k] = tq_lo + minf*(tk - tq_lo); ++below; }
        brm |= (1u << (int)T.left[k]) | (1u << (int)T.right[k]);
        if (T.parent[k] >= 0) brm |= 1u << k;
        ndm |= smp::path_mask(T, k);
      }
      const double lp_new = gdensity(S, T, cn, sp, spl, s_tau, nullptr, nullptr, 0);
      A.logpr_new[i] = lp_new;
      A.delta[i] = ((lp_new - logpr_cur) + below*lminf) + above*lmaxf;          // p_delta of the host driver
      A.lnl_cur[i] = lnl_cur;
      if (ndm) smp::install<TT>(S, T, brm, ndm);
      else { S.brm = 0; S.nops = 0; ok = false; }                              // no gene node moves here: only the density changes
    }
    else if (MODE >= 6)
    {
      // a substitution-parameter move (param_step of a00_driver.c): new value, then every branch, every inner node
      double * m = A.sm + (size_t)i*11;
      if (MODE == 8 && L.R < 2) ok = false;
      else
      {
        if (MODE == 8)
        {
          const double a_old = m[10], la_old = log(a_old);
          const double la_new = reflect(la_old + A.ft_alpha*g_window(T.rng, A.bpp), -99.0, 99.0);
          const double a_new = exp(la_new);
          A.sm_old[2*i] = a_old;
          m[10] = a_new;
          hast = (la_new - la_old) + ((A.alpha_a - 1)*log(a_new/a_old) - A.alpha_b*(a_new - a_old));
        }
        else
        {
          double * v = MODE == 6 ? m : m + 4;
          const int j = (int)A.k, ref = MODE == 6 ? 3 : 1;
          const double sum = v[j] + v[ref], lo = log(1e-5), hi = log(sum);
          const double l_old = log(v[j]);
          const double l_new = reflect(l_old + (MODE == 6 ? A.ft_freqs : A.ft_qrates)*g_window(T.rng, A.bpp), lo, hi);
          A.sm_old[2*i] = v[j]; A.sm_old[2*i + 1] = v[ref];
          v[j] = exp(l_new); v[ref] = sum - v[j];
          hast = l_new - l_old;
        }
        propose_par = true;
        const int nn_ = 2*T.tips - 1;
        uint32_t brm = 0, ndm = 0;
        for (int k = 0; k < nn_; ++k)
        {
          if (T.left[k] >= 0) ndm |= 1u << k;
          if (T.parent[k] >= 0) brm |= 1u << k;
        }
        A.hast[i] = hast;
        A.logpr_new[i] = logpr_cur;
        smp::install<TT>(S, T, brm, ndm);
      }
    }
    else
    {
      // mixing (mix_step of a00_driver.c) or start-up: every branch, every inner node
      const int nn_ = 2*T.tips - 1;
      uint32_t brm = 0, ndm = 0; int ninner = 0;
      for (int k = 0; k < nn_; ++k)
      {
        if (T.left[k] >= 0) { if (MODE == 3) S.time[k] *= A.mix_c; ndm |= 1u << k; ++ninner; }
        if (T.parent[k] >= 0) brm |= 1u << k;
      }
      if (MODE == 5)
      {
        for (uint32_t m = brm; m; m &= m - 1) smp::swap_pmat(T, __ffs(m) - 1);       // start-up evaluates in place:
        for (uint32_t m = ndm; m; m &= m - 1) smp::swap_clv(T, __ffs(m) - 1);        // toggle twice = no toggle
      }
      const double lp_new = gdensity(S, T, cn, sp, spl, s_tau, nullptr, nullptr, 0);
      A.logpr_new[i] = lp_new;
      A.delta[i] = (lp_new - logpr_cur) + (double)ninner*A.mix_lnc;
      A.lnl_cur[i] = lnl_cur;
      smp::install<TT>(S, T, brm, ndm);
    }
    if (ok && MODE <= 1)
    {
      A.logpr_new[i] = gdensity(S, T, cn, sp, spl, s_tau, nullptr, nullptr, 0);
      A.hast[i] = hast;
    }
    if (ok && MODE != 5) { w_nupd += (uint32_t)S.nops; w_nbr += (uint32_t)__popc(S.brm); ++w_nev; }
    evaluate = ok;
    A.active[i] = ok ? 1 : 0;
  }
  else if (valid) A.active[i] = 0;

  // ---- 4b. the parameter block follows the sampler's copy: first what a rejected step puts back, then the new proposal
  if (valid && (restore_par || propose_par))
  {
    const double * m = A.sm + (size_t)i*11;
    for (int pass = 0; pass < 2; ++pass)
    {
      const bool on = pass == 0 ? restore_par : propose_par;
      const uint32_t md = pass == 0 ? A.pend_mode : MODE;
      if (!on) continue;
      // (restored and proposed component may be the same one: this order leaves the proposal in the block.  NB the
      //  values of `m` are final here — a restored component and a newly proposed one never overlap in time)
      write_par(L.par, L.R, md, m);
    }
    // (round 6) K6 right here: the lane that moved its locus's frequencies / exchangeabilities (proposed new ones, or put a rejected
    // proposal's back) refreshes the eigensystem of the block's rate matrix 0 — the one these moves write — in the same launch.
    // It was a launch of its own (eigen_kernel over ALL loci: 20 us + a launch boundary) between every parameter proposal and its
    // P-matrices: 16 launches of an iteration of config 3.  Same routine, same inputs, same bits.
    const bool eig = (restore_par && (A.pend_mode == 6 || A.pend_mode == 7)) || (propose_par && (MODE == 6 || MODE == 7));
    if (A.fuse_eigen && eig)
    {
      double * pm = L.par + par_matrix(L.R, 4, 0);
      update_eigen_regs<4>(pm + pm_freqs(4), pm + pm_subst(4), pm + pm_evals(4), pm + pm_evecs(4), pm + pm_ievecs(4));
    }
  }

  // ---- 5. the step's records for the engine's kernels
  if (valid && MODE != 4 && A.fmt20)
  {
    const uint32_t e0 = i*A.maxmat, o0 = i*A.maxops20;
    uint32_t nm = 0;
    if (evaluate)
    {
      for (uint32_t m = S.brm; m; m &= m - 1, ++nm)
      {
        const int x = __ffs(m) - 1;
        A.mat_task20[e0 + nm] = i; A.mat_pm20[e0 + nm] = (uint32_t)(int)T.pmat[x];
        A.mat_length[e0 + nm] = (S.time[(int)T.parent[x]] - S.time[x])*1.0;      // rate_mui = 1 (locus.c:2350)
      }
      for (int o = 0; o < S.nops; ++o)
      {
        const Op w = S.ops[o];
        OpDev q;
        q.parent_clv = (uint32_t)(w & 255u); q.left_clv = (uint32_t)((w >> 8) & 255u); q.left_pmatrix = (uint32_t)((w >> 16) & 255u);
        q.right_clv = (uint32_t)((w >> 24) & 255u); q.right_pmatrix = (uint32_t)((w >> 32) & 255u);
        q.parent_scaler = q.left_scaler = q.right_scaler = BPA_SCALE_BUFFER_NONE;
        A.ops20[o0 + o] = q;
      }
    }
    for (; nm < A.maxmat; ++nm) A.mat_task20[e0 + nm] = 0xffffffffu;
    A.op_rng20[2*i] = o0; A.op_rng20[2*i + 1] = o0 + (evaluate ? (uint32_t)S.nops : 0u);
    A.root20[i] = (uint32_t)(int)T.clv[T.root];
  }
  else if (valid && MODE != 4)
  {
    uint4 * rec = A.recs2 + (size_t)L.slot*A.units;
    MatRec2 * m2 = A.mat2 + (size_t)L.slot*A.maxmat;
    double * ml = A.mat_length + (size_t)L.slot*A.maxmat;
    const uint32_t e0 = L.slot*A.maxmat;
    StepRec h{};
    h.task = evaluate ? i : 0xffffffffu; h.pat_off = L.pat_off;
    h.root_clv = (uint8_t)(int)T.clv[T.root]; h.root_scaler = (int8_t)BPA_SCALE_BUFFER_NONE; h.nops = (uint8_t)(evaluate ? S.nops : 0);
    *reinterpret_cast<StepRec *>(rec) = h;
    uint32_t nm = 0;
    if (evaluate)
    {
      for (uint32_t m = S.brm; m; m &= m - 1, ++nm)
      {
        const int x = __ffs(m) - 1;
        m2[nm] = MatRec2{L.slot, (uint32_t)(int)T.pmat[x]};
        ml[nm] = (S.time[(int)T.parent[x]] - S.time[x])*1.0;                   // rate_mui = 1 (locus.c:2350)
      }
      for (int o = 0; o < S.nops; ++o)
      {
        const Op w = S.ops[o];
        StepOp q{};
        q.parent_clv = (uint8_t)(w & 255u); q.left_clv = (uint8_t)((w >> 8) & 255u); q.left_pmatrix = (uint8_t)((w >> 16) & 255u);
        q.right_clv = (uint8_t)((w >> 24) & 255u); q.right_pmatrix = (uint8_t)((w >> 32) & 255u);
        q.parent_scaler = q.left_scaler = q.right_scaler = (int8_t)BPA_SCALE_BUFFER_NONE;
        q.left_e = q.right_e = -1;
        uint32_t j = 0;
        for (uint32_t m = S.brm; m; m &= m - 1, ++j)
        {
          const uint32_t pm = (uint32_t)(int)T.pmat[__ffs(m) - 1];
          if (pm == q.left_pmatrix)  q.left_e = (int32_t)(e0 + j);
          if (pm == q.right_pmatrix) q.right_e = (int32_t)(e0 + j);
        }
        *reinterpret_cast<StepOp *>(rec + 1 + o) = q;
      }
    }
    for (; nm < A.maxmat; ++nm) m2[nm] = MatRec2{0xffffffffu, 0u};              // entries of the last step this one does not use
  }

  // ---- 6. mode 4: the sufficient statistics of the settled state, for the THETA kernel
  if (valid && MODE == 4) (void)gdensity(S, T, cn, sp, spl, s_tau, A.pop_nc + i, A.pop_t2h + i, A.T);

  // ---- 7. store
  if (valid)
  {
    GTree & g = A.trees[i];
    T.left.store(g.left); T.right.store(g.right); T.parent.store(g.parent); T.clv.store(g.clv); T.pmat.store(g.pmat); T.pop.store(g.pop);
    for (int k = 0; k < TN; ++k) g.time[k] = S.time[k];
    g.rng = T.rng; g.root = T.root; g.lnl = lnl_cur; g.logpr = logpr_cur; g.proposals = nprop; g.accepted = nacc;
    g.work_nupd = w_nupd; g.work_nbr = w_nbr; g.work_neval = w_nev;
  }
}

// an all-loci step: the sum of the loci's terms (fixed order) and the ONE decision (tau_step / mix_step of a00_driver.c)
__global__ void __launch_bounds__(1024) gsum_decide_kernel(const double * __restrict__ lnl_cur, const double * __restrict__ lnl_new,
                                                           const double * __restrict__ delta, const uint8_t * __restrict__ active,
                                                           uint32_t T, double * sum_out, int decide_on, double u, uint32_t epoch,
                                                           uint32_t * flag, uint32_t * counters, double * taus, Species sp, int tau_q,
                                                           double win_u, double mix_c, double mix_lnc)
{
  __shared__ double sh[1024];
  double acc = 0;
  for (uint32_t i = threadIdx.x; i < T; i += 1024) acc += (active[i] ? lnl_new[i] - lnl_cur[i] : 0.0) + delta[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t w = 512; w > 0; w >>= 1)
  {
    if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x) return;
  if (sum_out) sum_out[0] = sh[0];
  if (decide_on) smp::decide(sh[0], u, epoch, flag, counters, taus, sp, tau_q, -1, win_u, mix_c, mix_lnc);
}

// ---- the program's THETA / TAU / MIX for the generic samplers: the loci's sums on the device, the decision on the host ----
// (bpa_sampler_set_program_moves on a generic sampler; theta_step_gibbs / tau_step / mix_step of a00_driver.c are the model.)
// The decisions are long scalar chains — a 35-step bisection per theta fit, gamma variates, a dozen logarithms — over a few
// numbers; the persistent kernel gives them a control wave, here they run on the host between two launches: 8 to 72 bytes
// come back per all-loci step, one synchronisation each, 9 per iteration of config 3 against its ~230 launches.
// out[0] = sum over the loci of (lnL' - lnL, where a likelihood was evaluated) + delta — gsum_decide_kernel's sum, same order;
// TAU: out[1..3] = the T2h of q and its two children after the move, summed as 2^-40 fixed point (a00_driver.c: llrint(x 2^40)),
// out[4] = 1 when a term was unusable (NaN or >= 256)
constexpr int GPROG_FLAG0 = 3*MAXPOP;          // the sum kernels' output block: [0, 3 MAXPOP) values, then one arrival word per workgroup
__global__ void __launch_bounds__(1024) gprog_sums_kernel(const double * __restrict__ lnl_cur, const double * __restrict__ lnl_new,
                                                          const double * __restrict__ delta, const uint8_t * __restrict__ active,
                                                          const double * __restrict__ t2h3, uint32_t T, int with_t2h, double * out,
                                                          unsigned long long seq)
{
  __shared__ double sh[1024];
  __shared__ long long shl[3][1024];
  __shared__ int shb[1024];
  double acc = 0; long long c[3] = {0, 0, 0}; int bad = 0;
  for (uint32_t i = threadIdx.x; i < T; i += 1024)
  {
    acc += (active[i] ? lnl_new[i] - lnl_cur[i] : 0.0) + delta[i];
    if (with_t2h)
      for (int j = 0; j < 3; ++j)
      {
        const double x = t2h3[(size_t)3*i + j];
        if (!(fabs(x) < 256.0)) bad = 1; else c[j] += llrint(x*1099511627776.0);
      }
  }
  sh[threadIdx.x] = acc; shb[threadIdx.x] = bad;
  for (int j = 0; j < 3; ++j) shl[j][threadIdx.x] = c[j];
  __syncthreads();
  for (uint32_t w = 512; w > 0; w >>= 1)
  {
    if (threadIdx.x < w)
    {
      sh[threadIdx.x] += sh[threadIdx.x + w]; shb[threadIdx.x] |= shb[threadIdx.x + w];
      for (int j = 0; j < 3; ++j) shl[j][threadIdx.x] += shl[j][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x) return;
  out[0] = sh[0];
  for (int j = 0; j < 3; ++j) reinterpret_cast<long long *>(out)[1 + j] = shl[j][0];
  reinterpret_cast<long long *>(out)[4] = shb[0];
  // out in pinned host memory (gs_prog_out): the host polls this word instead of waiting for the launch to retire
  if (seq) { __threadfence_system(); __hip_atomic_store(reinterpret_cast<unsigned long long *>(out) + GPROG_FLAG0, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}

// THETA: per population the coalescences and the T2h over all loci (theta_sums of a00_driver.c: integers, no order)
__global__ void __launch_bounds__(1024) gprog_theta_sums_kernel(const int8_t * __restrict__ pop_nc, const double * __restrict__ pop_t2h,
                                                                uint32_t T, uint32_t onmask, long long * out, unsigned long long seq)
{
  __shared__ long long shk[1024], sht[1024];
  __shared__ int shb[1024];
  const int p = (int)blockIdx.x;
  long long k = 0, t = 0; int bad = 0;
  if ((onmask >> p) & 1u)
    for (uint32_t i = threadIdx.x; i < T; i += 1024)
    {
      const double x = pop_t2h[(size_t)p*T + i];
      if (!(fabs(x) < 256.0)) bad = 1; else { k += pop_nc[(size_t)p*T + i]; t += llrint(x*1099511627776.0); }
    }
  shk[threadIdx.x] = k; sht[threadIdx.x] = t; shb[threadIdx.x] = bad;
  __syncthreads();
  for (uint32_t w = 512; w > 0; w >>= 1)
  {
    if (threadIdx.x < w) { shk[threadIdx.x] += shk[threadIdx.x + w]; sht[threadIdx.x] += sht[threadIdx.x + w]; shb[threadIdx.x] |= shb[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    out[3*p] = shk[0]; out[3*p + 1] = sht[0]; out[3*p + 2] = shb[0];
    if (seq) { __threadfence_system(); __hip_atomic_store(reinterpret_cast<unsigned long long *>(out) + GPROG_FLAG0 + p, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
}

// the host's decision brought to the device: the rejection flag, the species tree as it now is, the counters
struct GApply { uint32_t accept, set_tau_q, scale_taus, nprop, nacc, ngprop, ngacc, theta_mask; double tau_new, mix_c; double theta[MAXPOP]; };
__global__ void gprog_apply_kernel(const GApply a, uint32_t epoch, uint32_t * flag, uint32_t * counters, double * taus, int npop)
{
  if (threadIdx.x || blockIdx.x) return;
  counters[0] += a.nprop; counters[1] += a.nacc; counters[2] += a.ngprop; counters[3] += a.ngacc;
  if (!a.accept) { *flag = epoch; return; }
  if (a.set_tau_q != 0xffffffffu) taus[a.set_tau_q] = a.tau_new;
  if (a.scale_taus) for (int p = 0; p < npop; ++p) taus[p] *= a.mix_c;
  for (int p = 0; p < npop; ++p)
    if ((a.theta_mask >> p) & 1u) { taus[MAXPOP + p] = a.theta[p]; taus[2*MAXPOP + p] = log(2.0/(1.0*a.theta[p])); }
}


// ======================= the program's THETA / TAU / MIX decided ON THE DEVICE (round 6) =======================================
// The host-decided form above costs an iteration nine synchronisations: the stream drains, the host fits inverse gammas and
// draws, a one-lane launch installs the result — 25-40 us each during which the GPU has nothing queued, and the reason an
// iteration of the generic sampler could not be one uninterrupted sequence of launches (config 3, 193 launches and
// 9 host synchronisations).  Here ONE WAVE takes the decision where the sums are: gdec_kernel runs the persistent kernel's
// control-wave functions (sweep2.hpp: prog_theta_decide / prog_tau_decide / prog_mix_redraw — lane p = population p, the fits of
// all thetas side by side, the gamma variates drawn in parallel, bit-equal to get_gamma_conditional_approx on 1e5 cases,
// tests/test_bpp_kernel.py), installs it (rejection flag, taus, thetas, counters: gprog_apply_kernel's job) and makes the NEXT
// step's species-tree proposal (the window variate of the coming TAU; log c and the theta re-draws of the coming MIX), which the
// step kernels read from GDecState instead of their arguments.  The global stream, the fits (k, T, a, b, c per population) and
// the move-type counters live in GDecState between launches.  Stream order = a00_driver.c's = the host form's:
// [first TAU's window] [THETA: choices + windows] [Gibbs variates] [acceptance numbers] [TAU: variates, acceptance] [next window] ...
// The host form stays as the trajectory reference (BPA_GS_HOSTDEC=1) — same decisions, tests/test_gpu_gsampler.py.
//
// The sums arrive as DOUBLES in the all-reduce callback's format (gs_prog_allreduce): a likelihood / Jacobian sum as it is, a
// non-negative 64-bit integer sum as two doubles holding its upper and lower 32 bits (exact for any number of ranks) — so several
// ranks put their collective between the sum kernel and this one, on the stream, and no host is in the loop either.
struct GDecState
{
  uint32_t z, run_ok, rd_mask, pad0;                 // the global stream (legacy_rndu); THETA's sums were usable; MIX's re-drawn thetas
  double tau_w, mix_c, mix_lnc, lnacc_theta;         // the coming step's proposal (step kernels: GArgs::dstep) and MIX's theta part
  smp2::PopFit pf[16];
  smp2::Redraw rd[16];
  unsigned long long pj[10];                         // [4..9] tau / mix / theta-window proposals and acceptances since the last adapt_finetune
};

__device__ __forceinline__ void gdec_split64(long long x, double * hi, double * lo) { *hi = (double)(x >> 32); *lo = (double)(x & 0xffffffffll); }
__device__ __forceinline__ long long gdec_join64(double hi, double lo) { return (long long)hi*4294967296ll + (long long)lo; }

// gprog_sums_kernel / gprog_theta_sums_kernel with their results as doubles in a device buffer
__global__ void __launch_bounds__(1024) gdec_sums_kernel(const double * __restrict__ lnl_cur, const double * __restrict__ lnl_new,
                                                         const double * __restrict__ delta, const uint8_t * __restrict__ active,
                                                         const double * __restrict__ t2h3, uint32_t T, int with_t2h, double * out)
{
  __shared__ double sh[1024];
  __shared__ long long shl[3][1024];
  __shared__ int shb[1024];
  double acc = 0; long long c[3] = {0, 0, 0}; int bad = 0;
  for (uint32_t i = threadIdx.x; i < T; i += 1024)
  {
    acc += (active[i] ? lnl_new[i] - lnl_cur[i] : 0.0) + delta[i];
    if (with_t2h)
      for (int j = 0; j < 3; ++j)
      {
        const double x = t2h3[(size_t)3*i + j];
        if (!(fabs(x) < 256.0)) bad = 1; else c[j] += llrint(x*1099511627776.0);
      }
  }
  sh[threadIdx.x] = acc; shb[threadIdx.x] = bad;
  for (int j = 0; j < 3; ++j) shl[j][threadIdx.x] = c[j];
  __syncthreads();
  for (uint32_t w = 512; w > 0; w >>= 1)
  {
    if (threadIdx.x < w)
    {
      sh[threadIdx.x] += sh[threadIdx.x + w]; shb[threadIdx.x] |= shb[threadIdx.x + w];
      for (int j = 0; j < 3; ++j) shl[j][threadIdx.x] += shl[j][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x) return;
  bool neg = false;
  for (int j = 0; j < 3; ++j) neg = neg || shl[j][0] < 0;
  out[0] = sh[0];
  for (int j = 0; j < 3; ++j) gdec_split64(neg ? 0 : shl[j][0], &out[1 + 2*j], &out[2 + 2*j]);
  out[7] = (neg || shb[0]) ? 1.0 : 0.0;
}

__global__ void __launch_bounds__(1024) gdec_theta_sums_kernel(const int8_t * __restrict__ pop_nc, const double * __restrict__ pop_t2h,
                                                               uint32_t T, uint32_t onmask, double * out)
{
  __shared__ long long shk[1024], sht[1024];
  __shared__ int shb[1024];
  const int p = (int)blockIdx.x;
  long long k = 0, t = 0; int bad = 0;
  if ((onmask >> p) & 1u)
    for (uint32_t i = threadIdx.x; i < T; i += 1024)
    {
      const double x = pop_t2h[(size_t)p*T + i];
      if (!(fabs(x) < 256.0)) bad = 1; else { k += pop_nc[(size_t)p*T + i]; t += llrint(x*1099511627776.0); }
    }
  shk[threadIdx.x] = k; sht[threadIdx.x] = t; shb[threadIdx.x] = bad;
  __syncthreads();
  for (uint32_t w = 512; w > 0; w >>= 1)
  {
    if (threadIdx.x < w) { shk[threadIdx.x] += shk[threadIdx.x + w]; sht[threadIdx.x] += sht[threadIdx.x + w]; shb[threadIdx.x] |= shb[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    const bool neg = shk[0] < 0 || sht[0] < 0;         // (never: counts and waiting times)
    out[4*p] = (double)(neg ? 0 : shk[0]); gdec_split64(neg ? 0 : sht[0], &out[4*p + 1], &out[4*p + 2]); out[4*p + 3] = (neg || shb[0]) ? 1.0 : 0.0;
  }
}

// PHASE 0: THETA (v: 4 doubles per population) + the first TAU's window; 1: TAU q (v: 8 doubles) + the next step's proposal
// (next < npop: TAU `next`, next == npop: MIX with its re-draws); 2: MIX (v: 1 double).  One wave; dynamic LDS = smp2::WgBase.
template <int PHASE>
__global__ void __launch_bounds__(64) gdec_kernel(GDecState * __restrict__ st, const double * __restrict__ v, const Species sp, const uint32_t tm,
                                                  const int q, const int next, const uint32_t epoch, uint32_t * __restrict__ flag,
                                                  uint32_t * __restrict__ counters, double * __restrict__ taus)
{
  smp2::WgBase & wg = smp2::wg_base();
  const uint32_t lane = threadIdx.x & 63u;
  const int npop = sp.npop, nsp = sp.S;
  const double qnan = smp2::prog_qnan();
  {
    const uint32_t * src = reinterpret_cast<const uint32_t *>(&sp);
    uint32_t * dst = reinterpret_cast<uint32_t *>(&wg.sp);
    for (uint32_t i = lane; i < sizeof(Species)/4; i += 64u) dst[i] = src[i];
  }
  if (lane < (uint32_t)(3*MAXPOP)) wg.tau[lane] = taus[lane];
  if (lane < 16u) { wg.pf[lane] = st->pf[lane]; wg.rd[lane] = st->rd[lane]; }
  if (lane < 32u) wg.xtot[lane] = 0.0;
  if (lane == 0) { wg.dec.run_ok = st->run_ok; wg.dec.rd_mask = st->rd_mask; wg.dec.lnacc_theta = st->lnacc_theta; wg.dec.accm = 0; wg.dec.acc_step = 0; }
  smp2::wsync();
  smp2::Stream<true> g{(a00_rng_t)st->z};
  uint32_t c_prop = 0, c_acc = 0, c_gprop = 0, c_gacc = 0;
  unsigned long long pj4 = 0, pj5 = 0, pj6 = 0, pj7 = 0, pj8 = 0, pj9 = 0;
  double tau_w = st->tau_w, mix_c = st->mix_c, mix_lnc = st->mix_lnc;

  if (PHASE == 0)
  {
    // (a00_iterate: the first TAU's window comes before the THETA step's numbers in the global stream)
    const double pre = tm ? g.window() : 0.0;
    uint32_t slidem = 0; double tslide = 0;
    for (int p = 0; p < npop; ++p)
      if ((tm >> p) & 1u)
      {
        if (!(g.u() < sp.theta_slide_prob)) continue;
        slidem |= 1u << p;
        const double tn = reflect(wg.tau[MAXPOP + p] + sp.ft_theta*g.window(), 0.0, 999.0);
        if (p == (int)lane) tslide = tn;
      }
    // the sums: k_p, T_p of the populations of the mask, in order; an unusable term anywhere: every total is NaN
    bool bad = false;
    for (int p = 0; p < npop; ++p) bad = bad || v[4*p + 3] != 0.0;
    if (lane < (uint32_t)npop && ((tm >> lane) & 1u))
    {
      const int kx = __popc(tm & ((1u << lane) - 1u));
      wg.xtot[(2*kx) & 31] = bad ? qnan : (double)(long long)v[4*lane];
      wg.xtot[(2*kx + 1) & 31] = bad ? qnan : (double)gdec_join64(v[4*lane + 1], v[4*lane + 2])*(1.0/1099511627776.0);
    }
    smp2::wsync();
    g.r = smp2::prog_theta_decide((uint32_t)g.r, tm, slidem, tslide, 1, -1, 0);
    const uint32_t accm = wg.dec.accm;
    c_prop = (uint32_t)__popc(tm); c_acc = (uint32_t)__popc(accm);
    c_gprop = (uint32_t)__popc(tm & ~slidem); c_gacc = (uint32_t)__popc(accm & tm & ~slidem);
    pj8 = (unsigned long long)__popc(tm & slidem); pj9 = (unsigned long long)__popc(accm & tm & slidem);
    // the first TAU's window: drawn above when there is a theta to move, here otherwise (the host form's gs_prog_tau)
    tau_w = tm ? pre : g.window();
  }
  else if (PHASE == 1)
  {
    const int pq = sp.parent[q], cl = sp.left[q], cr = sp.right[q];
    const double tq_old = wg.tau[q], tq_lo = fmax(wg.tau[cl], wg.tau[cr]), tq_hi = pq >= 0 ? wg.tau[pq] : 999.0;
    const double tq_new = reflect(tq_old + sp.ft_tau*tau_w, tq_lo, tq_hi);
    double lnprior = 0;
    if (pq < 0 && sp.tau_alpha > 0) lnprior = (sp.tau_alpha - 1 - (nsp - 1) + 1)*log(tq_new/tq_old) - sp.tau_beta*(tq_new - tq_old);
    const bool bad = v[7] != 0.0;
    if (lane == 0) { wg.xtot[0] = v[0]; wg.xtot[1] = 0.0; }
    if (lane >= 2u && lane <= 4u) wg.xtot[lane] = bad ? qnan : (double)gdec_join64(v[2*lane - 3], v[2*lane - 2])*(1.0/1099511627776.0);
    smp2::wsync();
    g.r = smp2::prog_tau_decide((uint32_t)g.r, tm, q, 0, lnprior, false);
    const bool accept = wg.dec.acc_step != 0u;
    const uint32_t rd_mask = wg.dec.rd_mask;
    c_prop = 1; c_acc = accept ? 1u : 0u; pj4 = 1; pj5 = accept ? 1ull : 0ull;
    if (accept)
    {
      if (lane == 0) wg.tau[q] = tq_new;
      if (lane < 16u)
      {
        const smp2::Redraw r = wg.rd[lane];
        if ((rd_mask >> lane) & 1u) { wg.tau[MAXPOP + lane] = r.tn; wg.tau[2*MAXPOP + lane] = r.l2t; }
        const bool moved = ((tm >> lane) & 1u) && ((int)lane == q || (int)lane == cl || (int)lane == cr);
        if (moved) { smp2::PopFit & f = wg.pf[lane]; f.T = r.T; f.a = r.a; f.b = r.b; f.c = r.c; }
      }
    }
    else if (lane == 0) *flag = epoch;
    smp2::wsync();
    // the coming step's proposal: the next TAU's window, or MIX's log c and — nothing of it depends on the loci — its re-draws
    const double wprop = g.window();
    if (next < npop) tau_w = wprop;
    else
    {
      mix_lnc = sp.ft_mix*wprop; mix_c = exp(mix_lnc);
      g.r = smp2::prog_mix_redraw((uint32_t)g.r, tm, mix_c);
    }
  }
  else
  {
    double lnacc = (v[0] + 0.0) + (double)(nsp - 1)*mix_lnc;
    if (sp.tau_alpha > 0)
    {
      const double troot = wg.tau[npop - 1];
      lnacc += (sp.tau_alpha - 1)*mix_lnc - sp.tau_beta*(troot*mix_c - troot) - (double)(nsp - 2)*mix_lnc;
    }
    lnacc += wg.dec.lnacc_theta;
    const uint32_t rd_mask = wg.dec.rd_mask;
    const bool accept = g.accept(lnacc);
    c_prop = 1; c_acc = accept ? 1u : 0u; pj6 = 1; pj7 = accept ? 1ull : 0ull;
    if (accept)
    {
      if (lane < (uint32_t)npop) wg.tau[lane] *= mix_c;
      if (lane < 16u)
      {
        const smp2::Redraw r = wg.rd[lane];
        if ((rd_mask >> lane) & 1u) { wg.tau[MAXPOP + lane] = r.tn; wg.tau[2*MAXPOP + lane] = r.l2t; }
        smp2::PopFit & f = wg.pf[lane];
        if (lane < (uint32_t)npop && ((tm >> lane) & 1u)) { f.T = r.T; f.a = r.a; f.b = r.b; f.c = r.c; }
        else if (lane < (uint32_t)npop) f.T *= mix_c;
      }
    }
    else if (lane == 0) *flag = epoch;
  }
  smp2::wsync();
  // ---- the state for the launches that follow
  if (lane < (uint32_t)(3*MAXPOP)) taus[lane] = wg.tau[lane];
  if (lane < 16u) { st->pf[lane] = wg.pf[lane]; st->rd[lane] = wg.rd[lane]; }
  if (lane == 0)
  {
    st->z = (uint32_t)g.r; st->run_ok = wg.dec.run_ok; st->rd_mask = wg.dec.rd_mask; st->lnacc_theta = wg.dec.lnacc_theta;
    st->tau_w = tau_w; st->mix_c = mix_c; st->mix_lnc = mix_lnc;
    counters[0] += c_prop; counters[1] += c_acc; counters[2] += c_gprop; counters[3] += c_gacc;
    st->pj[4] += pj4; st->pj[5] += pj5; st->pj[6] += pj6; st->pj[7] += pj7; st->pj[8] += pj8; st->pj[9] += pj9;
  }
}

} // namespace gsm
