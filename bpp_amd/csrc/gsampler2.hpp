// gsampler2.hpp — the generic sampler's proposal steps (GAGE, GSPR, TAU, MIX) with a GROUP of lanes per locus, and the
// per-locus steps of an iteration as one launch for small JC69 sets.
//
// gsampler.hpp's gstep_kernel gives every locus ONE lane that runs the host driver's proposal code alone: ~3 500 dependent
// instructions, 24-50 us for a launch whatever the number of loci — the fixed cost of every step of configs 3 / 5, of a
// strong-scaling rank's share, of a composite's small parts.  Here a locus owns 2 NT lanes (16 for <= 8 tips, 32 for <= 16)
// and the proposal is sweep2.hpp's: the integer tree replicated in every lane's registers (byte arrays), lane i = node i,
// population i, branch i; loops over nodes and populations are ballots restricted to the group (propose_gage / propose_gspr
// of sweep2.hpp, the persistent kernel's own functions: same streams, same draws, same arithmetic as the host driver —
// tests/test_gpu_gsampler.py walks its trajectory); TAU and MIX are sweep2.hpp's step_locus.  Everything around the
// proposal is gstep_kernel's, statement for statement: the previous step is settled first (its decision, or the roll-back
// from the undo copy), the state a rejection comes back to is saved, the step is written as the records of the engine's
// kernels (StepRec / StepOp / MatRec2, or the OpDev ranges of the 20-state kernels and of the chain below), the tree goes
// back to HBM.  BPA_GS_DIFF=1 (gsampler_host.hpp) runs a step by both kernels from the same state and compares all of that.
// The other modes (settle + THETA statistics, start-up, the substitution-parameter moves) stay with gstep_kernel.
// 12-14 us a launch for <= 8 tips, 15-20 for <= 16 (profiles/r4/gstep2.txt): one wave's ~9 500 instructions.
#pragma once

namespace gsm2 {
using smp::MAXPOP; using smp::Op; using smp::make_op; using smp::nth_bit; using smp::rndu;

template <int NT> struct LocLDS
{
  double time[2*NT];
  double contrib[2*NT], contrib_new[2*NT];
  alignas(16) double pmblk[48];           // fuse_pm: the parameter block of the locus's first substitution matrix (46 doubles)
};

// what changes from step to step (a launch takes it from its arguments, the chain kernel below counts it itself)
struct StepCtl { uint32_t k, pend, refresh_logpr, fmt20, pend_mode; };     // pend_mode: the kind of the step being settled (0: a gene-node age move)

template <int NT> struct Step2LDS
{
  LocLDS<NT> loc[smp2::Cfg<NT>::LPW];
  double tau[3*MAXPOP];
  double lograt[smp2::Cfg<NT>::NN*smp2::Cfg<NT>::NN];
  uint32_t anc[16];
};

// one step of the loci of ONE wave (lane = threadIdx.x & 63; group `slot` of the wave holds locus i when valid).  WAVE_ONLY:
// the caller's other waves do not take part — the hand-overs through LDS are the wave's own (chain kernel)
template <uint32_t MODE, int NT, bool WAVE_ONLY, bool BPP = false>
__device__ __forceinline__ void gstep2_body(const gsm::GArgs & A, const StepCtl & C, Step2LDS<NT> & SH, const uint32_t lane, const uint32_t i, const bool valid)
{
  static_assert(MODE <= 3, "GAGE, GSPR, TAU, MIX");
  using Cf = smp2::Cfg<NT>;
  constexpr int G = Cf::G, NN = Cf::NN, W = Cf::W;
  LocLDS<NT> * s_loc = SH.loc;
  double * s_tau = SH.tau, * s_lograt = SH.lograt;
  uint32_t * s_anc = SH.anc;
#ifdef GS2_PROF
  uint64_t tp_[9]; tp_[0] = wall_clock64();
#define GS2_T(k_) tp_[k_] = wall_clock64()
#else
#define GS2_T(k_)
#endif
  const int li = (int)(lane & (uint32_t)(G - 1)); const uint32_t gbase = lane - (uint32_t)li, slot = lane/(uint32_t)G;
  const smp::Species & SP = A.sp;
  const int npop = SP.npop;
  // ---- every load of the step, issued together: ONE round trip to HBM where the species tree, the tree, the last step's
  // decision inputs and the undo copy (a rejection is the common case) one after the other were four
  const uint32_t ic = valid ? i : A.i0;                         // (an idle group reads a locus that exists; nothing of it is used)
  const double tau_r = lane < (uint32_t)(3*MAXPOP) ? A.taus[lane] : 0.0;
  constexpr int NLR = MODE == 1 ? (NN*NN + 63)/64 : 1;
  double lr_[NLR];
  if (MODE == 1)
  {
#pragma unroll
    for (int k = 0; k < NLR; ++k) { const uint32_t q = lane + 64u*(uint32_t)k; lr_[k] = q < (uint32_t)(NN*NN) ? A.lograt[(q/NN)*gsm::NN + q % NN] : 0.0; }
  }
  const gsm::GTree & g = A.trees[ic];
  const gsm::GTree & ud = A.undo[ic];
  uint32_t gw[4][W], uw[4][W];
#pragma unroll
  for (int k = 0; k < W; ++k)
  {
    gw[0][k] = reinterpret_cast<const uint32_t *>(g.left)[k];   gw[1][k] = reinterpret_cast<const uint32_t *>(g.right)[k];
    gw[2][k] = reinterpret_cast<const uint32_t *>(g.parent)[k]; gw[3][k] = reinterpret_cast<const uint32_t *>(g.pop)[k];
    uw[0][k] = reinterpret_cast<const uint32_t *>(ud.left)[k];   uw[1][k] = reinterpret_cast<const uint32_t *>(ud.right)[k];
    uw[2][k] = reinterpret_cast<const uint32_t *>(ud.parent)[k]; uw[3][k] = reinterpret_cast<const uint32_t *>(ud.pop)[k];
  }
  const int g_clv = g.clv[li], g_pm = g.pmat[li], u_clv = ud.clv[li], u_pm = ud.pmat[li];
  const double g_time = g.time[li], u_time = ud.time[li];
  const int g_root = g.root, g_tips = g.tips, u_root = ud.root;
  smp2::Stream<BPP> rng{g.rng};                                // (BPP: the reference's generator, Bactrian-Laplace windows, its acceptance rule)
  double lnl_cur = g.lnl, logpr_cur = g.logpr;
  uint32_t nprop = g.proposals, nacc = g.accepted, w_nupd = g.work_nupd, w_nbr = g.work_nbr, w_nev = g.work_neval, pj_gage = g.pj_gage, pj_gage_acc = g.pj_gage_acc, pj_gspr = g.pj_gspr, pj_gspr_acc = g.pj_gspr_acc;
  const gsm::GLocus L = A.loc[ic];
  const int gl_i = li < MAXPOP ? (int)L.gl[li] : 0;
  const uint32_t d_active = A.active[ic], d_flag = *A.flag;
  // (the program's moves decided on the device, gdec_kernel: this step's window variate / factor lies in its state)
  const double d_tau_w = (MODE == 2 && A.dstep) ? A.dstep->tau_w : A.tau_w;
  const double d_mix_c = (MODE == 3 && A.dstep) ? A.dstep->mix_c : A.mix_c, d_mix_lnc = (MODE == 3 && A.dstep) ? A.dstep->mix_lnc : A.mix_lnc;
  const double d_lnl = A.lnl_new[ic], d_logpr = A.logpr_new[ic], d_hast = A.hast[ic];

  if (lane < (uint32_t)(3*MAXPOP)) s_tau[lane] = tau_r;
  if (lane < 16u) s_anc[lane] = lane < (uint32_t)MAXPOP ? (uint32_t)SP.anc[lane] : 0u;
  if (MODE == 1)
  {
#pragma unroll
    for (int k = 0; k < NLR; ++k) { const uint32_t q = lane + 64u*(uint32_t)k; if (q < (uint32_t)(NN*NN)) s_lograt[q] = lr_[k]; }
  }
  smp2::PopLane pl;
  {
    const int lp = li < MAXPOP ? li : 0;
    pl.parent = li < npop ? (int)SP.parent[lp] : -1;
    pl.anc = li < npop ? (uint32_t)SP.anc[lp] : 0u;
    uint32_t below = 0;
    for (int q2 = 0; q2 < npop; ++q2) if (q2 != li && (((uint32_t)SP.anc[q2] >> li) & 1u)) below |= 1u << q2;
    pl.below = li < npop ? below : 0u;
  }
  if (WAVE_ONLY) smp2::wsync(); else __syncthreads();
  {
    const int lp = li < MAXPOP ? li : 0;
    pl.tau = s_tau[lp]; pl.theta = s_tau[MAXPOP + lp]; pl.l2t = s_tau[2*MAXPOP + lp];
    pl.ptau = pl.parent >= 0 ? s_tau[pl.parent] : -1.0;
  }
  LocLDS<NT> & S = s_loc[slot];
  // fuse_pm (step 4 fills the fresh branches' P-matrices): what that needs of the locus is asked for NOW and arrives while
  // the proposal is made — the lane's category rate, the matrix index, the first matrix's parameter block (dealt over the
  // group's lanes, handed over through LDS), the P-matrix buffer
  constexpr uint32_t PMB = 46;                                  // par_matrix_stride(4)
  constexpr int NPF = (int)((PMB + (uint32_t)G - 1u)/(uint32_t)G);
  double pf_pm[NPF], pf_rate = 0; uint32_t pf_mi = 0, pf_model = 0, pf_pstride = 0; double * pf_pmat = nullptr;
  if (A.fuse_pm)
  {
    const uint32_t R = L.R, k0 = (uint32_t)li % R;
    pf_rate = L.par[par_rates(R) + k0]; pf_mi = (uint32_t)L.par[par_param_idx(R) + k0];
#pragma unroll
    for (int q = 0; q < NPF; ++q) { const uint32_t x = (uint32_t)li + (uint32_t)(q*G); pf_pm[q] = L.par[par_matrix(R, 4, 0) + (x < PMB ? x : 0u)]; }
    const SlotStatic & M = A.slot_tab[L.slot];
    pf_pmat = M.pmat; pf_model = M.model; pf_pstride = M.pstride;
  }
  GS2_T(1);

  // ---- 1. settle the step whose evaluation just finished (gstep_kernel's step 1): the decision, from registers
  bool restore_par = false, back = false;
  if (valid && C.pend)
  {
    if (C.pend == 1)
    {
      if (d_active)
      {
        const double lnacc = (d_logpr - logpr_cur) + (d_lnl - lnl_cur) + d_hast;
        ++nprop;
        const bool acc_ = rng.accept(lnacc);
        if (C.pend_mode == 0) { ++pj_gage; pj_gage_acc += acc_ ? 1u : 0u; } else if (C.pend_mode == 1) { ++pj_gspr; pj_gspr_acc += acc_ ? 1u : 0u; }
        if (acc_) { lnl_cur = d_lnl; logpr_cur = d_logpr; ++nacc; }
        else back = true;
      }
    }
    else if (C.pend == 2)
    {
      if (d_flag == A.epoch) back = true;
      else { logpr_cur = d_logpr; if (d_active) lnl_cur = d_lnl; }
    }
    else if (C.pend == 4)
    {
      if (d_active)
      {
        const double lnacc = (d_lnl - lnl_cur) + d_hast;
        ++nprop;
        if (rng.accept(lnacc)) { lnl_cur = d_lnl; ++nacc; }
        else
        {
          back = true;
          if (li == 0)
          {
            double * m = A.sm + (size_t)i*11;
            if (A.pend_mode == 8) m[10] = A.sm_old[2*i];
            else { double * v = A.pend_mode == 6 ? m : m + 4; const int ref = A.pend_mode == 6 ? 3 : 1; v[A.pend_k] = A.sm_old[2*i]; v[ref] = A.sm_old[2*i + 1]; }
          }
          restore_par = true;
        }
      }
    }
    else { lnl_cur = d_lnl; logpr_cur = d_logpr; }
    // (every lane of a group takes the same branch: the conditions are the locus's)
  }
  GS2_T(2);
  // ---- the tree as settled: the arriving one, or the undo copy (the tree of an idle group: all -1, two tips)
  smp2::GTree<NT> T;
#pragma unroll
  for (int k = 0; k < W; ++k)
  {
    T.left.w[k]   = !valid ? 0xffffffffu : back ? uw[0][k] : gw[0][k];
    T.right.w[k]  = !valid ? 0xffffffffu : back ? uw[1][k] : gw[1][k];
    T.parent.w[k] = !valid ? 0xffffffffu : back ? uw[2][k] : gw[2][k];
    T.pop.w[k]    = !valid ? 0xffffffffu : back ? uw[3][k] : gw[3][k];
  }
  T.root = !valid ? 0 : back ? u_root : g_root; T.tips = valid ? g_tips : 2;
  {
    const int n0 = 2*T.tips - 1;
    T.cf = smp2::gballot<G>(valid && li >= T.tips && li < n0 && (back ? u_clv : g_clv) != li, gbase);
    T.pf = smp2::gballot<G>(valid && li < n0 && (back ? u_pm : g_pm) != li, gbase);
    S.time[li] = valid && li < n0 ? (back ? u_time : g_time) : 0.0;
  }
  smp2::wsync();
  if (valid && restore_par && li == 0) gsm::write_par(L.par, L.R, A.pend_mode, A.sm + (size_t)i*11);

  GS2_T(3);
  const int tips = T.tips, n = 2*tips - 1;
  const bool inner_i = li >= tips && li < n;
  // ---- the density's pieces, lane li = population li (sweep2.hpp's density_counts / density_term, ages from LDS)
  uint32_t mynodes = 0; int mync = 0, mynin = 0;
  auto density_counts = [&]()
  {
    const int pop_i = T.pop[li];
    int below = 0;
    for (int q = 0; q < npop; ++q)
    {
      const uint32_t m = smp2::gballot<G>(inner_i && pop_i == q, gbase);
      if (li == q) mynodes = m;
      below += ((pl.below >> q) & 1u) ? __popc(m) : 0;
    }
    mync = __popc(mynodes);
    mynin = gl_i - below;
  };
  double my_t2h = 0;                                             // the T2h of the lane's population, as density_term last left it
  auto density_term = [&]() -> double
  {
    uint32_t nodes = mynodes;
    const int ncoal = mync, nin = mynin;
    int steps = ncoal + (pl.ptau >= 0 ? 1 : 0);
    if (nin == steps) --steps;
    double T2h = 0, prev = pl.tau;
    int nn = nin;
    for (int k = 0; k < steps; ++k, --nn)
    {
      double tk = pl.ptau;
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
    if (ncoal) c += ncoal*pl.l2t;
    if (T2h) c -= T2h/(pl.theta*1.0);
    my_t2h = T2h;
    return c;
  };
  // the current tree's terms (the tree arrives with their sum only); THETA moved the thetas since that sum was stored: again
  density_counts();
  if (li < npop) S.contrib[li] = density_term();
  smp2::wsync();
  if (valid && C.refresh_logpr)
  {
    double lp = 0;
    for (int p = 0; p < npop; ++p) lp += S.contrib[p];
    logpr_cur = lp;
  }

  // ---- 2. the state a rejection comes back to
  if (valid)
  {
    gsm::GTree & u = A.undo[i];
    {
      uint32_t wl_ = 0, wr_ = 0, wp_ = 0, wq_ = 0;
#pragma unroll
      for (int k = 0; k < W; ++k) if (li == k) { wl_ = T.left.w[k]; wr_ = T.right.w[k]; wp_ = T.parent.w[k]; wq_ = T.pop.w[k]; }
      if (li < W)
      {
        reinterpret_cast<uint32_t *>(u.left)[li] = wl_; reinterpret_cast<uint32_t *>(u.right)[li] = wr_;
        reinterpret_cast<uint32_t *>(u.parent)[li] = wp_; reinterpret_cast<uint32_t *>(u.pop)[li] = wq_;
      }
    }
    if (li < n) { u.time[li] = S.time[li]; u.clv[li] = (int8_t)T.cidx(li); u.pmat[li] = (int8_t)T.pidx(li); }
    if (li == 0) u.root = T.root;
  }

  GS2_T(4);
  // ---- 3. propose (sweep2.hpp), the proposed tree's density, buffer toggles, the node updates children first
  smp2::Prop pr{0, 0, 0, 0.0};
  bool ok = false;
  const smp2::GTree<NT> U = T;
  const double tsave = S.time[li];
  int above = 0, below = 0;
  double lminf = 0, lmaxf = 0;
  if (MODE <= 1)
  {
    if (valid)
      ok = MODE == 0 ? smp2::propose_gage<NT, BPP>(T, rng, S.time, (int)C.k, pl, s_anc, s_tau, SP.ft_gage, li, gbase, pr)
                     : smp2::propose_gspr<NT, BPP>(T, rng, S.time, (int)C.k, pl, gl_i, s_anc, s_tau, s_lograt, SP.ft_gspr, li, gbase, pr);
  }
  else if (valid)
  {
    // an all-loci step (sweep2.hpp's step_locus): the proposed species tree in the lanes' registers, every population's term
    pr.chain = (1u << npop) - 1u;
    if (MODE == 2)
    {
      // TAU q (tau_step of a00_driver.c): the gene nodes of q and its children between the bounds ride the rubber band
      const int q = (int)C.k, pq = SP.parent[q], cl = SP.left[q], cr = SP.right[q];
      const double tq_old = s_tau[q], tq_lo = fmax(s_tau[cl], s_tau[cr]), tq_hi = pq >= 0 ? s_tau[pq] : 999.0;
      const double tnew = smp::reflect(tq_old + SP.ft_tau*(A.bpp ? d_tau_w : A.tau_u - 0.5), tq_lo, tq_hi);
      const double minf = (tnew - tq_lo)/(tq_old - tq_lo), maxf = (tnew - tq_hi)/(tq_old - tq_hi);
      lminf = log(minf); lmaxf = log(maxf);
      if (li == q) pl.tau = tnew;
      if (pl.parent == q) pl.ptau = tnew;
      const int pk = T.pop[li];
      const bool moved = inner_i && (pk == q || pk == cl || pk == cr) && !(tsave < tq_lo || tsave > tq_hi);
      const bool up = moved && tsave >= tq_old;
      if (moved) S.time[li] = up ? tq_hi + maxf*(tsave - tq_hi) : tq_lo + minf*(tsave - tq_lo);
      const uint32_t mm = smp2::gballot<G>(moved, gbase);
      above = __popc(smp2::gballot<G>(up, gbase)); below = __popc(mm) - above;
      const int par = T.parent[li];
      pr.brm = smp2::gballot<G>(li < n && par >= 0 && (((mm >> li) & 1u) || ((mm >> (par & 31)) & 1u)), gbase);
      uint32_t m = mm;
      const int l = T.left[li], r = T.right[li];
#pragma unroll
      for (int d = 0; d < NT - 2; ++d) m |= smp2::gballot<G>(inner_i && (((m >> (l & 31)) & 1u) || ((m >> (r & 31)) & 1u)), gbase);
      pr.ndm = m;
    }
    else
    {
      // mixing (mix_step of a00_driver.c): every tau, every age times c
      pl.tau *= d_mix_c;
      if (pl.parent >= 0) pl.ptau *= d_mix_c;
      if (inner_i) S.time[li] = tsave*d_mix_c;
      pr.ndm = smp2::gballot<G>(inner_i, gbase);
      pr.brm = smp2::gballot<G>(li < n && (int)T.parent[li] >= 0, gbase);
    }
    ok = pr.ndm != 0;                      // no gene node moves here: only the density changes
  }
  GS2_T(5);
  Op opw[NT - 1]; int nops = 0;
  for (int k = 0; k < NT - 1; ++k) opw[k] = 0;
  double lp_new = logpr_cur;
  if (ok || (MODE >= 2 && valid))
  {
    smp2::wsync();
    density_counts();
    if ((pr.chain >> li) & 1u) S.contrib_new[li] = density_term();
    if (ok)
    {
      T.pf ^= pr.brm; T.cf ^= pr.ndm;
      const double myage = S.time[li];
      nops = __popc(pr.ndm);
      int rank = 0;
      for (uint32_t m = pr.ndm; m; m &= m - 1)
      {
        const int x = __ffs(m) - 1; const double tx = S.time[x];
        rank += (tx < myage || (tx == myage && x < li)) ? 1 : 0;
      }
      const bool mine = (pr.ndm >> li) & 1u;
#pragma unroll
      for (int k = 0; k < NT - 1; ++k)
        if (k < nops)
        {
          const int x = __ffs(smp2::gballot<G>(mine && rank == k, gbase)) - 1;
          const int l = T.left[x], r = T.right[x];
          opw[k] = make_op(T.cidx(x), T.cidx(l), T.pidx(l), T.cidx(r), T.pidx(r));
        }
    }
    smp2::wsync();
    double lp = 0;
    for (int p = 0; p < npop; ++p) lp += ((pr.chain >> p) & 1u) ? S.contrib_new[p] : S.contrib[p];
    lp_new = lp;
  }
  else if (valid) { T = U; S.time[li] = tsave; }
  smp2::wsync();
  if (MODE == 2 && valid && A.prog)
  {
    // the program's rubber band re-draws the thetas of q and its two children from (k, sum of the T2h AFTER the move)
    const int q = (int)C.k, cl = SP.left[q], cr = SP.right[q];
    const int j = li == q ? 0 : li == cl ? 1 : li == cr ? 2 : -1;
    if (j >= 0) A.t2h3[(size_t)3*i + j] = lp_new == lp_new ? my_t2h : __longlong_as_double(0x7ff8000000000000ll);
  }
  if (valid && li == 0)
  {
    if (MODE <= 1) { if (ok) { A.logpr_new[i] = lp_new; A.hast[i] = pr.hast; } }
    else
    {
      A.logpr_new[i] = lp_new;
      // p_delta of the host driver; the program's moves (A.prog): the densities' change over all loci follows from the sums of
      // k and T2h with the re-drawn thetas, on the host (tau_step / mix_step of a00_driver.c) — the loci bring their Jacobian only
      if (A.prog) A.delta[i] = MODE == 2 ? below*lminf + above*lmaxf : (double)(tips - 1)*d_mix_lnc;
      else A.delta[i] = MODE == 2 ? ((lp_new - logpr_cur) + below*lminf) + above*lmaxf : (lp_new - logpr_cur) + (double)(tips - 1)*d_mix_lnc;
      A.lnl_cur[i] = lnl_cur;
    }
    if (ok) { w_nupd += (uint32_t)nops; w_nbr += (uint32_t)__popc(pr.brm); ++w_nev; }
    A.active[i] = ok ? 1 : 0;
  }

  GS2_T(6);
  // ---- 4. the step's records for the engine's kernels (gstep_kernel's step 5): fresh branch j by lane j, node update k by lane k
  const uint32_t brm = ok ? pr.brm : 0u;
  const int nbr = __popc(brm);
  if (valid && C.fmt20)
  {
    const uint32_t e0 = i*A.maxmat, o0 = i*A.maxops20;
    if (li < nbr)
    {
      const int x = nth_bit(brm, li);
      A.mat_task20[e0 + li] = i; A.mat_pm20[e0 + li] = (uint32_t)T.pidx(x);
      A.mat_length[e0 + li] = (S.time[(int)T.parent[x]] - S.time[x])*1.0;                 // rate_mui = 1 (locus.c:2350)
    }
    else if ((uint32_t)li < A.maxmat) A.mat_task20[e0 + li] = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < NT - 1; ++k)
      if (k < nops && li == k)
      {
        const Op w = opw[k];
        OpDev q;
        q.parent_clv = (uint32_t)(w & 255u); q.left_clv = (uint32_t)((w >> 8) & 255u); q.left_pmatrix = (uint32_t)((w >> 16) & 255u);
        q.right_clv = (uint32_t)((w >> 24) & 255u); q.right_pmatrix = (uint32_t)((w >> 32) & 255u);
        q.parent_scaler = q.left_scaler = q.right_scaler = BPA_SCALE_BUFFER_NONE;
        A.ops20[o0 + k] = q;
      }
    if (li == 0) { A.op_rng20[2*i] = o0; A.op_rng20[2*i + 1] = o0 + (uint32_t)nops; A.root20[i] = (uint32_t)T.cidx(T.root); }
  }
  else if (valid)
  {
    uint4 * rec = A.recs2 + (size_t)L.slot*A.units;
    MatRec2 * m2 = A.mat2 + (size_t)L.slot*A.maxmat;
    double * ml = A.mat_length + (size_t)L.slot*A.maxmat;
    const uint32_t e0 = L.slot*A.maxmat;
    if (li == 0)
    {
      StepRec h{};
      h.task = ok ? i : 0xffffffffu; h.pat_off = L.pat_off;
      h.root_clv = (uint8_t)T.cidx(T.root); h.root_scaler = (int8_t)BPA_SCALE_BUFFER_NONE; h.nops = (uint8_t)nops;
      *reinterpret_cast<StepRec *>(rec) = h;
    }
    if (li < nbr)
    {
      const int x = nth_bit(brm, li);
      m2[li] = MatRec2{L.slot, (uint32_t)T.pidx(x)};
      ml[li] = (S.time[(int)T.parent[x]] - S.time[x])*1.0;                                // rate_mui = 1 (locus.c:2350)
    }
    else if ((uint32_t)li < A.maxmat) m2[li] = MatRec2{0xffffffffu, 0u};                    // entries of the last step this one does not use
    if (A.fuse_pm)
    {
      // the fresh branches' P-matrices, one (branch, rate category) per lane of the group and round: the arithmetic of
      // pmatrix_s4_dense_kernel (pmatrix_s4_core), the launch it was is gone.  The host leaves this off while a
      // substitution-parameter step is pending (its roll-back and the eigensystems' refresh come first: gs_step)
#pragma unroll
      for (int q = 0; q < NPF; ++q) { const uint32_t x = (uint32_t)li + (uint32_t)(q*G); if (x < PMB) S.pmblk[x] = pf_pm[q]; }
      smp2::wsync();
      const uint32_t R = L.R, k0 = (uint32_t)li % R;
      for (uint32_t q = (uint32_t)li; q < (uint32_t)nbr*R; q += (uint32_t)G)
      {
        const uint32_t j = q/R, k = q - j*R;
        const int x = nth_bit(brm, (int)j);
        double * dst = pf_pmat + (size_t)T.pidx(x)*R*pf_pstride;
        const double t = (S.time[(int)T.parent[x]] - S.time[x])*1.0;
        const double rate = k == k0 ? pf_rate : L.par[par_rates(R) + k];
        const uint32_t mi = pf_model == 0 ? 0u : k == k0 ? pf_mi : (uint32_t)L.par[par_param_idx(R) + k];
        double pmv[PMB];                                        // (constant indices below: registers)
        if (mi == 0)
        {
#pragma unroll
          for (uint32_t u = 0; u < PMB; u += 2) { const d2v_t v = *(const __attribute__((address_space(3))) d2v_t *)(&S.pmblk[u]); pmv[u] = v.x; pmv[u + 1] = v.y; }
        }
        else
        {
          const double * gp = L.par + par_matrix(R, 4, mi);
#pragma unroll
          for (uint32_t u = 0; u < PMB; ++u) pmv[u] = *(const __attribute__((address_space(1))) double *)(gp + u);
        }
        pmatrix_s4_core(dst, pf_model, rate, pmv, t, k);
      }
    }
#pragma unroll
    for (int k = 0; k < NT - 1; ++k)
      if (k < nops && li == k)
      {
        const Op w = opw[k];
        StepOp q{};
        q.parent_clv = (uint8_t)(w & 255u); q.left_clv = (uint8_t)((w >> 8) & 255u); q.left_pmatrix = (uint8_t)((w >> 16) & 255u);
        q.right_clv = (uint8_t)((w >> 24) & 255u); q.right_pmatrix = (uint8_t)((w >> 32) & 255u);
        q.parent_scaler = q.left_scaler = q.right_scaler = (int8_t)BPA_SCALE_BUFFER_NONE;
        q.left_e = q.right_e = -1;
        uint32_t j = 0;
        for (uint32_t m = brm; m; m &= m - 1, ++j)
        {
          const uint32_t pm = (uint32_t)T.pidx(__ffs(m) - 1);
          if (pm == q.left_pmatrix)  q.left_e = (int32_t)(e0 + j);
          if (pm == q.right_pmatrix) q.right_e = (int32_t)(e0 + j);
        }
        *reinterpret_cast<StepOp *>(rec + 1 + k) = q;
      }
  }

  GS2_T(7);
  // ---- 5. store
  if (valid)
  {
    gsm::GTree & g = A.trees[i];
    {
      uint32_t wl_ = 0, wr_ = 0, wp_ = 0, wq_ = 0;
#pragma unroll
      for (int k = 0; k < W; ++k) if (li == k) { wl_ = T.left.w[k]; wr_ = T.right.w[k]; wp_ = T.parent.w[k]; wq_ = T.pop.w[k]; }
      if (li < W)
      {
        reinterpret_cast<uint32_t *>(g.left)[li] = wl_; reinterpret_cast<uint32_t *>(g.right)[li] = wr_;
        reinterpret_cast<uint32_t *>(g.parent)[li] = wp_; reinterpret_cast<uint32_t *>(g.pop)[li] = wq_;
      }
    }
    if (li < n) { g.time[li] = S.time[li]; g.clv[li] = (int8_t)T.cidx(li); g.pmat[li] = (int8_t)T.pidx(li); }
    if (li == 0)
    {
      g.rng = rng.r; g.root = T.root; g.lnl = lnl_cur; g.logpr = logpr_cur; g.proposals = nprop; g.accepted = nacc;
      g.work_nupd = w_nupd; g.work_nbr = w_nbr; g.work_neval = w_nev; g.pj_gage = pj_gage; g.pj_gage_acc = pj_gage_acc; g.pj_gspr = pj_gspr; g.pj_gspr_acc = pj_gspr_acc;
    }
  }
  if (WAVE_ONLY) smp2::wsync();                  // (the next step of a chain reuses the LDS block)
#ifdef GS2_PROF
  __builtin_amdgcn_s_waitcnt(0); GS2_T(8);
  if (MODE <= 1 && valid && li == 0) { A.delta[i] = (double)(tp_[1 + (i & 7u)] - tp_[0]);      // (GAGE / GSPR do not use the two arrays)
    A.lnl_cur[i] = (double)(tp_[(i & 1u) ? 8 : 0] & 0xffffffffffffull); }
#endif
}

template <uint32_t MODE, int NT, bool BPP = false>
__global__ void __launch_bounds__(64) gstep2_kernel(const gsm::GArgs A)
{
  __shared__ Step2LDS<NT> SH;
  constexpr uint32_t LPW = (uint32_t)smp2::Cfg<NT>::LPW, G = (uint32_t)smp2::Cfg<NT>::G;
  const uint32_t i = A.i0 + blockIdx.x*LPW + threadIdx.x/G;
  const StepCtl C{MODE >= 2 ? A.tau_q : A.k, A.pend, A.refresh_logpr, A.fmt20, A.pend_mode};
  gstep2_body<MODE, NT, false, BPP>(A, C, SH, threadIdx.x, i, i < A.iend);
}


// ---- the per-locus steps of an iteration as ONE launch ------------------------------------------------------------------------
// GAGE and GSPR of a locus need nothing from other loci (threads.c:87-200: a worker walks its loci's proposals without a
// barrier), so a workgroup that owns the locus can walk all of them: propose (the lane group above, in the workgroup's first
// wave), the step's P-matrices, node updates and pattern terms by the workgroup's threads with the engine's own device
// functions (pmatrix_s4_entry, walk_s4, lnl_reduce_wave of kernels.hpp: the arithmetic of every 4-state path, one lane per
// pattern — the records are written in the OpDev form those read), the decision at the head of the next step.  No barrier
// between workgroups, so no residency condition; the last step's decision is left pending exactly as the launch loop leaves
// it.  What this buys is the launches: a small set (a strong-scaling rank's share, a composite's small part, config 5) pays
// 12-30 us per launch three times per step, whatever the work.
struct GChain { uint32_t ngage, ngspr, pend, refresh_logpr; };
constexpr uint32_t GCHAIN_THREADS = 64;          // one wave per locus: the lane group proposes, the 64 lanes share the patterns

template <int NT, bool BPP = false>
__global__ void __launch_bounds__(GCHAIN_THREADS) gchain_kernel(const gsm::GArgs A, const PlanDev P, const GChain ch)
{
  __shared__ Step2LDS<NT> SH;
  constexpr uint32_t G = (uint32_t)smp2::Cfg<NT>::G;
  const uint32_t i = A.i0 + blockIdx.x, tid = threadIdx.x;
  StepCtl C{0, ch.pend, ch.refresh_logpr, 1u, A.pend_mode};
  const LocusDev & L = P.loci[P.task_locus[i]];
  const uint32_t np = L.np, R = L.rate_cats, e0 = i*A.maxmat, poff = P.task_pat_off[i];
  const uint32_t nsteps = ch.ngage + ch.ngspr;
  for (uint32_t st = 0; st < nsteps; ++st)
  {
    const bool gage = st < ch.ngage;
    C.k = gage ? st : st - ch.ngage;
    if (gage) gstep2_body<0, NT, true, BPP>(A, C, SH, tid, i, tid < G);
    else      gstep2_body<1, NT, true, BPP>(A, C, SH, tid, i, tid < G);
    C.pend = 1u; C.refresh_logpr = 0u; C.pend_mode = gage ? 0u : 1u;
    __syncthreads();                                  // the step's records are out (written and read on this CU)
    if (A.active[i])
    {
      for (uint32_t q = tid; q < A.maxmat*R; q += GCHAIN_THREADS) pmatrix_s4_entry(P, e0 + q/R, q % R);
      __syncthreads();
      for (uint32_t n = tid; n < np; n += GCHAIN_THREADS) P.site_term[poff + n] = walk_s4(P, i, n, L, true);
      __syncthreads();
      lnl_reduce_wave(P, i, tid);
    }
    __syncthreads();
  }
}

} // namespace gsm2
