// gsampler_host.hpp — host side of the generic device-resident sampler (gsampler.hpp): uploads, the launch loop of an
// iteration, downloads.  Included at the end of sampler.hpp (same translation unit as engine.hip).
#pragma once

// the substitution-parameter moves' device tables (every locus's frequencies | exchangeabilities | alpha, the roll-back
// pairs, the loci's ids): made when a window width first becomes positive — the widths themselves travel with every
// launch, so switching a move on or changing a width in mid-run (BPP's burn-in finetune adjustment) touches no state
// BPA_S20_KERNEL=pipe: round 4's 20-state node-update kernel (a workgroup barrier per update) instead of partials_lnl_wave20_kernel (A/B)
// BPA_S20_KERNEL=wave2: partials_lnl_wave20_kernel with two patterns per lane (tiles of 128 patterns, one wave per SIMD)
// (the A/B forms of the 20-state node-update kernel inside the sampler: an experimental build's; the default build launches
//  partials_lnl_wave20_kernel)
#ifdef BPA_EXPERIMENTAL
static unsigned gs_tile20() { static const unsigned v = (getenv("BPA_S20_KERNEL") && std::string(getenv("BPA_S20_KERNEL")) == "wave2") ? 128u : 64u; return v; }
static bool gs_waverl() { static const bool v = getenv("BPA_S20_KERNEL") && std::string(getenv("BPA_S20_KERNEL")) == "waverl"; return v; }
static bool gs_pipe20() { static const bool v = getenv("BPA_S20_KERNEL") && std::string(getenv("BPA_S20_KERNEL")) == "pipe"; return v; }
#else
static constexpr unsigned gs_tile20() { return 64u; }
static constexpr bool gs_waverl() { return false; }
static constexpr bool gs_pipe20() { return false; }
#endif

static int gs_subst_ready(bpa_sampler * s)
{
  if (!(s->g_ft[0] > 0 || s->g_ft[1] > 0 || s->g_ft[2] > 0) || s->g_sm.p) return 1;
  const unsigned T = s->nloci;
  if (s->g_alljc) return fail("bpa_sampler: the substitution-parameter moves need loci with an eigendecomposition (GTR) and several rate categories");
  if (s->g_s20) return fail("bpa_sampler: no substitution-parameter moves for amino-acid loci (the empirical models have none; the alpha move is 4-state only here)");
  if (s->g_sm_host.size() != (size_t)T*11) return fail("bpa_sampler: call bpa_sampler_set_subst_model for every locus before the substitution-parameter moves");
  std::vector<uint32_t> ids(T);
  for (unsigned i = 0; i < T; ++i) ids[i] = s->loci[i]->id;
  // (no move has run yet: the host copy is what bpa_sampler_set_subst_model left)
  if (!upload(s->g_sm, s->g_sm_host.data(), (size_t)T*11) || !s->g_sm_old.reserve((size_t)T*2) || !upload(s->g_ids, ids.data(), T)) return 0;
  return 1;
}

// K6 for every locus of the sampler from the values now in its parameter block (pll_update_eigen, locus.c:2462-2476)
static int gs_refresh_eigen(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  if (!s->g_eigen_dirty) return 1;
  if (!s->g_ids.p)
  {
    std::vector<uint32_t> ids(s->nloci);
    for (unsigned i = 0; i < s->nloci; ++i) ids[i] = s->loci[i]->id;
    if (!upload(s->g_ids, ids.data(), s->nloci)) return 0;
  }
  // the generic sampler's loci are 4-state; forked: each half on its own stream
  for (unsigned h = 0; h < (s->g_forked ? s->g_np : 1u); ++h)
  {
    const unsigned lo = s->g_forked ? s->g_pi[h] : 0u, hi = s->g_forked ? s->g_pi[h + 1] : s->nloci;
    hipLaunchKernelGGL(eigen_kernel<4>, dim3((hi - lo + 63)/64), dim3(64), 0, h ? s->g_st[h] : e->stream, e->d_loci.p, s->g_ids.p + lo, (uint32_t)(hi - lo));
    s->launches++;
  }
  HIPCHK(hipGetLastError());
  s->g_eigen_dirty = false;
  return 1;
}

// The per-locus steps (gene-tree ages, SPR, the substitution-parameter moves) of the two halves of the loci are
// independent chains of launches — propose, P-matrices, node updates + lnL, settle-and-propose ... — so they go to two
// streams: one half's (latency-bound, 16-wave) step kernel runs under the other half's likelihood kernels, which are
// throughput-bound (config 3: 88 us for all loci, 48 us for half of them).  Measured on config 3: 175.6 -> 185.4 it/s with
// the streams left to themselves — they settle into running nearly in phase, and the step kernel under load takes 30-80 us
// instead of 29; making the halves' likelihood launches take turns through a pair of events (before the P-matrix
// launch: 169 it/s, before the node-update launch: 178) or starting the second stream half a step late (184-185) was no
// better, and neither was one stream for all likelihood launches + one for all proposal launches with an event each way per
// half (166: a cross-stream event costs ~11 us between the kernels it separates), so there are no events between the halves.  All-loci steps (THETA, TAU, MIX, the downloads) run on the engine's
// stream over all loci after a join; the trajectory does not depend on any of this.
static int gs_fork(bpa_sampler * s)
{
  if (s->g_forked) return 1;
  HIPCHK(hipEventRecord(s->g_ev_fork, s->eng->stream));
  for (unsigned p = 1; p < s->g_np; ++p) HIPCHK(hipStreamWaitEvent(s->g_st[p], s->g_ev_fork, 0));
  s->g_forked = true;
  return 1;
}

static int gs_join(bpa_sampler * s)
{
  if (!s->g_forked) return 1;
  for (unsigned p = 1; p < s->g_np; ++p)
  {
    HIPCHK(hipEventRecord(s->g_ev_join[p], s->g_st[p]));
    HIPCHK(hipStreamWaitEvent(s->eng->stream, s->g_ev_join[p], 0));
  }
  s->g_forked = false;
  return 1;
}

static int gs_upload(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  const unsigned T = s->nloci;
  // (the locus table first: flush_state marks the packing stale when it had to send the table)
  if (!s->g_s20 && (!flush_state(e) || !engine_pack(e))) return 0;
  if (s->g_s20 && !flush_state(e)) return 0;        // (tip states, weights, parameter blocks, eigensystems on the device)
  for (unsigned i = 0; i < T; ++i) if (!assign_pops_host(s, s->g_trees[i])) return 0;
  for (int p = 0; p < smp::MAXPOP; ++p) s->has_theta[p] = p >= s->sp.S && p < s->sp.npop;
  std::vector<gsm::GLocus> loc(T);
  unsigned npat = 0;
  for (unsigned i = 0; i < T; ++i)
  {
    const bpa_locus * l = s->loci[i];
    const gsm::GTree & t = s->g_trees[i];
    int cnt[smp::MAXPOP] = {0};
    for (int k = 0; k < t.tips; ++k) if (++cnt[t.pop[k]] >= 2) s->has_theta[t.pop[k]] = true;
    if (!s->g_s20 && e->slot_of[l->id] < 0) return fail("bpa_sampler: a locus is not on the engine's packing");
    gsm::GLocus & g = loc[i];
    g.slot = s->g_s20 ? i : (uint32_t)e->slot_of[l->id]; g.pat_off = npat; g.R = l->rate_cats; g.pad = 0; g.par = l->dev.par;
    for (int p = 0; p < smp::MAXPOP; ++p) g.gl[p] = g.nin[p] = 0;
    for (int k = 0; k < t.tips; ++k)
    {
      g.nin[t.pop[k]]++;
      for (int q = t.pop[k]; q >= 0; q = s->sp.parent[q]) g.gl[q]++;
    }
    npat += l->sites;
  }
  s->g_npat = npat;
  s->g_units = 1 + std::max(s->maxtips - 1, 3u);
  s->g_maxmat = 2*s->maxtips - 2;
  s->g_pack_epoch = e->pack_epoch;
  const size_t nslots = s->g_s20 ? T : e->pack_slots;
  const size_t nrec = s->g_s20 ? 1 : nslots*s->g_units, nmat = std::max(nslots, (size_t)T)*s->g_maxmat;
  std::vector<uint32_t> bmo(s->g_s20 ? 1 : e->pack_blocks + 1);
  if (!s->g_s20) for (unsigned b = 0; b <= e->pack_blocks; ++b) bmo[b] = e->h_blk_slot_off[b]*s->g_maxmat;
  if (s->g_s20)
  {
    // the static part of the tiled kernels' plan: one task per locus, 64-pattern tiles, no scalers
    std::vector<uint32_t> tl(T), tp(T + 1), tt, tn;
    std::vector<int32_t> rs(T, BPA_SCALE_BUFFER_NONE);
    unsigned off = 0;
    for (unsigned i = 0; i < T; ++i)
    {
      tl[i] = s->loci[i]->id; tp[i] = off; off += s->loci[i]->sites;
      for (unsigned n0 = 0; n0 < s->loci[i]->sites; n0 += gs_tile20()) { tt.push_back(i); tn.push_back(n0); }
    }
    tp[T] = off;
    s->g_ntiles = (unsigned)tt.size(); s->g_maxops = s->maxtips - 1;
    if (!upload(s->g_tlocus, tl.data(), T) || !upload(s->g_tpat, tp.data(), T + 1) || !upload(s->g_ttask, tt.data(), tt.size()) ||
        !upload(s->g_tn0, tn.data(), tn.size()) || !upload(s->g_rscaler, rs.data(), T) || !s->g_ops20.reserve((size_t)T*s->g_maxops) ||
        !s->g_oprng.reserve((size_t)2*T) || !s->g_root20.reserve(T) || !s->g_mtask.reserve(nmat) || !s->g_mpm.reserve(nmat))
      return 0;
    HIPCHK(hipMemsetAsync(s->g_oprng.p, 0, (size_t)2*T*sizeof(uint32_t), e->stream));
    HIPCHK(hipMemsetAsync(s->g_root20.p, 0, (size_t)T*sizeof(uint32_t), e->stream));
    HIPCHK(hipMemsetAsync(s->g_mtask.p, 0xff, nmat*sizeof(uint32_t), e->stream));
    HIPCHK(hipMemsetAsync(s->g_mpm.p, 0, nmat*sizeof(uint32_t), e->stream));
  }
  if (!s->g_s20)
  {
    // the chain launch of the per-locus steps (gchain_kernel) evaluates with the engine's one-lane-per-pattern functions:
    // a task per locus and the step's records in the OpDev form, next to the packing's compact records
    std::vector<uint32_t> tl(T), tp(T + 1);
    std::vector<int32_t> rs(T, BPA_SCALE_BUFFER_NONE);
    unsigned off = 0;
    for (unsigned i = 0; i < T; ++i) { tl[i] = s->loci[i]->id; tp[i] = off; off += s->loci[i]->sites; }
    tp[T] = off;
    s->g_maxops = s->maxtips - 1;
    const size_t nm = (size_t)T*s->g_maxmat;
    if (!upload(s->g_tlocus, tl.data(), T) || !upload(s->g_tpat, tp.data(), T + 1) || !upload(s->g_rscaler, rs.data(), T) ||
        !s->g_ops20.reserve((size_t)T*s->g_maxops) || !s->g_oprng.reserve((size_t)2*T) || !s->g_root20.reserve(T) || !s->g_mtask.reserve(nm) || !s->g_mpm.reserve(nm))
      return 0;
    HIPCHK(hipMemsetAsync(s->g_oprng.p, 0, (size_t)2*T*sizeof(uint32_t), e->stream));
    HIPCHK(hipMemsetAsync(s->g_root20.p, 0, (size_t)T*sizeof(uint32_t), e->stream));
    HIPCHK(hipMemsetAsync(s->g_mtask.p, 0xff, nm*sizeof(uint32_t), e->stream));
    HIPCHK(hipMemsetAsync(s->g_mpm.p, 0, nm*sizeof(uint32_t), e->stream));
  }
  uint32_t zero2[2] = {0, 0};
  if (!upload(s->g_dev, s->g_trees.data(), T) || !s->g_undo.reserve(T) || !upload(s->g_loc, loc.data(), T) ||
      !s->g_lnl.reserve(T) || !s->g_lnlcur.reserve(T) || !s->g_hast.reserve(T) || !s->g_logpr.reserve(T) || !s->g_delta.reserve(T) || !s->g_active.reserve(T) ||
      !s->g_site.reserve(npat) || !s->g_recs.reserve(nrec) || !s->g_mat2.reserve(nmat) || !s->g_len.reserve(nmat) ||
      !upload(s->g_bmo, bmo.data(), bmo.size()) || !s->g_lograt.reserve(gsm::NN*gsm::NN) ||
      !upload(s->flag, zero2, 1) || !upload(s->counters, s->h_counters, 4) || !s->mix_sum.reserve(1) ||
      !upload(s->taus, s->h_taus.data(), s->h_taus.size()) || !s->pop_t2h.reserve((size_t)T*smp::MAXPOP) ||
      !s->pop_nc.reserve((size_t)T*smp::MAXPOP) || !s->theta_sums.reserve(smp::MAXPOP))
    return 0;
  s->g_sm.free();                                    // (re-made from the host copy by gs_subst_ready when a move is on)
  if (!gs_subst_ready(s)) return 0;
  // every slot starts as "not part of the step", every matrix entry as a hole
  HIPCHK(hipMemsetAsync(s->g_recs.p, 0xff, nrec*sizeof(uint4), e->stream));
  HIPCHK(hipMemsetAsync(s->g_mat2.p, 0xff, nmat*sizeof(MatRec2), e->stream));
  HIPCHK(hipMemsetAsync(s->g_len.p, 0, nmat*sizeof(double), e->stream));
  HIPCHK(hipMemsetAsync(s->g_lnl.p, 0, T*sizeof(double), e->stream));
  HIPCHK(hipMemsetAsync(s->g_hast.p, 0, T*sizeof(double), e->stream));
  HIPCHK(hipMemsetAsync(s->g_logpr.p, 0, T*sizeof(double), e->stream));
  HIPCHK(hipMemsetAsync(s->g_delta.p, 0, T*sizeof(double), e->stream));
  HIPCHK(hipMemsetAsync(s->g_active.p, 0, T, e->stream));
  if (s->allreduce)
  {
    // the THETA mask is a property of all ranks' loci (see sampler_upload)
    double m[smp::MAXPOP];
    for (int p = 0; p < smp::MAXPOP; ++p) m[p] = s->has_theta[p] ? 1.0 : 0.0;
    double * ar = s->sum_ext ? s->sum_ext : s->theta_sums.p;
    HIPCHK(hipMemcpyAsync(ar, m, sizeof m, hipMemcpyHostToDevice, e->stream));
    if (!s->allreduce(s->allreduce_ctx, ar, (unsigned)smp::MAXPOP, (void *)e->stream)) return fail("bpa_sampler: the all-reduce callback failed");
    HIPCHK(hipMemcpyAsync(m, ar, sizeof m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int p = 0; p < smp::MAXPOP; ++p) s->has_theta[p] = m[p] > 0.5;
  }
  hipLaunchKernelGGL(gsm::glograt_kernel, dim3(1), dim3(gsm::NN*gsm::NN), 0, e->stream, s->g_lograt.p);
  HIPCHK(hipGetLastError());
  s->epoch = 0; s->mix_pending = false; s->g_pend = 0;
  // part-batches (two by default) when there is enough of the packing to divide and the sampler's loci are in slot order
  s->g_split = false; s->g_forked = false; s->g_np = 1;
  static const bool no_split = BPA_EXP_SWITCH("BPA_GS_NOSPLIT") != nullptr;
  // Two part-batches by default; THREE for a 4-state set whose packing spans >= 2 400 workgroups (round 6, config 3: 10 000 loci 231.7 ->
  // 241.1 it/s, 7 500: 287.6 -> 298.9, 5 000: 357.6 -> 362.7; below that the extra launches cost more than the overlap buys — 2 500:
  // 491.7 -> 480.1, 1 250: 592.8 -> 562.8; four parts: 191 at 10 000 loci — and 20-state sets lose with three: config 4 139.7 -> 136.1)
  unsigned want_np = (unsigned)std::min(std::max(s->env_parts > 0 ? s->env_parts : 2, 2), (int)bpa_sampler::GPARTS);
  auto part_streams = [&](unsigned np) -> int
  {
    s->g_st[0] = e->stream;
    for (unsigned p = 1; p < np; ++p)
    {
      if (!s->g_st[p]) HIPCHK(hipStreamCreateWithFlags(&s->g_st[p], hipStreamNonBlocking));
      if (!s->g_ev_join[p]) HIPCHK(hipEventCreateWithFlags(&s->g_ev_join[p], hipEventDisableTiming));
    }
    if (!s->g_ev_fork) HIPCHK(hipEventCreateWithFlags(&s->g_ev_fork, hipEventDisableTiming));
    return 1;
  };
  if (!s->g_s20 && !s->g_alljc && !no_split && e->usedata && T >= 2)
  {
    bool mono = true;
    for (unsigned i = 0; i + 1 < T && mono; ++i) mono = loc[i].slot < loc[i + 1].slot;
    auto block_of = [&](uint32_t slot) { return (unsigned)(std::upper_bound(e->h_blk_slot_off.begin(), e->h_blk_slot_off.end(), slot) - e->h_blk_slot_off.begin()) - 1u; };
    const unsigned b_lo = block_of(loc[0].slot), b_hi = block_of(loc[T - 1].slot) + 1u, span = b_hi - b_lo;
    if (s->env_parts <= 0 && span >= 2400u) want_np = 3;
    if (mono && span >= 48u*want_np)
    {
      // part p starts with the workgroup of the packing that holds locus p T / np
      bool ok = true;
      s->g_pi[0] = 0; s->g_ps[0] = 0; s->g_pb[0] = 0;
      for (unsigned p = 1; p < want_np && ok; ++p)
      {
        const unsigned bs = block_of(loc[(size_t)T*p/want_np].slot), ss = e->h_blk_slot_off[bs];
        unsigned is = s->g_pi[p - 1];
        while (is < T && loc[is].slot < ss) ++is;
        ok = is > s->g_pi[p - 1] && is < T && bs > s->g_pb[p - 1];
        s->g_pi[p] = is; s->g_ps[p] = ss; s->g_pb[p] = bs;
      }
      s->g_pi[want_np] = T; s->g_ps[want_np] = e->pack_slots; s->g_pb[want_np] = e->pack_blocks;
      if (ok)
      {
        if (!part_streams(want_np)) return 0;
        s->g_split = true; s->g_np = want_np;
      }
    }
  }
  // 20-state sets: parts of the loci (the step's records are per locus, the tiles in locus order): the proposal, P-matrix
  // and sum launches of one part — latency, 70 us of a 385 us step on config 4 — run under the others' node updates
  if (s->g_s20 && !no_split && e->usedata && T >= 64u*want_np)
  {
    if (!part_streams(want_np)) return 0;
    s->g_split = true; s->g_np = want_np;
    unsigned nt = 0, p = 1;
    s->g_pi[0] = 0; s->g_pt[0] = 0; s->g_ps[0] = 0; s->g_pb[0] = 0;
    for (unsigned i = 0; i < T; ++i)
    {
      if (p < want_np && i == (unsigned)((size_t)T*p/want_np)) { s->g_pi[p] = i; s->g_pt[p] = nt; s->g_ps[p] = 0; s->g_pb[p] = 0; ++p; }
      nt += (s->loci[i]->sites + gs_tile20() - 1u)/gs_tile20();
    }
    s->g_pi[want_np] = T; s->g_pt[want_np] = nt;
  }
  s->uploaded = true; s->gp_mirror = false;
  return 1;
}

// one gstep_kernel launch: settle what is pending, then propose `mode`
// the program's THETA / TAU / MIX, decided on the host (BPP's proposal kernel + bpa_sampler_set_program_moves + a theta prior + a theta to move)
static bool gs_prog(const bpa_sampler * s) { return s->kernel_bpp && s->sp.program_moves && s->sp.theta_alpha > 0 && s->sp.npop > s->sp.S; }

// BPA_GS_FUSEA=1: the eigensystem refresh and the P-matrix phase inside the node-update launch (step_s4_klane kernels with
// FUSE_A).  Rounds 4's default for sets of up to 1 536 workgroups of the packing (config 3, 1 250 loci: 546 -> 567 it/s; slower
// above: 10 000 loci 189 -> 173); since round 5 the proposal's lane groups fill the step's P-matrices (gs_step: fuse_pm), which
// is ahead at every size (1 250 loci 507 -> 513 it/s, 2 500: 408 -> 423, 10 000: 194 -> 207), so this is a switch only
static bool gs_fuse_a(const bpa_sampler * s)
{
  return !s->g_alljc && !s->g_s20 && s->env_fusea == 1;         // (read at the sampler's creation: this runs per launch)
}
// Round 6: a step whose P-matrices the proposal kernel could NOT fill — a substitution-parameter step (every matrix of every locus
// from a moved parameter block, the eigensystems first) and the step after one (the rejected proposals rolled back) — was four
// launches: proposal, eigen_kernel, pmatrix_s4_dense_kernel, node updates.  On a small set (a strong-scaling rank's share: every
// launch ~15-25 us whatever it does, an iteration a chain of them) the FUSE_A form of the node-update kernel takes the middle two
// inside: two launches.  Sets of more than 1 536 workgroups of the packing keep the separate launches (the fused kernel's registers
// cost it a wave per SIMD: round 4's threshold).  BPA_GS_FUSEA=0: never.
static bool gs_fuse_a_step(const bpa_sampler * s)
{
  if (gs_fuse_a(s)) return true;
  return s->env_fusea != 0 && !s->g_alljc && !s->g_s20 && !s->g_pm_fused && s->eng->pack_blocks <= 1536u;
}

static int gs_step(bpa_sampler * s, unsigned mode, unsigned k = 0, double tau_u = 0, double mix_c = 1.0, double mix_lnc = 0, double tau_w = 0)
{
  bpa_engine * e = s->eng;
  if (s->g_pack_epoch != e->pack_epoch) return fail("bpa_sampler: the engine's loci changed since the sampler was set up");
  if (s->g_split && (mode <= 1 || mode >= 6)) { if (!gs_fork(s)) return 0; }
  else if (!gs_join(s)) return 0;
  gsm::GArgs a{};
  a.trees = s->g_dev.p; a.undo = s->g_undo.p; a.loc = s->g_loc.p; a.T = s->nloci; a.mode = mode; a.k = k;
  a.pend = s->g_pend; a.lnl_new = s->g_lnl.p; a.hast = s->g_hast.p; a.logpr_new = s->g_logpr.p; a.delta = s->g_delta.p;
  a.active = s->g_active.p; a.flag = s->flag.p; a.epoch = s->epoch; a.lnl_cur = s->g_lnlcur.p;
  a.recs2 = s->g_recs.p; a.units = s->g_units; a.mat2 = s->g_mat2.p; a.mat_length = s->g_len.p; a.maxmat = s->g_maxmat;
  a.taus = s->taus.p; a.lograt = s->g_lograt.p; a.tau_q = k; a.tau_u = tau_u; a.mix_c = mix_c; a.mix_lnc = mix_lnc;
  a.pop_nc = s->pop_nc.p; a.pop_t2h = s->pop_t2h.p;
  a.refresh_logpr = s->logpr_stale ? 1u : 0u; s->logpr_stale = false;
  a.sp = s->sp;
  a.pend_mode = s->g_pend_mode; a.pend_k = s->g_pend_k; a.sm = s->g_sm.p; a.sm_old = s->g_sm_old.p;
  a.fmt20 = s->g_s20 ? 1u : 0u; a.maxops20 = s->g_maxops;
  a.ops20 = s->g_ops20.p; a.op_rng20 = s->g_oprng.p; a.root20 = s->g_root20.p; a.mat_task20 = s->g_mtask.p; a.mat_pm20 = s->g_mpm.p;
  a.ft_freqs = s->g_ft[0]; a.ft_qrates = s->g_ft[1]; a.ft_alpha = s->g_ft[2]; a.alpha_a = s->g_alpha_a; a.alpha_b = s->g_alpha_b;
  a.bpp = s->kernel_bpp ? 1u : 0u; a.prog = gs_prog(s) ? 1u : 0u; a.t2h3 = s->g_t2h3.p; a.tau_w = tau_w;
  a.slot_tab = e->d_slot_tab.p;
  a.dstep = (s->gp_dev && (mode == 2 || mode == 3)) ? s->g_dst.p : nullptr;
  // a frequency / exchangeability step — proposed now, or rolled back now for the loci that rejected it — leaves
  // parameter blocks whose eigensystems are stale: refreshed before the next evaluation (gs_eval)
  // ... unless the launch is the one-lane kernel (settle, start-up, the parameter moves) on 4-state loci whose eigensystems are
  // level now: its lanes then refresh what they write themselves (GArgs::fuse_eigen, round 6) and nothing is left stale
  const bool touches_eigen = mode == 6 || mode == 7 || (s->g_pend == 4 && (s->g_pend_mode == 6 || s->g_pend_mode == 7));
  a.fuse_eigen = (touches_eigen && mode >= 4 && !s->g_s20 && !s->g_alljc && e->usedata && !s->g_eigen_dirty && !s->env_noeigfuse) ? 1u : 0u;
  if (touches_eigen && !a.fuse_eigen) s->g_eigen_dirty = true;
  // diagnostics (BPA_GS_DIFF=1): a GAGE / GSPR step by both kernels from the same state, whatever differs is reported;
  // the run goes on with the one-lane kernel's result
#define GS2_CASES(NT_, BPP_, GRID_, ST_) switch (a.mode) { \
      case 0: hipLaunchKernelGGL((gsm2::gstep2_kernel<0, NT_, BPP_>), GRID_, dim3(64), 0, ST_, a); break; case 1: hipLaunchKernelGGL((gsm2::gstep2_kernel<1, NT_, BPP_>), GRID_, dim3(64), 0, ST_, a); break; \
      case 2: hipLaunchKernelGGL((gsm2::gstep2_kernel<2, NT_, BPP_>), GRID_, dim3(64), 0, ST_, a); break; default: hipLaunchKernelGGL((gsm2::gstep2_kernel<3, NT_, BPP_>), GRID_, dim3(64), 0, ST_, a); break; }
#define GS2_LAUNCH(GRID_, ST_) do { \
    if (s->kernel_bpp) { if (s->maxtips <= 8) GS2_CASES(8, true, GRID_, ST_) else GS2_CASES(16, true, GRID_, ST_) } \
    else               { if (s->maxtips <= 8) GS2_CASES(8, false, GRID_, ST_) else GS2_CASES(16, false, GRID_, ST_) } } while (0)
#define GS1_LAUNCH(GRID_, ST_) do { \
    if (s->maxtips <= 8) switch (a.mode) { case 0: hipLaunchKernelGGL((gsm::gstep_kernel<0, 8>), GRID_, dim3(gsm::GBS), 0, ST_, a); break; case 1: hipLaunchKernelGGL((gsm::gstep_kernel<1, 8>), GRID_, dim3(gsm::GBS), 0, ST_, a); break; \
                                           case 2: hipLaunchKernelGGL((gsm::gstep_kernel<2, 8>), GRID_, dim3(gsm::GBS), 0, ST_, a); break; default: hipLaunchKernelGGL((gsm::gstep_kernel<3, 8>), GRID_, dim3(gsm::GBS), 0, ST_, a); break; } \
    else switch (a.mode) { case 0: hipLaunchKernelGGL((gsm::gstep_kernel<0, 16>), GRID_, dim3(gsm::GBS), 0, ST_, a); break; case 1: hipLaunchKernelGGL((gsm::gstep_kernel<1, 16>), GRID_, dim3(gsm::GBS), 0, ST_, a); break; \
                           case 2: hipLaunchKernelGGL((gsm::gstep_kernel<2, 16>), GRID_, dim3(gsm::GBS), 0, ST_, a); break; default: hipLaunchKernelGGL((gsm::gstep_kernel<3, 16>), GRID_, dim3(gsm::GBS), 0, ST_, a); break; } } while (0)
  static const bool gs_diff = getenv("BPA_GS_DIFF") != nullptr;
  static const bool gs_v1 = BPA_EXP_SWITCH("BPA_GS_V1") != nullptr;
  // the step's P-matrices by the proposal's lane groups (gstep2_body) instead of a launch of their own: 4-state loci on the
  // packing whose step launch does not make them itself (gs_fuse_a), no substitution-parameter step pending or rolled back
  // since the eigensystems were last refreshed (BPA_GS_FUSEPM=0: the dense launch)
  s->g_pm_fused = false;
  if (mode <= 3 && !gs_v1 && !gs_diff && !s->g_s20 && !s->g_alljc && e->usedata && s->g_pend != 4 && !s->g_eigen_dirty && !gs_fuse_a(s))
  {
    s->g_pm_fused = s->env_fusepm;
  }
  a.fuse_pm = s->g_pm_fused ? 1u : 0u;
  if (s->kernel_bpp && mode <= 3 && (gs_diff || gs_v1)) return fail("bpa_sampler: BPP's proposal kernel has no one-lane form (BPA_GS_DIFF / BPA_GS_V1 are the uniform kernel's diagnostics)");
  if (gs_diff && mode <= 3 && !s->g_forked)
  {
    const unsigned n = s->nloci;
    std::vector<gsm::GTree> t0(n), u0(n), t2(n), u2(n), t1(n), u1(n);
    std::vector<double> lp2(n), h2(n), lp1(n), h1(n), dd1(n), dd2(n), lc1(n), lc2(n); std::vector<uint8_t> a2(n), a1(n);
    const size_t nrec = (size_t)e->pack_slots*s->g_units, nmat = (size_t)e->pack_slots*s->g_maxmat;
    std::vector<uint4> r2(nrec), r1(nrec); std::vector<MatRec2> m2(nmat), m1(nmat); std::vector<double> l2(nmat), l1(nmat);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(t0.data(), s->g_dev.p, n*sizeof(gsm::GTree), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(u0.data(), s->g_undo.p, n*sizeof(gsm::GTree), hipMemcpyDeviceToHost));
    std::vector<double> lp0(n), h0(n); std::vector<uint8_t> a0(n);             // the step reads the last step's, then writes its own
    HIPCHK(hipMemcpy(lp0.data(), s->g_logpr.p, n*sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(h0.data(), s->g_hast.p, n*sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(a0.data(), s->g_active.p, n, hipMemcpyDeviceToHost));
    a.i0 = 0; a.iend = n;
    for (int pass = 0; pass < 2; ++pass)
    {
      if (pass == 0)
      {
        const unsigned lpw = s->maxtips <= 8 ? 4u : 2u;
        GS2_LAUNCH(dim3((n + lpw - 1)/lpw), e->stream);
      }
      else
      {
        HIPCHK(hipMemcpy(s->g_dev.p, t0.data(), n*sizeof(gsm::GTree), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(s->g_undo.p, u0.data(), n*sizeof(gsm::GTree), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(s->g_logpr.p, lp0.data(), n*sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(s->g_hast.p, h0.data(), n*sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(s->g_active.p, a0.data(), n, hipMemcpyHostToDevice));
        GS1_LAUNCH(dim3((n + gsm::GBS - 1)/gsm::GBS), e->stream);
      }
      HIPCHK(hipStreamSynchronize(e->stream));
      HIPCHK(hipMemcpy((pass ? t1 : t2).data(), s->g_dev.p, n*sizeof(gsm::GTree), hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy((pass ? u1 : u2).data(), s->g_undo.p, n*sizeof(gsm::GTree), hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy((pass ? lp1 : lp2).data(), s->g_logpr.p, n*sizeof(double), hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy((pass ? h1 : h2).data(), s->g_hast.p, n*sizeof(double), hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy((pass ? a1 : a2).data(), s->g_active.p, n, hipMemcpyDeviceToHost));
      if (mode >= 2)
      {
        HIPCHK(hipMemcpy((pass ? dd1 : dd2).data(), s->g_delta.p, n*sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy((pass ? lc1 : lc2).data(), s->g_lnlcur.p, n*sizeof(double), hipMemcpyDeviceToHost));
      }
      if (!s->g_s20)
      {
        HIPCHK(hipMemcpy((pass ? r1 : r2).data(), s->g_recs.p, nrec*sizeof(uint4), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy((pass ? m1 : m2).data(), s->g_mat2.p, nmat*sizeof(MatRec2), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy((pass ? l1 : l2).data(), s->g_len.p, nmat*sizeof(double), hipMemcpyDeviceToHost));
      }
    }
    int shown = 0;
    auto dump = [&](const char * name, const gsm::GTree & g)
    {
      const int nn = 2*g.tips - 1;
      fprintf(stderr, "   %s root %d tips %d lnl %.17g logpr %.17g rng %llx prop %u acc %u\n", name, g.root, g.tips, g.lnl, g.logpr, (unsigned long long)g.rng, g.proposals, g.accepted);
      for (int k = 0; k < nn && k < gsm::NN; ++k)
        fprintf(stderr, "     %2d l %3d r %3d p %3d clv %3d pm %3d pop %3d t %.17g\n", k, g.left[k], g.right[k], g.parent[k], g.clv[k], g.pmat[k], g.pop[k], g.time[k]);
    };
    auto tree_differs = [&](const gsm::GTree & x, const gsm::GTree & y, bool head)
    {
      const int m = 2*y.tips - 1;
      bool d = x.root != y.root;
      if (head) d = d || x.tips != y.tips || x.lnl != y.lnl || x.logpr != y.logpr || x.rng != y.rng || x.proposals != y.proposals || x.accepted != y.accepted
                      || x.work_nupd != y.work_nupd || x.work_nbr != y.work_nbr || x.work_neval != y.work_neval;
      for (int k = 0; k < m && k < gsm::NN; ++k)
        d = d || x.left[k] != y.left[k] || x.right[k] != y.right[k] || x.parent[k] != y.parent[k] || x.clv[k] != y.clv[k] || x.pmat[k] != y.pmat[k]
              || x.pop[k] != y.pop[k] || x.time[k] != y.time[k];
      return d;
    };
    for (unsigned i = 0; i < n; ++i)
    {
      gsm::GTree uu2 = u2[i], uu1 = u1[i]; uu2.tips = uu1.tips = t1[i].tips;
      const bool dt = tree_differs(t2[i], t1[i], true), du = tree_differs(uu2, uu1, false);
      const bool da = a2[i] != a1[i], dl = (a1[i] && (lp2[i] != lp1[i] || h2[i] != h1[i])) || (mode >= 2 && (lp2[i] != lp1[i] || dd2[i] != dd1[i] || lc2[i] != lc1[i]));
      if ((dt || du || da || dl) && shown++ < 3)
      {
        fprintf(stderr, "[gs diff] mode %u k %u pend %u locus %u: tree %d undo %d active %d/%d logpr %.17g/%.17g hast %.17g/%.17g delta %.17g/%.17g\n", mode, k, a.pend, i,
                (int)dt, (int)du, (int)a2[i], (int)a1[i], lp2[i], lp1[i], h2[i], h1[i], dd2[i], dd1[i]);
        dump("before ", t0[i]); dump("group  ", t2[i]); dump("one    ", t1[i]);
        if (du) { dump("undo g ", uu2); dump("undo 1 ", uu1); }
      }
    }
    if (!s->g_s20)
    {
      int nr = 0;
      for (size_t q = 0; q < nrec; ++q)
        if (memcmp(&r2[q], &r1[q], sizeof(uint4)) && nr++ < 4)
          fprintf(stderr, "[gs diff] mode %u k %u rec slot %zu unit %zu: %08x %08x %08x %08x / %08x %08x %08x %08x\n", mode, k, q/s->g_units, q % s->g_units,
                  r2[q].x, r2[q].y, r2[q].z, r2[q].w, r1[q].x, r1[q].y, r1[q].z, r1[q].w);
      for (size_t q = 0; q < nmat; ++q)
        if ((m2[q].slot != m1[q].slot || (m1[q].slot != 0xffffffffu && (m2[q].pmatrix != m1[q].pmatrix || l2[q] != l1[q]))) && nr++ < 8)
          fprintf(stderr, "[gs diff] mode %u k %u mat slot %zu entry %zu: %x %u %.17g / %x %u %.17g\n", mode, k, q/s->g_maxmat, q % s->g_maxmat,
                  m2[q].slot, m2[q].pmatrix, l2[q], m1[q].slot, m1[q].pmatrix, l1[q]);
      if (nr) shown += nr;
    }
    fprintf(stderr, "[gs diff] mode %u k %u pend %u: %d differences\n", mode, k, a.pend, shown);
    s->launches++;
    s->g_pend = mode <= 1 ? 1u : 2u; s->g_pend_mode = mode; s->g_pend_k = k;
    return 1;
  }
  for (unsigned h = 0; h < (s->g_forked ? s->g_np : 1u); ++h)
  {
    a.i0 = s->g_forked ? s->g_pi[h] : 0u; a.iend = s->g_forked ? s->g_pi[h + 1] : s->nloci;
    const dim3 grid((a.iend - a.i0 + gsm::GBS - 1)/gsm::GBS), block(gsm::GBS);
    hipStream_t st = h ? s->g_st[h] : e->stream;
    // GAGE / GSPR / TAU / MIX: a group of lanes per locus (gsampler2.hpp; BPA_GS_V1=1: the one-lane-per-locus kernel)
    if (a.mode <= 3 && !gs_v1)
    {
      const unsigned lpw = s->maxtips <= 8 ? 4u : 2u;
      GS2_LAUNCH(dim3((a.iend - a.i0 + lpw - 1)/lpw), st);
      s->launches++;
      continue;
    }
    switch (a.mode)
    {
#define GS_CASE(M_) case M_: if (s->maxtips <= 8) hipLaunchKernelGGL((gsm::gstep_kernel<M_, 8>), grid, block, 0, st, a); \
                            else                 hipLaunchKernelGGL((gsm::gstep_kernel<M_, 16>), grid, block, 0, st, a); break
      GS_CASE(0); GS_CASE(1); GS_CASE(2); GS_CASE(3); GS_CASE(4); GS_CASE(5); GS_CASE(6); GS_CASE(7); GS_CASE(8);
#undef GS_CASE
      default: return fail("bpa_sampler: unknown step mode");
    }
    s->launches++;
  }
  HIPCHK(hipGetLastError());
  static const bool dbg_sync = BPA_EXP_SWITCH("BPA_GS_SYNC") != nullptr;        // diagnostics: wait for every launch and say which it was
  if (dbg_sync)
  {
    HIPCHK(hipStreamSynchronize(e->stream));
    if (s->g_forked) for (unsigned p = 1; p < s->g_np; ++p) HIPCHK(hipStreamSynchronize(s->g_st[p]));
    fprintf(stderr, "[gs] step mode %u k %u pend %u done\n", mode, k, a.pend);
  }
#ifdef GS2_PROF
  if (mode <= 1)
  {
    HIPCHK(hipStreamSynchronize(e->stream));
    std::vector<double> d(s->nloci), c(s->nloci);
    HIPCHK(hipMemcpy(d.data(), s->g_delta.p, s->nloci*sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c.data(), s->g_lnlcur.p, s->nloci*sizeof(double), hipMemcpyDeviceToHost));
    double sum[8] = {0}; int cnt[8] = {0}; double t0 = 1e300, t1 = 0;
    for (unsigned i = 0; i < s->nloci; ++i) { sum[i & 7] += d[i]; cnt[i & 7]++; if (i & 1) t1 = std::max(t1, c[i]); else t0 = std::min(t0, c[i]); }
    fprintf(stderr, "[gs2 prof] mode %u k %u: phases (us)", mode, k);
    for (int q = 0; q < 8; ++q) fprintf(stderr, " %.2f", cnt[q] ? sum[q]/cnt[q]*0.01 : 0.0);
    fprintf(stderr, "  first start -> last end %.2f us\n", (t1 - t0)*0.01);
  }
#endif
  s->g_pend = mode <= 1 ? 1u : (mode == 2 || mode == 3) ? 2u : mode == 5 ? 3u : mode >= 6 ? 4u : 0u;
  s->g_pend_mode = mode; s->g_pend_k = k;
  return 1;
}

// flags bit 11 of the node-update kernels (step_s4_klane_v3_kernel, partials_lnl_wave20_kernel): the root's CLV of a step is not
// stored — no step of a sampler reads a root's buffer (the root term is taken from registers in the same launch; a node that
// stops being the root is recomputed by the step that moves it).  What else may read the loci's buffers (the single-locus API,
// a plan, bpa_batch_evaluate) comes after a download, which recomputes every buffer in place first (gs_download).
// BPA_GS_ROOTSTORE=1: every parent stored (A/B).
static uint32_t gs_skip_root_flag(const bpa_sampler * s) { return s->env_rootstore ? 0u : 2048u; }
static uint32_t gs_root_flag(const bpa_sampler * s) { return s->g_level_eval ? 0u : gs_skip_root_flag(s); }

// the step's likelihood: the engine's kernels over the records the step kernel wrote
static int gs_eval(bpa_sampler * s, int kind /* 0 per-locus step, 1 all-loci step */)
{
  bpa_engine * e = s->eng;
  if (!e->usedata) return 1;                       // lnL = 0 for every locus (the buffer was zeroed): the MSC prior
  if (!s->g_alljc && !s->g_level_eval && gs_skip_root_flag(s)) s->g_root_stale = true;
  if (s->g_s20)
  {
    // amino-acid loci: fresh P-matrices (pmatrix_wg2_kernel, one workgroup per entry, holes return at once), the tiled
    // node-update + site-term kernel over the loci that have updates, the per-locus sums in pattern order
    PlanDev d{};
    d.loci = e->d_loci.p; d.bfbeta = e->bfbeta; d.task_locus = s->g_tlocus.p; d.task_pat_off = s->g_tpat.p;
    d.tile_task = s->g_ttask.p; d.tile_n0 = s->g_tn0.p; d.op_off = s->g_oprng.p; d.ops = s->g_ops20.p; d.root_clv = s->g_root20.p;
    d.root_scaler = s->g_rscaler.p; d.site_term = s->g_site.p; d.lnl = s->g_lnl.p; d.mat_task = s->g_mtask.p; d.mat_pmatrix = s->g_mpm.p;
    d.mat_length = s->g_len.p; d.nmat = s->nloci*s->g_maxmat; d.ntasks = s->nloci; d.npatterns = s->g_npat; d.pad = s->g_rmax;
    hipEvent_t k0 = nullptr, k1 = nullptr;
    if (s->timing_stride && (s->timing_phase++ % s->timing_stride) == 0)
    {
      if (s->timed.size() >= 4096 && !sampler_timing_drain(s)) return 0;
      bpa_sampler::Timed t{nullptr, nullptr, kind};
      HIPCHK(hipEventCreate(&t.e0)); HIPCHK(hipEventCreate(&t.e1));
      s->timed.push_back(t);
      k0 = t.e0; k1 = t.e1;
    }
    const size_t lds20 = ((size_t)4*s->g_rmax*400 + (size_t)s->g_rmax*64)*sizeof(double);
    // BPA_S20_SUM_FUSED=1: the per-locus sum inside the node-update kernel (its last tile of a locus adds the terms up; one launch
    // fewer per step and half).  Measured on config 4 and NOT the default: 119 it/s against 132 with lnl_reduce_wave_kernel as
    // a launch of its own — the arriving wave must drain its CLV stores before its arrival atomic, holding the workgroup's
    // registers and LDS meanwhile, and the last tile adds alone; with an agent-scope release fence instead of write-through
    // terms: 72 it/s (the fence writes back the XCD's whole L2, i.e. the CLV planes just stored)
    static const bool sum_fused = BPA_EXP_SWITCH("BPA_S20_SUM_FUSED") != nullptr;
    const bool fuse_sum = sum_fused && !gs_pipe20() && gs_tile20() == 64u;
    if (fuse_sum && !s->g_arrive.p)
    {
      if (!s->g_arrive.reserve(s->nloci)) return fail("out of device memory (arrival counters)");
      HIPCHK(hipMemsetAsync(s->g_arrive.p, 0, (size_t)s->nloci*sizeof(uint32_t), e->stream));
    }
    d.tile_arrive = s->g_arrive.p;
    const uint32_t fsum = fuse_sum ? 512u : 0u;
    // the step's P-matrices: one workgroup per locus (its entries are adjacent in the step image), BPA_S20_PMGROUP=0: per entry
    const bool pm_group = s->env_pmgroup;
    if (s->g_forked)
    {
      for (unsigned h = 0; h < s->g_np; ++h)
      {
        const unsigned i0 = s->g_pi[h], i1 = s->g_pi[h + 1], t0 = s->g_pt[h], t1 = s->g_pt[h + 1];
        hipStream_t st = h ? s->g_st[h] : e->stream;
        d.ent0 = i0*s->g_maxmat;
        d.flags = 1u | 256u;
        if (pm_group) hipLaunchKernelGGL(pmatrix_wg2_group_kernel<20>, dim3(i1 - i0), dim3(256), 0, st, d, s->g_maxmat);
        else hipLaunchKernelGGL(pmatrix_wg2_kernel<20>, dim3((i1 - i0)*s->g_maxmat), dim3(256), 0, st, d);
        d.blk0 = t0;
        d.flags = 4u | 64u | 256u | fsum | gs_root_flag(s);
#ifdef BPA_EXPERIMENTAL
        if (gs_pipe20()) hipExtLaunchKernelGGL((partials_lnl_pipe20_kernel<20, true, 2>), dim3(t1 - t0), dim3(64*s->g_rmax), lds20, st, h ? nullptr : k0, h ? nullptr : k1, 0, d);
        else if (gs_tile20() == 128u) hipExtLaunchKernelGGL((partials_lnl_wave20_kernel<20, true, 1, 2>), dim3(t1 - t0), dim3(64*s->g_rmax), lds20 + (size_t)s->g_rmax*64*sizeof(double), st, h ? nullptr : k0, h ? nullptr : k1, 0, d);
        else if (gs_waverl()) hipExtLaunchKernelGGL((partials_lnl_wave20_kernel<20, true, 2, 1, true>), dim3(t1 - t0), dim3(64*s->g_rmax), lds20, st, h ? nullptr : k0, h ? nullptr : k1, 0, d);
        else
#endif
        hipExtLaunchKernelGGL((partials_lnl_wave20_kernel<20, true, 2>), dim3(t1 - t0), dim3(64*s->g_rmax), lds20, st, h ? nullptr : k0, h ? nullptr : k1, 0, d);
        d.blk0 = i0;
        if (!fuse_sum) hipLaunchKernelGGL(lnl_reduce_wave_kernel, dim3(i1 - i0), dim3(64), 0, st, d);
      }
      HIPCHK(hipGetLastError());
      s->launches += (fuse_sum ? 2 : 3)*s->g_np; s->g_evals += s->g_np;
      return 1;
    }
    d.flags = 1u;
    if (pm_group) hipLaunchKernelGGL(pmatrix_wg2_group_kernel<20>, dim3(s->nloci), dim3(256), 0, e->stream, d, s->g_maxmat);
    else hipLaunchKernelGGL(pmatrix_wg2_kernel<20>, dim3(d.nmat), dim3(256), 0, e->stream, d);
    d.flags = 4u | 64u | fsum | gs_root_flag(s);
#ifdef BPA_EXPERIMENTAL
    if (gs_pipe20()) hipExtLaunchKernelGGL((partials_lnl_pipe20_kernel<20, true, 2>), dim3(s->g_ntiles), dim3(64*s->g_rmax), lds20, e->stream, k0, k1, 0, d);
    else if (gs_tile20() == 128u) hipExtLaunchKernelGGL((partials_lnl_wave20_kernel<20, true, 1, 2>), dim3(s->g_ntiles), dim3(64*s->g_rmax), lds20 + (size_t)s->g_rmax*64*sizeof(double), e->stream, k0, k1, 0, d);
    else if (gs_waverl()) hipExtLaunchKernelGGL((partials_lnl_wave20_kernel<20, true, 2, 1, true>), dim3(s->g_ntiles), dim3(64*s->g_rmax), lds20, e->stream, k0, k1, 0, d);
    else
#endif
    hipExtLaunchKernelGGL((partials_lnl_wave20_kernel<20, true, 2>), dim3(s->g_ntiles), dim3(64*s->g_rmax), lds20, e->stream, k0, k1, 0, d);
    if (!fuse_sum) hipLaunchKernelGGL(lnl_reduce_wave_kernel, dim3(s->nloci), dim3(64), 0, e->stream, d);
    HIPCHK(hipGetLastError());
    s->launches += fuse_sum ? 2 : 3; s->g_evals++;
    return 1;
  }
  const bool fuse_a = gs_fuse_a_step(s);
  const bool pm_done = s->g_pm_fused && !fuse_a;
  s->g_pm_fused = false;
  const bool fuse_eigen = fuse_a && s->g_eigen_dirty;
  if (fuse_eigen) s->g_eigen_dirty = false;
  else if (!gs_refresh_eigen(s)) return 0;
  PlanDev d{};
  d.loci = e->d_loci.p; d.bfbeta = e->bfbeta;
  d.site_term = s->g_site.p; d.lnl = s->g_lnl.p; d.ntasks = s->nloci; d.npatterns = s->g_npat; d.nmat = e->pack_slots*s->g_maxmat;
  d.recs2 = s->g_recs.p; d.mat2 = s->g_mat2.p; d.mat_length = s->g_len.p; d.blk_mat_off = s->g_bmo.p; d.rec2_units = s->g_units;
  d.lane_tab = e->d_lane_tab.p; d.slot_tab = e->d_slot_tab.p; d.blk_slot_off = e->d_blk_slot_off.p; d.nblocks2 = e->pack_blocks;
  hipEvent_t k0 = nullptr, k1 = nullptr;
  if (s->timing_stride && (s->timing_phase++ % s->timing_stride) == 0)
  {
    if (s->timed.size() >= 4096 && !sampler_timing_drain(s)) return 0;
    bpa_sampler::Timed t{nullptr, nullptr, kind};
    HIPCHK(hipEventCreate(&t.e0)); HIPCHK(hipEventCreate(&t.e1));
    s->timed.push_back(t);
    k0 = t.e0; k1 = t.e1;
  }
  const dim3 grid(e->pack_blocks), block(PACK_BS);
  if (s->g_alljc)
  {
    d.flags = 1u | 2u | 4u;
    hipExtLaunchKernelGGL((step_jc69_v2_kernel<PACK_BS>), grid, block, 0, e->stream, k0, k1, 0, d);
    s->launches += 1;
  }
  else
  {
    d.pad = s->g_rmax;
    if (s->g_forked)
    {
      for (unsigned h = 0; h < s->g_np; ++h)
      {
        const unsigned b0 = s->g_pb[h], b1 = s->g_pb[h + 1];
        const unsigned e0 = s->g_ps[h]*s->g_maxmat, e1 = s->g_ps[h + 1]*s->g_maxmat;
        hipStream_t st = h ? s->g_st[h] : e->stream;
        d.blk0 = b0; d.ent0 = e0;
        if (fuse_a)
        {
          d.flags = 1u | 2u | 4u | (fuse_eigen ? 32u : 0u) | gs_root_flag(s);
          launch_klane<true>(dim3(b1 - b0), st, h ? nullptr : k0, h ? nullptr : k1, d);
          s->launches += 1;
          continue;
        }
        d.flags = 1u;
        if (!pm_done) hipLaunchKernelGGL(pmatrix_s4_dense_kernel, dim3(((e1 - e0)*d.pad + 255u)/256u), dim3(256), 0, st, d, e1);
        d.flags = 2u | 4u | gs_root_flag(s);
        launch_klane<false>(dim3(b1 - b0), st, h ? nullptr : k0, h ? nullptr : k1, d);
        s->launches += pm_done ? 1 : 2;
      }
      HIPCHK(hipGetLastError());
      s->g_evals += s->g_np;
      return 1;
    }
    if (fuse_a)
    {
      d.flags = 1u | 2u | 4u | (fuse_eigen ? 32u : 0u) | gs_root_flag(s);
      launch_klane<true>(grid, e->stream, k0, k1, d);
      s->launches += 1;
    }
    else
    {
      d.flags = 1u;
      if (!pm_done) hipLaunchKernelGGL(pmatrix_s4_dense_kernel, dim3((d.nmat*d.pad + 255u)/256u), dim3(256), 0, e->stream, d, d.nmat);
      d.flags = 2u | 4u | gs_root_flag(s);
      launch_klane<false>(grid, e->stream, k0, k1, d);
      s->launches += pm_done ? 1 : 2;
    }
  }
  HIPCHK(hipGetLastError());
  s->g_evals++;
  static const bool dbg_sync_v = BPA_EXP_SWITCH("BPA_GS_SYNC") != nullptr;
  if (dbg_sync_v) { HIPCHK(hipStreamSynchronize(e->stream)); fprintf(stderr, "[gs] eval done\n"); }
  return 1;
}

// GAGE + GSPR of every locus in one launch (gsampler2.hpp: gchain_kernel).  BPA_GS_CHAIN=0 / 1: never / whenever possible.
// Default: small JC69 sets only — up to 1 024 loci of at most 64 patterns on average (config 5: 913 -> 1 096 it/s).  With
// several rate categories or hundreds of patterns a wave per locus walking one pattern per lane loses to the packing's
// kernels even when those are launch-bound (config 3's share of 1 250 loci: 516 it/s by launches, 337 chained).
static bool gs_chain_wanted(const bpa_sampler * s)
{
  if (s->g_s20 || !s->eng->usedata || s->maxtips < 2) return false;
  if (s->env_chain >= 0) return s->env_chain != 0;
  return s->g_alljc && s->nloci <= 1024u && s->g_npat <= 64u*s->nloci;
}
static int gs_chain(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  if (s->g_pack_epoch != e->pack_epoch) return fail("bpa_sampler: the engine's loci changed since the sampler was set up");
  // a pending parameter move is settled by a launch of its own: what it rolls back needs the eigensystems refreshed
  if (s->g_pend == 4) { if (!gs_step(s, 4) || !gs_refresh_eigen(s)) return 0; }
  if (!gs_join(s)) return 0;
  if (!gs_refresh_eigen(s)) return 0;
  gsm::GArgs a{};
  a.trees = s->g_dev.p; a.undo = s->g_undo.p; a.loc = s->g_loc.p; a.T = s->nloci;
  a.lnl_new = s->g_lnl.p; a.hast = s->g_hast.p; a.logpr_new = s->g_logpr.p; a.delta = s->g_delta.p;
  a.active = s->g_active.p; a.flag = s->flag.p; a.epoch = s->epoch; a.lnl_cur = s->g_lnlcur.p;
  a.recs2 = s->g_recs.p; a.units = s->g_units; a.mat2 = s->g_mat2.p; a.mat_length = s->g_len.p; a.maxmat = s->g_maxmat;
  a.taus = s->taus.p; a.lograt = s->g_lograt.p;
  a.pop_nc = s->pop_nc.p; a.pop_t2h = s->pop_t2h.p;
  a.sp = s->sp;
  a.pend_mode = s->g_pend_mode; a.pend_k = s->g_pend_k; a.sm = s->g_sm.p; a.sm_old = s->g_sm_old.p;
  a.bpp = s->kernel_bpp ? 1u : 0u; a.prog = gs_prog(s) ? 1u : 0u; a.t2h3 = s->g_t2h3.p;
  a.fmt20 = 1u; a.maxops20 = s->g_maxops;
  a.ops20 = s->g_ops20.p; a.op_rng20 = s->g_oprng.p; a.root20 = s->g_root20.p; a.mat_task20 = s->g_mtask.p; a.mat_pm20 = s->g_mpm.p;
  a.i0 = 0; a.iend = s->nloci;
  PlanDev d{};
  d.loci = e->d_loci.p; d.bfbeta = e->bfbeta; d.task_locus = s->g_tlocus.p; d.task_pat_off = s->g_tpat.p;
  d.op_off = s->g_oprng.p; d.ops = s->g_ops20.p; d.root_clv = s->g_root20.p; d.root_scaler = s->g_rscaler.p;
  d.site_term = s->g_site.p; d.lnl = s->g_lnl.p; d.mat_task = s->g_mtask.p; d.mat_pmatrix = s->g_mpm.p; d.mat_length = s->g_len.p;
  d.nmat = s->nloci*s->g_maxmat; d.ntasks = s->nloci; d.npatterns = s->g_npat; d.pad = s->g_rmax;
  d.flags = 2u | 4u | 64u;
  gsm2::GChain ch{s->maxtips - 1, 2*s->maxtips - 2, s->g_pend, s->logpr_stale ? 1u : 0u};
  s->logpr_stale = false;
  if (s->kernel_bpp)
  {
    if (s->maxtips <= 8) hipLaunchKernelGGL((gsm2::gchain_kernel<8, true>), dim3(s->nloci), dim3(gsm2::GCHAIN_THREADS), 0, e->stream, a, d, ch);
    else                 hipLaunchKernelGGL((gsm2::gchain_kernel<16, true>), dim3(s->nloci), dim3(gsm2::GCHAIN_THREADS), 0, e->stream, a, d, ch);
  }
  else if (s->maxtips <= 8) hipLaunchKernelGGL((gsm2::gchain_kernel<8>), dim3(s->nloci), dim3(gsm2::GCHAIN_THREADS), 0, e->stream, a, d, ch);
  else                 hipLaunchKernelGGL((gsm2::gchain_kernel<16>), dim3(s->nloci), dim3(gsm2::GCHAIN_THREADS), 0, e->stream, a, d, ch);
  HIPCHK(hipGetLastError());
  s->launches++; s->g_evals += ch.ngage + ch.ngspr;
  s->g_pend = 1u; s->g_pend_mode = 1u; s->g_pend_k = ch.ngspr - 1;
  return 1;
}

// the ONE decision of an all-loci step
static int gs_decide(bpa_sampler * s, double uacc, int tau_q, double win_u, double mix_c, double mix_lnc)
{
  bpa_engine * e = s->eng;
  s->epoch++;
  if (!s->allreduce)
    hipLaunchKernelGGL(gsm::gsum_decide_kernel, dim3(1), dim3(1024), 0, e->stream, (const double *)s->g_lnlcur.p, s->g_lnl.p, s->g_delta.p, s->g_active.p, s->nloci,
                       (double *)nullptr, 1, uacc, s->epoch, s->flag.p, s->counters.p, s->taus.p, s->sp, tau_q, win_u, mix_c, mix_lnc);
  else
  {
    double * out = s->sum_ext ? s->sum_ext : s->mix_sum.p;
    hipLaunchKernelGGL(gsm::gsum_decide_kernel, dim3(1), dim3(1024), 0, e->stream, (const double *)s->g_lnlcur.p, s->g_lnl.p, s->g_delta.p, s->g_active.p, s->nloci,
                       out, 0, uacc, s->epoch, s->flag.p, s->counters.p, s->taus.p, s->sp, tau_q, win_u, mix_c, mix_lnc);
    if (!s->allreduce(s->allreduce_ctx, out, 1u, (void *)e->stream)) return fail("bpa_sampler: the all-reduce callback failed");
    hipLaunchKernelGGL(smp::decide_kernel, dim3(1), dim3(1), 0, e->stream, out, uacc, s->epoch, s->flag.p, s->counters.p, s->taus.p, s->sp,
                       tau_q, -1, win_u, mix_c, mix_lnc);
    s->launches++;
  }
  HIPCHK(hipGetLastError());
  s->launches++;
  return 1;
}

static int gs_initialize(bpa_sampler * s)
{
  if (!gs_step(s, 5)) return 0;
  return gs_eval(s, 1);
}

// ======================= the program's THETA / TAU / MIX on a generic sampler: decisions on the host =======================
// theta_step_gibbs / tau_step / mix_step of csrc/host/a00_driver.c, statement for statement on the host side (same global
// stream, same order of draws: [first TAU's window] [THETA: choices + windows] [Gibbs variates] [acceptance numbers] [TAU:
// variates, acceptance] [next window] ...), the loci's part — proposal, likelihood, the sums — on the device.
static void gs_declog(const char * what, int k, double lnacc, int acc)
{
  static const bool on = getenv("A00_DECLOG") != nullptr;          // (the host driver's switch: the two logs line up)
  if (on) fprintf(stderr, "[gsp] %s %d lnacc %.17g u -1 -> %d\n", what, k, lnacc, acc);
}

// the decisions on the device (gdec_kernel, gsampler.hpp) unless BPA_GS_HOSTDEC=1 (the host form below: the trajectory
// reference).  Several ranks: the sums — doubles in the callback's own format — go through the all-reduce callback ON THE STREAM
// between the sum kernel and gdec_kernel (gs_dev_allreduce): with a stream-ordered collective (RCCL) no host waits for anything
static bool gs_prog_dev_wanted(const bpa_sampler * s)
{
  return !s->env_hostdec;
}

// the device's counters by move type and its copy of the global stream come back to the host's (adapt_finetune, a download)
static int gs_prog_pull(bpa_sampler * s)
{
  if (!s->gp_dev || !s->gp_mirror || !s->g_dst.p) return 1;
  bpa_engine * e = s->eng;
  gsm::GDecState st;
  HIPCHK(hipMemcpyAsync(&st, s->g_dst.p, sizeof st, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int k = 4; k < 10; ++k) s->gp_pj[k] += st.pj[k];
  s->grng = (a00_rng_t)st.z;
  HIPCHK(hipMemsetAsync(reinterpret_cast<char *>(s->g_dst.p) + offsetof(gsm::GDecState, pj), 0, sizeof st.pj, e->stream));
  return 1;
}

static int gs_prog_ready(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  if (!s->g_t2h3.p && (!s->g_t2h3.reserve((size_t)3*s->nloci) || !s->g_progout.reserve(64))) return fail("out of device memory (program moves)");
  if (!s->gp_mirror && gs_prog_dev_wanted(s))
  {
    // the decisions' state starts on the device: the global stream where the host left it, no sums yet
    if (!s->g_dst.reserve(1) || !s->g_dsum.reserve(4*smp::MAXPOP)) return fail("out of device memory (program moves)");
    gsm::GDecState st;
    std::memset(&st, 0, sizeof st);
    st.z = (uint32_t)(unsigned int)s->grng; st.mix_c = 1.0;
    const double qn = std::nan("");
    for (int p = 0; p < 16; ++p) { st.pf[p].a = st.pf[p].b = st.pf[p].c = qn; }
    HIPCHK(hipMemcpyAsync(s->g_dst.p, &st, sizeof st, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));            // (st is on this frame)
    s->gp_dev = true; s->gp_mirror = true; s->gp_ok = false;
    return 1;
  }
  if (!s->gp_mirror)
  {
    s->gp_dev = false;
    double t[3*smp::MAXPOP];
    HIPCHK(hipMemcpyAsync(t, s->taus.p, sizeof t, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int p = 0; p < smp::MAXPOP; ++p) { s->gp_tau[p] = t[p]; s->gp_theta[p] = t[smp::MAXPOP + p]; }
    s->gp_mirror = true; s->gp_ok = false;
  }
  return 1;
}

// where the sum kernels of the program's moves write and how the host gets it: pinned host memory the kernel stores into
// itself, its last store an arrival word the host polls — no copy launch, no wait for the launch to retire between the sums and
// the host's decision (config 5: a synchronisation 25 -> 16 -> ~8 us).  BPA_GS_PINOUT=1: pinned memory + hipStreamSynchronize,
// =0: device buffer + hipMemcpy.  A poll that sees nothing for 20 ms falls back to the stream's synchronisation (and its errors).
static int gs_prog_mode(const bpa_sampler * s) { return s->env_pinout; }
static double * gs_prog_out(bpa_sampler * s, unsigned long long * seq)
{
  *seq = 0;
  const int mode = gs_prog_mode(s);
  if (mode == 0) return s->g_progout.p;
  if (!s->gp_pin)
  {
    if (hipHostMalloc((void **)&s->gp_pin, 64*sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void **)&s->gp_pin_dev, s->gp_pin, 0) != hipSuccess)
    {
      (void)hipGetLastError();
      if (s->gp_pin) (void)hipHostFree(s->gp_pin);
      s->gp_pin = s->gp_pin_dev = nullptr;
      return s->g_progout.p;
    }
    std::memset(s->gp_pin, 0, 64*sizeof(double));
  }
  if (mode == 2) *seq = ++s->gp_seq;
  return s->gp_pin_dev;
}
static int gs_prog_fetch(bpa_sampler * s, const double * dev, void * host, size_t bytes, unsigned long long seq, int nflags)
{
  bpa_engine * e = s->eng;
  const bool pinned = s->gp_pin && dev == s->gp_pin_dev;
  if (!pinned) HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, e->stream));
  bool arrived = false;
  if (pinned && seq)
  {
    const unsigned long long * f = reinterpret_cast<const unsigned long long *>(s->gp_pin) + gsm::GPROG_FLAG0;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long spins = 0; !arrived; ++spins)
    {
      arrived = true;
      for (int q = 0; q < nflags; ++q) arrived = arrived && __atomic_load_n(f + q, __ATOMIC_ACQUIRE) == seq;
      if (!arrived && (spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
  }
  if (!arrived) HIPCHK(hipStreamSynchronize(e->stream));
  if (pinned) std::memcpy(host, s->gp_pin, bytes);
  return 1;
}

// several ranks (bpa_sampler_set_allreduce): the host decides from the sums over ALL ranks' loci.  This rank's sums go through the
// caller's collective as doubles, BPA_SAMPLER_SUMS at a time: a likelihood / Jacobian sum as it is (the same bits on every rank
// after the all-reduce), a non-negative 64-bit integer sum (coalescence counts, 2^-40 fixed-point T2h) as two doubles holding its
// upper and lower 32 bits — exact for any number of ranks the double's 53 bits can count to
static int gs_prog_allreduce(bpa_sampler * s, double * v, unsigned n)
{
  bpa_engine * e = s->eng;
  double * ar = s->sum_ext ? s->sum_ext : s->theta_sums.p;
  for (unsigned o = 0; o < n; o += BPA_SAMPLER_SUMS)
  {
    const unsigned c = std::min<unsigned>(BPA_SAMPLER_SUMS, n - o);
    HIPCHK(hipMemcpyAsync(ar, v + o, c*sizeof(double), hipMemcpyHostToDevice, e->stream));
    if (!s->allreduce(s->allreduce_ctx, ar, c, (void *)e->stream)) return fail("bpa_sampler: the all-reduce callback failed");
    HIPCHK(hipMemcpyAsync(v + o, ar, c*sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  return 1;
}
static void gs_split64(long long x, double * hi, double * lo) { *hi = (double)(x >> 32); *lo = (double)(x & 0xffffffffll); }
static long long gs_join64(double hi, double lo) { return (long long)hi*4294967296ll + (long long)lo; }

static int gs_prog_apply(bpa_sampler * s, const gsm::GApply & a, bool with_flag)
{
  bpa_engine * e = s->eng;
  if (with_flag) s->epoch++;
  hipLaunchKernelGGL(gsm::gprog_apply_kernel, dim3(1), dim3(1), 0, e->stream, a, s->epoch, s->flag.p, s->counters.p, s->taus.p, s->sp.npop);
  HIPCHK(hipGetLastError());
  s->launches++;
  return 1;
}

static int gs_prog_theta(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  const int npop = s->sp.npop;
  unsigned int gz = (unsigned int)s->grng;
  int slide[smp::MAXPOP]; double tnew[smp::MAXPOP], fa[smp::MAXPOP], fb[smp::MAXPOP];
  uint32_t onmask = 0;
  for (int p = 0; p < npop; ++p)
  {
    slide[p] = 0; tnew[p] = s->gp_theta[p];
    if (!s->has_theta[p]) continue;
    onmask |= 1u << p;
    slide[p] = a00_bpp_rndu(&gz) < s->sp.theta_slide_prob;
    if (slide[p]) tnew[p] = a00_reflect(s->gp_theta[p] + s->sp.ft_theta*a00_bpp_rnd_symmetrical(&gz), 0.0, 999.0);
  }
  long long h[3*smp::MAXPOP];
  unsigned long long pseq = 0;
  double * pout = gs_prog_out(s, &pseq);
  hipLaunchKernelGGL(gsm::gprog_theta_sums_kernel, dim3(npop), dim3(1024), 0, e->stream, s->pop_nc.p, s->pop_t2h.p, s->nloci, onmask,
                     reinterpret_cast<long long *>(pout), pseq);
  HIPCHK(hipGetLastError());
  if (!gs_prog_fetch(s, pout, h, (size_t)3*npop*sizeof(long long), pseq, npop)) return 0;
  if (s->allreduce)
  {
    double v[4*smp::MAXPOP];
    for (int p = 0; p < npop; ++p)
    {
      const bool neg = h[3*p] < 0 || h[3*p + 1] < 0;         // (never: counts and waiting times)
      v[4*p] = (double)(neg ? 0 : h[3*p]); gs_split64(neg ? 0 : h[3*p + 1], &v[4*p + 1], &v[4*p + 2]); v[4*p + 3] = (neg || h[3*p + 2] != 0) ? 1.0 : 0.0;
    }
    if (!gs_prog_allreduce(s, v, 4u*(unsigned)npop)) return 0;
    for (int p = 0; p < npop; ++p) { h[3*p] = (long long)v[4*p]; h[3*p + 1] = gs_join64(v[4*p + 1], v[4*p + 2]); h[3*p + 2] = v[4*p + 3] != 0.0 ? 1 : 0; }
  }
  s->launches++;
  bool bad = false;
  for (int p = 0; p < npop; ++p) { bad = bad || h[3*p + 2] != 0; s->gp_k[p] = h[3*p]; s->gp_T[p] = (double)h[3*p + 1]*(1.0/1099511627776.0); }
  s->gp_ok = !bad;
  for (int p = 0; p < npop; ++p)
  {
    fa[p] = fb[p] = NAN;
    if (!s->has_theta[p] || bad || slide[p]) continue;
    a00_theta_conditional_invgamma(s->sp.theta_alpha, s->sp.theta_beta, (long)s->gp_k[p], s->gp_T[p], &fa[p], &fb[p]);
    if (fa[p] == fa[p]) tnew[p] = 1/(a00_bpp_rndgamma(&gz, fa[p])/fb[p]);
  }
  gsm::GApply a{};
  a.accept = 1u; a.set_tau_q = 0xffffffffu;
  for (int p = 0; p < npop; ++p)
  {
    double lnacc = NAN; bool acc = false;
    if (!s->has_theta[p]) continue;
    a.nprop++;
    const double T = s->gp_T[p];
    if (!bad)
    {
      if (slide[p]) lnacc = a00_theta_lnacc((long)s->gp_k[p], T, s->gp_theta[p], tnew[p], s->sp.theta_alpha, s->sp.theta_beta);
      else if (fa[p] == fa[p])
        lnacc = a00_theta_lnacc((long)s->gp_k[p], T, s->gp_theta[p], tnew[p], s->sp.theta_alpha, s->sp.theta_beta)
              + a00_theta_gibbs_hastings(fa[p], fb[p], s->gp_theta[p], tnew[p]);
      acc = lnacc == lnacc && tnew[p] > 0 && (lnacc >= -1e-10 || a00_bpp_rndu(&gz) < std::exp(lnacc));
    }
    gs_declog(slide[p] ? "theta" : "thetag", p, lnacc, acc);
    if (slide[p]) { s->gp_pj[8]++; s->gp_pj[9] += acc ? 1u : 0u; }
    if (acc) { a.nacc++; s->gp_theta[p] = tnew[p]; a.theta_mask |= 1u << p; a.theta[p] = tnew[p]; if (!slide[p]) a.ngacc++; }
    if (!slide[p]) a.ngprop++;
  }
  s->grng = (a00_rng_t)gz;
  s->logpr_stale = true;
  return gs_prog_apply(s, a, false);
}

static int gs_prog_tau(bpa_sampler * s, int q)
{
  bpa_engine * e = s->eng;
  unsigned int gz = (unsigned int)s->grng;
  const int cl = s->sp.left[q], cr = s->sp.right[q], pq = s->sp.parent[q];
  const int aff[3] = { q, cl, cr };
  const double old = s->gp_tau[q], lo = std::fmax(s->gp_tau[cl], s->gp_tau[cr]), hi = pq >= 0 ? s->gp_tau[pq] : 999.0;
  const double w = s->gp_pre_valid ? s->gp_pre_window : a00_bpp_rnd_symmetrical(&gz);
  s->gp_pre_valid = false;
  const double tnew = a00_reflect(old + s->sp.ft_tau*w, lo, hi);
  s->grng = (a00_rng_t)gz;
  if (!gs_step(s, 2, (unsigned)q, 0.0, 1.0, 0.0, w) || !gs_eval(s, 1)) return 0;
  double out[8];
  unsigned long long pseq = 0;
  double * pout = gs_prog_out(s, &pseq);
  hipLaunchKernelGGL(gsm::gprog_sums_kernel, dim3(1), dim3(1024), 0, e->stream, (const double *)s->g_lnlcur.p, (const double *)s->g_lnl.p, (const double *)s->g_delta.p,
                     (const uint8_t *)s->g_active.p, (const double *)s->g_t2h3.p, s->nloci, 1, pout, pseq);
  HIPCHK(hipGetLastError());
  if (!gs_prog_fetch(s, pout, out, 5*sizeof(double), pseq, 1)) return 0;
  s->launches++;
  if (s->allreduce)
  {
    long long * ow = reinterpret_cast<long long *>(out);
    bool neg = false;
    for (int j = 1; j <= 3; ++j) neg = neg || ow[j] < 0;
    double v[8] = { out[0], 0, 0, 0, 0, 0, 0, (neg || ow[4] != 0) ? 1.0 : 0.0 };
    for (int j = 0; j < 3; ++j) gs_split64(neg ? 0 : ow[1 + j], &v[1 + 2*j], &v[2 + 2*j]);
    if (!gs_prog_allreduce(s, v, 8u)) return 0;
    out[0] = v[0];
    for (int j = 0; j < 3; ++j) ow[1 + j] = gs_join64(v[1 + 2*j], v[2 + 2*j]);
    ow[4] = v[7] != 0.0 ? 1 : 0;
  }
  double sum = out[0];
  const long long * ol = reinterpret_cast<const long long *>(out);
  const bool bad = ol[4] != 0;
  if (pq < 0 && s->sp.tau_alpha > 0) sum += (s->sp.tau_alpha - 1 - (s->sp.S - 1) + 1)*std::log(tnew/old) - s->sp.tau_beta*(tnew - old);
  gsm::GApply a{};
  a.set_tau_q = (uint32_t)q; a.tau_new = tnew; a.nprop = 1;
  double oldtheta[3], Cn[3] = {0, 0, 0};
  for (int j = 0; j < 3; ++j)
  {
    const int p = aff[j]; const double to = s->gp_theta[p]; const long k = (long)s->gp_k[p];
    oldtheta[j] = to;
    if (!s->has_theta[p]) continue;
    if (bad || !s->gp_ok) { sum = NAN; continue; }
    double a1, b1, a1o, b1o;
    Cn[j] = (double)ol[1 + j]*(1.0/1099511627776.0);
    a00_theta_conditional_invgamma(s->sp.theta_alpha, s->sp.theta_beta, k, Cn[j], &a1, &b1);
    a00_theta_conditional_invgamma(s->sp.theta_alpha, s->sp.theta_beta, k, s->gp_T[p], &a1o, &b1o);
    if (!(a1 == a1 && a1o == a1o)) { sum = NAN; continue; }
    const double g = a00_bpp_rndgamma(&gz, a1);
    const double tn = 1.0/(g/b1);
    sum += (a00_invgamma_logpdf(to, a1o, b1o) - a00_invgamma_logpdf(tn, a1, b1))
         + ((s->sp.theta_alpha - 1)*std::log(tn/to) - s->sp.theta_beta*(tn - to))
         + (k*(std::log(2.0/tn) - std::log(2.0/to)) - (Cn[j]/tn - s->gp_T[p]/to));
    s->gp_theta[p] = tn;
  }
  const bool acc = sum >= -1e-10 || a00_bpp_rndu(&gz) < std::exp(sum);
  gs_declog("tau", q, sum, acc);
  s->gp_pj[4]++; s->gp_pj[5] += acc ? 1u : 0u;
  s->grng = (a00_rng_t)gz;
  if (acc)
  {
    a.accept = 1u; a.nacc = 1;
    s->gp_tau[q] = tnew;
    for (int j = 0; j < 3; ++j) if (s->has_theta[aff[j]]) { s->gp_T[aff[j]] = Cn[j]; a.theta_mask |= 1u << aff[j]; a.theta[aff[j]] = s->gp_theta[aff[j]]; }
    s->logpr_stale = true;                         // (the accepted trees' densities were taken with the old thetas)
  }
  else for (int j = 0; j < 3; ++j) s->gp_theta[aff[j]] = oldtheta[j];
  return gs_prog_apply(s, a, true);
}

static int gs_prog_mix(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  unsigned int gz = (unsigned int)s->grng;
  const int npop = s->sp.npop;
  const double lnc = s->sp.ft_mix*a00_bpp_rnd_symmetrical(&gz), c = std::exp(lnc);
  double oldtheta[smp::MAXPOP], lnacc_theta = 0;
  for (int p = 0; p < npop; ++p) oldtheta[p] = s->gp_theta[p];
  for (int p = 0; p < npop; ++p)
  {
    const double to = oldtheta[p]; const long k = (long)s->gp_k[p];
    if (!s->has_theta[p]) continue;
    if (!s->gp_ok) { lnacc_theta = NAN; continue; }
    double a1, b1, a1o, b1o;
    const double Ts = s->gp_T[p]*c;
    a00_theta_conditional_invgamma(s->sp.theta_alpha, s->sp.theta_beta, k, Ts, &a1, &b1);
    a00_theta_conditional_invgamma(s->sp.theta_alpha, s->sp.theta_beta, k, Ts/c, &a1o, &b1o);
    if (!(a1 == a1 && a1o == a1o)) { lnacc_theta = NAN; continue; }
    const double g = a00_bpp_rndgamma(&gz, a1);
    const double tn = 1.0/(g/b1);
    lnacc_theta += (a00_invgamma_logpdf(to, a1o, b1o) - a00_invgamma_logpdf(tn, a1, b1))
                 + ((s->sp.theta_alpha - 1)*std::log(tn/to) - s->sp.theta_beta*(tn - to))
                 + (k*(std::log(2.0/tn) - std::log(2.0/to)) - (Ts/tn - s->gp_T[p]/to));
    s->gp_theta[p] = tn;
  }
  s->grng = (a00_rng_t)gz;
  if (!gs_step(s, 3, 0, 0.0, c, lnc) || !gs_eval(s, 1)) return 0;
  double out[8];
  unsigned long long pseq = 0;
  double * pout = gs_prog_out(s, &pseq);
  hipLaunchKernelGGL(gsm::gprog_sums_kernel, dim3(1), dim3(1024), 0, e->stream, (const double *)s->g_lnlcur.p, (const double *)s->g_lnl.p, (const double *)s->g_delta.p,
                     (const uint8_t *)s->g_active.p, (const double *)s->g_t2h3.p, s->nloci, 0, pout, pseq);
  HIPCHK(hipGetLastError());
  if (!gs_prog_fetch(s, pout, out, 5*sizeof(double), pseq, 1)) return 0;
  s->launches++;
  if (s->allreduce && !gs_prog_allreduce(s, out, 1u)) return 0;
  const int root = npop - 1;
  double lnacc = out[0] + (double)(s->sp.S - 1)*lnc;
  if (s->sp.tau_alpha > 0)
    lnacc += (s->sp.tau_alpha - 1)*lnc - s->sp.tau_beta*(s->gp_tau[root]*c - s->gp_tau[root]) - (double)(s->sp.S - 2)*lnc;
  lnacc += lnacc_theta;
  const bool acc = lnacc >= -1e-10 || a00_bpp_rndu(&gz) < std::exp(lnacc);
  gs_declog("mix", 0, lnacc, acc);
  s->gp_pj[6]++; s->gp_pj[7] += acc ? 1u : 0u;
  s->grng = (a00_rng_t)gz;
  gsm::GApply a{};
  a.set_tau_q = 0xffffffffu; a.nprop = 1;
  if (acc)
  {
    a.accept = 1u; a.nacc = 1; a.scale_taus = 1u; a.mix_c = c;
    for (int p = 0; p < npop; ++p) { s->gp_tau[p] *= c; s->gp_T[p] *= c; if (s->has_theta[p]) { a.theta_mask |= 1u << p; a.theta[p] = s->gp_theta[p]; } }
    s->logpr_stale = true;
  }
  else for (int p = 0; p < npop; ++p) s->gp_theta[p] = oldtheta[p];
  return gs_prog_apply(s, a, true);
}

// ---- the same three steps with the decision on the device: sums -> [the ranks' collective] -> gdec_kernel, no synchronisation
static uint32_t gs_theta_mask(const bpa_sampler * s)
{
  uint32_t m = 0;
  for (int p = 0; p < s->sp.npop; ++p) if (s->has_theta[p]) m |= 1u << p;
  return m;
}
// n doubles of g_dsum summed over the ranks, BPA_SAMPLER_SUMS at a time, in the callback's buffer (the caller's device_sums or
// the sampler's own): copy in, collective, copy back — all enqueued on the engine's stream
static int gs_dev_allreduce(bpa_sampler * s, unsigned n)
{
  if (!s->allreduce) return 1;
  bpa_engine * e = s->eng;
  double * ar = s->sum_ext ? s->sum_ext : s->theta_sums.p;
  for (unsigned o = 0; o < n; o += BPA_SAMPLER_SUMS)
  {
    const unsigned c = std::min<unsigned>(BPA_SAMPLER_SUMS, n - o);
    HIPCHK(hipMemcpyAsync(ar, s->g_dsum.p + o, c*sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    if (!s->allreduce(s->allreduce_ctx, ar, c, (void *)e->stream)) return fail("bpa_sampler: the all-reduce callback failed");
    HIPCHK(hipMemcpyAsync(s->g_dsum.p + o, ar, c*sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  }
  return 1;
}
template <int PHASE>
static int gs_dec_launch(bpa_sampler * s, int q, int next)
{
  bpa_engine * e = s->eng;
  hipLaunchKernelGGL((gsm::gdec_kernel<PHASE>), dim3(1), dim3(64), sizeof(smp2::WgBase), e->stream, s->g_dst.p, (const double *)s->g_dsum.p, s->sp,
                     gs_theta_mask(s), q, next, s->epoch, s->flag.p, s->counters.p, s->taus.p);
  HIPCHK(hipGetLastError());
  s->launches++;
  return 1;
}
static int gs_dev_theta(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  hipLaunchKernelGGL(gsm::gdec_theta_sums_kernel, dim3(s->sp.npop), dim3(1024), 0, e->stream, s->pop_nc.p, s->pop_t2h.p, s->nloci, gs_theta_mask(s), s->g_dsum.p);
  HIPCHK(hipGetLastError());
  s->launches++;
  s->logpr_stale = true;
  if (!gs_dev_allreduce(s, 4u*(unsigned)s->sp.npop)) return 0;
  return gs_dec_launch<0>(s, -1, s->sp.S);
}
static int gs_dev_allloci(bpa_sampler * s, int q /* -1: MIX */)
{
  bpa_engine * e = s->eng;
  const bool mix = q < 0;
  if (!gs_step(s, mix ? 3u : 2u, mix ? 0u : (unsigned)q) || !gs_eval(s, 1)) return 0;
  hipLaunchKernelGGL(gsm::gdec_sums_kernel, dim3(1), dim3(1024), 0, e->stream, (const double *)s->g_lnlcur.p, (const double *)s->g_lnl.p, (const double *)s->g_delta.p,
                     (const uint8_t *)s->g_active.p, (const double *)s->g_t2h3.p, s->nloci, mix ? 0 : 1, s->g_dsum.p);
  HIPCHK(hipGetLastError());
  s->launches++;
  s->epoch++;
  // (the host does not know the decision: the next step takes every tree's density again — the lane groups compute its terms anyway)
  s->logpr_stale = true;
  if (!gs_dev_allreduce(s, mix ? 1u : 8u)) return 0;
  return mix ? gs_dec_launch<2>(s, -1, -1) : gs_dec_launch<1>(s, q, q + 1);
}

static int gs_iterate(bpa_sampler * s, unsigned iterations)
{
  bpa_engine * e = s->eng;
  if (!gs_subst_ready(s)) return 0;
  s->host_current = false;
  const bool prog = gs_prog(s);
  if (s->kernel_bpp && !prog) return fail("bpa_sampler: on a generic sampler BPP's proposal kernel comes with the program's moves (bpa_sampler_set_program_moves) and a theta prior");
  if (prog && !gs_prog_ready(s)) return 0;
  for (unsigned it = 0; it < iterations; ++it)
  {
    // the per-locus proposals, "step j of every locus" (gage_step / gspr_step of a00_driver.c): each launch first settles
    // the step before it
    if (gs_chain_wanted(s)) { if (!gs_chain(s)) return 0; }
    else
    {
      for (unsigned k = 0; k + 1 < s->maxtips; ++k)     { if (!gs_step(s, 0, k) || !gs_eval(s, 0)) return 0; }
      for (unsigned k = 0; k + 2 < 2*s->maxtips; ++k)   { if (!gs_step(s, 1, k) || !gs_eval(s, 0)) return 0; }
    }
    s->sweeps++;
    if (s->env_nomix) continue;
    if (prog)
    {
      // (a00_iterate: the first TAU's window comes before the THETA step's numbers in the global stream)
      bool any = false;
      for (int p = 0; p < s->sp.npop; ++p) any = any || s->has_theta[p];
      if (s->gp_dev)
      {
        // Nothing in an iteration makes the host wait any more, so it would run thousands of launches ahead of the GPU (a 400-iteration
        // burn-in of config 4 = 63 600 launches queued; under rocprofv3's counter collection that never finished).  It is paced
        // instead: before an iteration's all-loci steps it waits for the END of the iteration before the last — work the GPU
        // finished long ago unless the host is the faster of the two; no decision depends on this wait.
        if (s->gp_pace_n >= 2) HIPCHK(hipEventSynchronize(s->gp_pace[(s->gp_pace_n - 2) & 3u]));
        if (!gs_step(s, 4) || !gs_dev_theta(s)) return 0;
        for (int q = s->sp.S; q < s->sp.npop; ++q) if (!gs_dev_allloci(s, q)) return 0;
        if (!gs_dev_allloci(s, -1)) return 0;
        hipEvent_t & ev = s->gp_pace[s->gp_pace_n & 3u];
        if (!ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev, e->stream));
        s->gp_pace_n++;
      }
      else
      {
      if (any) { unsigned int gz = (unsigned int)s->grng; s->gp_pre_window = a00_bpp_rnd_symmetrical(&gz); s->gp_pre_valid = true; s->grng = (a00_rng_t)gz; }
      if (!gs_step(s, 4) || !gs_prog_theta(s)) return 0;
      for (int q = s->sp.S; q < s->sp.npop; ++q) if (!gs_prog_tau(s, q)) return 0;
      if (!gs_prog_mix(s)) return 0;
      }
    }
    else
    {
    if (s->sp.theta_alpha > 0)
    {
      // settle the last per-locus step and leave the statistics of every tree's density; then THETA as in the sweep path
      if (!gs_step(s, 4)) return 0;
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
        double * ar = s->sum_ext ? s->sum_ext : s->theta_sums.p;
        const size_t nb = (size_t)s->sp.npop*sizeof(double);
        if (ar != s->theta_sums.p) HIPCHK(hipMemcpyAsync(ar, s->theta_sums.p, nb, hipMemcpyDeviceToDevice, e->stream));
        if (!s->allreduce(s->allreduce_ctx, ar, (unsigned)s->sp.npop, (void *)e->stream)) return fail("bpa_sampler: the all-reduce callback failed");
        if (ar != s->theta_sums.p) HIPCHK(hipMemcpyAsync(s->theta_sums.p, ar, nb, hipMemcpyDeviceToDevice, e->stream));
        hipLaunchKernelGGL(smp::theta_sum_decide_kernel, dim3(s->sp.npop), dim3(1024), 0, e->stream, s->pop_nc.p, s->pop_t2h.p,
                           s->nloci, s->taus.p, s->sp, ta, s->counters.p, (double *)nullptr, (const double *)s->theta_sums.p, 1);
      }
      HIPCHK(hipGetLastError());
      s->logpr_stale = true;
      s->launches += 1;
    }
    for (int q = s->sp.S; q < s->sp.npop; ++q)                    // one rubber-band step per species divergence
    {
      const double uprop = a00_rndu(&s->grng), uacc_t = a00_rndu(&s->grng);
      if (!gs_step(s, 2, (unsigned)q, uprop) || !gs_eval(s, 1) || !gs_decide(s, uacc_t, q, uprop, 1.0, 0.0)) return 0;
    }
    const double lnc = s->sp.ft_mix*(a00_rndu(&s->grng) - 0.5), c = std::exp(lnc);
    const double uacc = a00_rndu(&s->grng);
    if (!gs_step(s, 3, 0, 0.0, c, lnc) || !gs_eval(s, 1) || !gs_decide(s, uacc, -1, 0.0, c, lnc)) return 0;
    }
    // the substitution-parameter moves come last (method.c:5699-5735; param_step of a00_driver.c)
    if (s->g_ft[0] > 0) for (unsigned j = 0; j < 3; ++j)  { if (!gs_step(s, 6, j) || !gs_eval(s, 0)) return 0; }
    if (s->g_ft[1] > 0) for (unsigned j = 0; j < 6; ++j)  { if (j != 1 && (!gs_step(s, 7, j) || !gs_eval(s, 0))) return 0; }
    if (s->g_ft[2] > 0)                                   { if (!gs_step(s, 8, 0) || !gs_eval(s, 0)) return 0; }
    if (s->g_ft[0] > 0 || s->g_ft[1] > 0 || s->g_ft[2] > 0)
      for (bpa_locus * l : s->loci) l->host_par_stale = true;            // the device blocks moved ahead of the host mirrors
  }
  return gs_join(s);                 // whoever uses the engine's stream next sees both halves
}

// The steps' evaluations left the roots' CLV buffers unstored (gs_skip_root_flag): with the last step settled, every buffer of
// every locus is recomputed in place from the settled trees — the start-up evaluation (mode 5: no toggles, nothing drawn, nothing
// counted), every parent stored, its result committed (same P-matrices, same CLVs, same sums: the values the chain already
// holds, tests/test_gpu_gsampler.py) — so that what reads a locus's buffers after a download finds them as a step-by-step
// caller of the reference's API would have left them.
static int gs_level_roots(bpa_sampler * s)
{
  if (!s->g_root_stale) return 1;
  s->g_level_eval = true;
  const int ok = gs_step(s, 5) && gs_eval(s, 1) && gs_step(s, 4);
  s->g_level_eval = false;
  if (ok) s->g_root_stale = false;
  return ok;
}

// settle whatever is pending and bring the trees to the host
static int gs_download(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  if (!gs_step(s, 4) || !gs_level_roots(s) || !gs_prog_pull(s)) return 0;
  // the settle launch rolls rejected frequency / exchangeability proposals back in the parameter blocks: the loci's
  // eigensystems must follow before anyone else (bpa_batch_evaluate, a plan) computes P-matrices from them
  if (!gs_refresh_eigen(s)) return 0;
  HIPCHK(hipMemcpyAsync(s->g_trees.data(), s->g_dev.p, s->nloci*sizeof(gsm::GTree), hipMemcpyDeviceToHost, e->stream));
  if (s->g_sm.p && !s->g_sm_host.empty())
    HIPCHK(hipMemcpyAsync(s->g_sm_host.data(), s->g_sm.p, s->g_sm_host.size()*sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 1;
}

// ---- substitution-parameter moves: the sampler's copy of every locus's values and the window widths
extern "C" int bpa_sampler_set_subst_model(bpa_sampler_t * s, unsigned i, const double * freqs, const double * qrates, double alpha)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) { unsigned j; bpa_sampler * p = comp_part(s, i, &j); if (!p) return fail("bpa_sampler_set_subst_model: bad argument"); return bpa_sampler_set_subst_model(p, j, freqs, qrates, alpha); }
  if (i >= s->nloci || !freqs || !qrates || !(alpha > 0)) return fail("bpa_sampler_set_subst_model: bad argument");
  if (!s->generic) return fail("bpa_sampler_set_subst_model: the loci are JC69 (no substitution parameters to move)");
  if (s->g_sm_host.size() != (size_t)s->nloci*11) s->g_sm_host.assign((size_t)s->nloci*11, 0.0);
  if (s->uploaded)
    return fail("bpa_sampler_set_subst_model: the sampler is running — set every locus's starting values before bpa_sampler_initialize");
  double * m = s->g_sm_host.data() + (size_t)i*11;
  std::copy(freqs, freqs + 4, m); std::copy(qrates, qrates + 6, m + 4); m[10] = alpha;
  return 1;
}

extern "C" int bpa_sampler_get_subst_model(bpa_sampler_t * s, unsigned i, double * freqs, double * qrates, double * alpha)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) { unsigned j; bpa_sampler * p = comp_part(s, i, &j); if (!p) return fail("bpa_sampler_get_subst_model: bad argument"); return comp_upload(s) && bpa_sampler_get_subst_model(p, j, freqs, qrates, alpha); }
  if (i >= s->nloci || s->g_sm_host.size() != (size_t)s->nloci*11) return fail("bpa_sampler_get_subst_model: no substitution model set");
  if (s->uploaded && !s->host_current && !sampler_download(s)) return 0;
  const double * m = s->g_sm_host.data() + (size_t)i*11;
  if (freqs) std::copy(m, m + 4, freqs);
  if (qrates) std::copy(m + 4, m + 10, qrates);
  if (alpha) *alpha = m[10];
  return 1;
}

extern "C" void bpa_sampler_set_subst_moves(bpa_sampler_t * s, double ft_freqs, double ft_qrates, double ft_alpha, double alpha_a, double alpha_b)
{
  std::lock_guard<std::recursive_mutex> lock_(s->eng->mtx);
  if (s->comp) { (void)comp_each(s, [&](bpa_sampler * p) { if (p->generic) bpa_sampler_set_subst_moves(p, ft_freqs, ft_qrates, ft_alpha, alpha_a, alpha_b); return 1; }); return; }
  // (nothing to upload: the widths travel with every launch, the moves' tables are made when first needed — gs_subst_ready)
  s->g_ft[0] = ft_freqs; s->g_ft[1] = ft_qrates; s->g_ft[2] = ft_alpha; s->g_alpha_a = alpha_a; s->g_alpha_b = alpha_b;
}
