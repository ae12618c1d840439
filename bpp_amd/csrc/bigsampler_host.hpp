// bigsampler_host.hpp — host side of the big-tree device sampler (bigsampler.hpp): uploads, the launch loop of an iteration,
// downloads.  Included at the end of sampler.hpp, after gsampler_host.hpp (whose buffers and all-loci kernels it shares).
#pragma once

static int gb_upload(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  const unsigned T = s->nloci;
  if (!flush_state(e)) return 0;                    // (tip states, weights, parameter blocks, eigensystems on the device)
  for (unsigned i = 0; i < T; ++i)
  {
    gbig::BTree & t = s->b_trees[i];
    if (!assign_pops_host(s, t)) return 0;
    // the scaler that goes with an inner node's CURRENT CLV buffer (gtree.c:2433-2439 at the start: its index among the
    // inner nodes; swap_clv toggles the two together, so scaler = clv - tips holds in mid-run too)
    for (int k = 0; k < 2*t.tips - 1; ++k) t.scaler[k] = (int16_t)((s->loci[i]->scale_buffers && k >= t.tips) ? t.clv[k] - t.tips : BPA_SCALE_BUFFER_NONE);
    for (int p = 0; p < smp::MAXPOP; ++p) t.gl[p] = 0;
    for (int k = 0; k < t.tips; ++k) for (int q = t.pop[k]; q >= 0; q = s->sp.parent[q]) t.gl[q]++;
  }
  for (int p = 0; p < smp::MAXPOP; ++p) s->has_theta[p] = p >= s->sp.S && p < s->sp.npop;
  std::vector<uint32_t> tl(T), tp(T + 1), thr;
  unsigned off = 0;
  for (unsigned i = 0; i < T; ++i)
  {
    const gbig::BTree & t = s->b_trees[i];
    int cnt[smp::MAXPOP] = {0};
    for (int k = 0; k < t.tips; ++k) if (++cnt[t.pop[k]] >= 2) s->has_theta[t.pop[k]] = true;
    tl[i] = s->loci[i]->id; tp[i] = off; off += s->loci[i]->sites;
    for (unsigned n = 0; n < s->loci[i]->sites; ++n) thr.push_back(i);
  }
  tp[T] = off;
  s->g_npat = off;
  s->g_maxmat = 2*s->maxtips - 2; s->g_maxops = s->maxtips - 1;
  const size_t nmat = (size_t)T*s->g_maxmat;
  uint32_t zero2[2] = {0, 0};
  if (!upload(s->b_dev, s->b_trees.data(), T) || !s->b_undo.reserve(T) || !upload(s->g_tlocus, tl.data(), T) || !upload(s->g_tpat, tp.data(), T + 1) ||
      !upload(s->b_thr, thr.data(), thr.size()) || !s->g_rscaler.reserve(T) || !s->g_ops20.reserve((size_t)T*s->g_maxops) ||
      !s->g_oprng.reserve((size_t)2*T) || !s->g_root20.reserve(T) || !s->g_mtask.reserve(nmat) || !s->g_mpm.reserve(nmat) || !s->g_len.reserve(nmat) ||
      !s->g_lnl.reserve(T) || !s->g_lnlcur.reserve(T) || !s->g_hast.reserve(T) || !s->g_logpr.reserve(T) || !s->g_delta.reserve(T) || !s->g_active.reserve(T) ||
      !s->g_site.reserve(off) || !upload(s->flag, zero2, 1) || !upload(s->counters, s->h_counters, 4) || !s->mix_sum.reserve(1) ||
      !upload(s->taus, s->h_taus.data(), s->h_taus.size()) || !s->pop_t2h.reserve((size_t)T*smp::MAXPOP) ||
      !s->pop_nc.reserve((size_t)T*smp::MAXPOP) || !s->theta_sums.reserve(smp::MAXPOP))
    return 0;
  HIPCHK(hipMemsetAsync(s->g_oprng.p, 0, (size_t)2*T*sizeof(uint32_t), e->stream));
  HIPCHK(hipMemsetAsync(s->g_root20.p, 0, (size_t)T*sizeof(uint32_t), e->stream));
  HIPCHK(hipMemsetAsync(s->g_rscaler.p, 0xff, (size_t)T*sizeof(int32_t), e->stream));
  HIPCHK(hipMemsetAsync(s->g_mtask.p, 0xff, nmat*sizeof(uint32_t), e->stream));
  HIPCHK(hipMemsetAsync(s->g_mpm.p, 0, nmat*sizeof(uint32_t), e->stream));
  HIPCHK(hipMemsetAsync(s->g_len.p, 0, nmat*sizeof(double), e->stream));
  HIPCHK(hipMemsetAsync(s->g_lnl.p, 0, T*sizeof(double), e->stream));
  HIPCHK(hipMemsetAsync(s->g_hast.p, 0, T*sizeof(double), e->stream));
  HIPCHK(hipMemsetAsync(s->g_logpr.p, 0, T*sizeof(double), e->stream));
  HIPCHK(hipMemsetAsync(s->g_delta.p, 0, T*sizeof(double), e->stream));
  HIPCHK(hipMemsetAsync(s->g_active.p, 0, T, e->stream));
  if (s->allreduce) return fail("bpa_sampler: the big-tree sampler runs on one rank (a handful of loci: nothing to shard)");
  s->epoch = 0; s->mix_pending = false; s->g_pend = 0;
  s->uploaded = true;
  return 1;
}

// one big_step_kernel launch: settle what is pending, then propose `mode`
static int gb_step(bpa_sampler * s, unsigned mode, unsigned k = 0, double tau_u = 0, double mix_c = 1.0, double mix_lnc = 0)
{
  bpa_engine * e = s->eng;
  gbig::BArgs a{};
  a.trees = s->b_dev.p; a.undo = s->b_undo.p; a.T = s->nloci; a.mode = mode; a.k = k;
  a.pend = s->g_pend; a.lnl_new = s->g_lnl.p; a.hast = s->g_hast.p; a.logpr_new = s->g_logpr.p; a.delta = s->g_delta.p;
  a.active = s->g_active.p; a.flag = s->flag.p; a.epoch = s->epoch; a.lnl_cur = s->g_lnlcur.p;
  a.ops = s->g_ops20.p; a.op_rng = s->g_oprng.p; a.root_clv = s->g_root20.p; a.root_scaler = s->g_rscaler.p;
  a.mat_task = s->g_mtask.p; a.mat_pm = s->g_mpm.p; a.mat_length = s->g_len.p; a.maxmat = s->g_maxmat; a.maxops = s->g_maxops;
  a.taus = s->taus.p; a.tau_q = k; a.tau_u = tau_u; a.mix_c = mix_c; a.mix_lnc = mix_lnc;
  a.pop_nc = s->pop_nc.p; a.pop_t2h = s->pop_t2h.p;
  a.refresh_logpr = s->logpr_stale ? 1u : 0u; s->logpr_stale = false;
  a.sp = s->sp;
  hipLaunchKernelGGL(gbig::big_step_kernel, dim3(s->nloci), dim3(gbig::BBS), 0, e->stream, a);       // one workgroup per locus
  HIPCHK(hipGetLastError());
  s->launches++;
  s->g_pend = mode <= 1 ? 1u : (mode == 2 || mode == 3) ? 2u : mode == 5 ? 3u : 0u;
  return 1;
}

// the step's likelihood: the engine's general kernels over the records the step kernel wrote
static int gb_eval(bpa_sampler * s, int kind)
{
  bpa_engine * e = s->eng;
  if (!e->usedata) return 1;
  PlanDev d{};
  d.loci = e->d_loci.p; d.bfbeta = e->bfbeta; d.task_locus = s->g_tlocus.p; d.task_pat_off = s->g_tpat.p; d.thr_task = s->b_thr.p;
  d.op_off = s->g_oprng.p; d.ops = s->g_ops20.p; d.root_clv = s->g_root20.p; d.root_scaler = s->g_rscaler.p;
  d.site_term = s->g_site.p; d.lnl = s->g_lnl.p; d.mat_task = s->g_mtask.p; d.mat_pmatrix = s->g_mpm.p; d.mat_length = s->g_len.p;
  d.nmat = s->nloci*s->g_maxmat; d.ntasks = s->nloci; d.npatterns = s->g_npat; d.pad = s->g_rmax;
  hipEvent_t k0 = nullptr, k1 = nullptr;
  if (s->timing_stride && (s->timing_phase++ % s->timing_stride) == 0)
  {
    if (s->timed.size() >= 4096 && !sampler_timing_drain(s)) return 0;
    bpa_sampler::Timed t{nullptr, nullptr, kind};
    HIPCHK(hipEventCreate(&t.e0)); HIPCHK(hipEventCreate(&t.e1));
    s->timed.push_back(t);
    k0 = t.e0; k1 = t.e1;
  }
  d.flags = 1u | 64u;
  hipLaunchKernelGGL(pmatrix_s4_kernel, dim3((d.nmat*s->g_rmax + BPA_BLOCK - 1)/BPA_BLOCK), dim3(BPA_BLOCK), 0, e->stream, d, (uint32_t)s->g_rmax);
  d.flags = 2u | 4u | 64u;
  hipExtLaunchKernelGGL(partials_lnl_s4_kernel, dim3((d.npatterns + BPA_BLOCK - 1)/BPA_BLOCK), dim3(BPA_BLOCK), 0, e->stream, k0, k1, 0, d);
  hipLaunchKernelGGL(lnl_reduce_wave_kernel, dim3(s->nloci), dim3(64), 0, e->stream, d);
  HIPCHK(hipGetLastError());
  s->launches += 3; s->g_evals++;
  return 1;
}

static int gb_decide(bpa_sampler * s, double uacc, int tau_q, double win_u, double mix_c, double mix_lnc)
{
  bpa_engine * e = s->eng;
  s->epoch++;
  hipLaunchKernelGGL(gsm::gsum_decide_kernel, dim3(1), dim3(1024), 0, e->stream, (const double *)s->g_lnlcur.p, s->g_lnl.p, s->g_delta.p,
                     s->g_active.p, s->nloci, (double *)nullptr, 1, uacc, s->epoch, s->flag.p, s->counters.p, s->taus.p, s->sp, tau_q, win_u, mix_c, mix_lnc);
  HIPCHK(hipGetLastError());
  s->launches++;
  return 1;
}

static int gb_initialize(bpa_sampler * s) { return gb_step(s, 5) && gb_eval(s, 1); }

static int gb_iterate(bpa_sampler * s, unsigned iterations)
{
  bpa_engine * e = s->eng;
  s->host_current = false;
  for (unsigned it = 0; it < iterations; ++it)
  {
    const unsigned ngage = s->env_gage >= 0 ? (unsigned)s->env_gage : s->maxtips - 1, ngspr = s->env_gspr >= 0 ? (unsigned)s->env_gspr : 2*s->maxtips - 2;   // (BPA_SMP_STEPS: diagnostics)
    for (unsigned k = 0; k < ngage; ++k)   { if (!gb_step(s, 0, k) || !gb_eval(s, 0)) return 0; }
    for (unsigned k = 0; k < ngspr; ++k)   { if (!gb_step(s, 1, k) || !gb_eval(s, 0)) return 0; }
    s->sweeps++;
    if (s->env_nomix) continue;
    if (s->sp.theta_alpha > 0)
    {
      if (!gb_step(s, 4)) return 0;
      smp::ThetaArgs ta{};
      for (int p = 0; p < s->sp.npop; ++p)
        if (s->has_theta[p]) { ta.on[p] = 1u; ta.win_u[p] = a00_rndu(&s->grng); ta.uacc[p] = a00_rndu(&s->grng); }
      hipLaunchKernelGGL(smp::theta_sum_decide_kernel, dim3(s->sp.npop), dim3(1024), 0, e->stream, s->pop_nc.p, s->pop_t2h.p,
                         s->nloci, s->taus.p, s->sp, ta, s->counters.p, (double *)nullptr, (const double *)nullptr, 1);
      HIPCHK(hipGetLastError());
      s->logpr_stale = true;
      s->launches += 1;
    }
    for (int q = s->sp.S; q < s->sp.npop; ++q)
    {
      const double uprop = a00_rndu(&s->grng), uacc_t = a00_rndu(&s->grng);
      if (!gb_step(s, 2, (unsigned)q, uprop) || !gb_eval(s, 1) || !gb_decide(s, uacc_t, q, uprop, 1.0, 0.0)) return 0;
    }
    const double lnc = s->sp.ft_mix*(a00_rndu(&s->grng) - 0.5), c = std::exp(lnc);
    const double uacc = a00_rndu(&s->grng);
    if (!gb_step(s, 3, 0, 0.0, c, lnc) || !gb_eval(s, 1) || !gb_decide(s, uacc, -1, 0.0, c, lnc)) return 0;
  }
  return 1;
}

static int gb_download(bpa_sampler * s)
{
  bpa_engine * e = s->eng;
  if (!gb_step(s, 4)) return 0;
  HIPCHK(hipMemcpyAsync(s->b_trees.data(), s->b_dev.p, s->nloci*sizeof(gbig::BTree), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 1;
}
