// experimental/kernels_s20_pipe_exp.hpp — 20-state node-update kernels that are no default and no fall-back any more (round 6: out of
// the default build; compiled with -DBPA_EXPERIMENTAL, where BPA_S20_KERNEL=generic / pipe selects them and
// tests/test_gpu_parity.py::test_20_state_kernel_variants_are_bit_exact still holds them to the reference's bits):
//   partials_lnl_sN_kernel      one lane per pattern, P-matrices read from HBM per use (round 1's first 20-state form)
//   partials_lnl_pipe20_kernel  round 2-4's default: a workgroup barrier per update (superseded by partials_lnl_wave20_kernel)
// Included by kernels.hpp at the places they were cut from (they use the helpers defined above those places).
#pragma once
template <int S, bool NTA = false, int OCC = 3>      // NTA: CLV planes streamed (nontemporal); OCC: waves per SIMD of the register budget
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
partials_lnl_pipe20_kernel(const PlanDev P)
{
  extern __shared__ __attribute__((aligned(16))) double s_p[];      // [2 buffers][2 children][R][S][S], then [R][64] scratch
  constexpr uint32_t SS = S*S;
  // flags bit 5: plain mapping (A/B); bit 8: the launch covers tiles blk0 .. blk0 + gridDim.x (a half-batch of the device sampler)
  const uint32_t b = ((P.flags & 256u) ? P.blk0 : 0u) + ((P.flags & 32u) ? blockIdx.x : xcd_tile(blockIdx.x, gridDim.x)), lane = threadIdx.x & 63u, nw = blockDim.x >> 6;
  const uint32_t k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t t = ((cu32_p)P.tile_task)[b];
  const uint32_t n = ((cu32_p)P.tile_n0)[b] + lane;
  const uint32_t lid = ((cu32_p)P.task_locus)[t];
  cu64_p L64 = (cu64_p)(P.loci + lid);
  cu32_p L32 = (cu32_p)(P.loci + lid);
  const gdbl_p   Lclv    = (gdbl_p)L64[0];
  const double * Lpmat   = (const double *)L64[1];
  const gu32_p   Lscaler = (gu32_p)L64[2];
  const gcu32_p  Ltips   = (gcu32_p)L64[3];
  const gcu32_p  Lwgt    = (gcu32_p)L64[4];
  const cdbl4_p  par     = (cdbl4_p)L64[5];
  static_assert(offsetof(LocusDev, np) == 72 && offsetof(LocusDev, ld) == 108, "LocusDev layout");
  const uint32_t np = L32[18], tips_n = L32[19], R = L32[20], unphased = L32[25], ld = L32[27];
  const bool active = n < np && k < R;
  const uint32_t bufsz = 2*P.pad*SS;                                 // doubles per staging buffer (P.pad = largest R of the plan)
  double * s_x = s_p + (size_t)2*bufsz;

  // flags bit 6: op_off holds a (begin, end) pair per task — the device-written steps of the generic sampler, where a task
  // with no update is not part of the step at all
  const bool ranges = (P.flags & 64u) != 0;
  const uint32_t op_begin = ((cu32_p)P.op_off)[ranges ? 2*t : t], op_end = ((cu32_p)P.op_off)[ranges ? 2*t + 1 : t + 1];
  if (ranges && op_begin == op_end) return;
  uint32_t cur = 0;
  if (op_begin < op_end)
  {
    const OpS op0 = load_op_scalar(P.ops, op_begin);
    stage_pmats_async<S>(s_p, Lpmat + (size_t)op0.left_pmatrix*R*SS, Lpmat + (size_t)op0.right_pmatrix*R*SS, R, k, nw, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
  }
  double ov[S];                                                      // the parent just computed (forwarded)
  uint32_t ov_clv = 0xffffffffu;
  for (uint32_t o = op_begin; o < op_end; ++o)
  {
    const OpS op = load_op_scalar(P.ops, o);
    const double * lm = s_p + (size_t)cur*bufsz + (size_t)k*SS;
    const double * rm = s_p + (size_t)cur*bufsz + (size_t)(R + k)*SS;
    const bool ltip = op.left_clv < tips_n, rtip = op.right_clv < tips_n;
    const bool lfwd = op.left_clv == ov_clv, rfwd = op.right_clv == ov_clv;
    double lv[S], rv[S];
    uint32_t lcode = 1u, rcode = 1u;
    bool all_small = true;
    // every request of this update first: tip codes, child planes, the next update's matrices
    if (active)
    {
      if (ltip) lcode = Ltips[(size_t)op.left_clv*np + n];
      if (rtip) rcode = Ltips[(size_t)op.right_clv*np + n];
      if (!ltip && !lfwd)
      {
        const gcdbl_p p = Lclv + (((size_t)(op.left_clv - tips_n)*R + k)*S)*ld + n;
#pragma unroll
        for (int s = 0; s < S; ++s) lv[s] = NTA ? __builtin_nontemporal_load(p + (size_t)s*ld) : p[(size_t)s*ld];
      }
      if (!rtip && !rfwd)
      {
        const gcdbl_p p = Lclv + (((size_t)(op.right_clv - tips_n)*R + k)*S)*ld + n;
#pragma unroll
        for (int s = 0; s < S; ++s) rv[s] = NTA ? __builtin_nontemporal_load(p + (size_t)s*ld) : p[(size_t)s*ld];
      }
    }
    if (o + 1 < op_end)
    {
      const OpS nx = load_op_scalar(P.ops, o + 1);
      stage_pmats_async<S>(s_p + (size_t)(cur ^ 1u)*bufsz, Lpmat + (size_t)nx.left_pmatrix*R*SS, Lpmat + (size_t)nx.right_pmatrix*R*SS, R, k, nw, lane);
    }
    if (active)
    {
      // tip children: partials_lnl_tiledk_kernel's tip-code fast path (wave-uniform), else the 0/1 expansion of the code
      const bool lfast = ltip && __all(__popc(lcode) == 1), rfast = rtip && __all(__popc(rcode) == 1);
      const int ls = __ffs(lcode) - 1, rs = __ffs(rcode) - 1;
      if (ltip && !lfast) {
#pragma unroll
        for (int s = 0; s < S; ++s) lv[s] = (double)((lcode >> s) & 1u); }
      if (rtip && !rfast) {
#pragma unroll
        for (int s = 0; s < S; ++s) rv[s] = (double)((rcode >> s) & 1u); }
      if (lfwd) {
#pragma unroll
        for (int s = 0; s < S; ++s) lv[s] = ov[s]; }
      if (rfwd) {
#pragma unroll
        for (int s = 0; s < S; ++s) rv[s] = ov[s]; }
#pragma unroll
      for (int i = 0; i < S; ++i)
      {
        const double x = lfast ? lm[i*S + ls] : dot_fma4<S>(lm + i*S, lv);
        const double y = rfast ? rm[i*S + rs] : dot_fma4<S>(rm + i*S, rv);
        const double v = x*y;
        all_small = all_small && (v < BPA_SCALE_THRESHOLD);
        ov[i] = v;
      }
      ov_clv = op.parent_clv;
    }
    if (op.parent_scaler >= 0)                         // uniform: the scaling test couples the categories
    {
      reinterpret_cast<uint32_t *>(s_x)[k*64 + lane] = all_small ? 1u : 0u;
      lds_barrier();
      if (active)
      {
        bool all = true;
        for (uint32_t q = 0; q < R; ++q) all = all && reinterpret_cast<const uint32_t *>(s_x)[q*64 + lane] != 0u;
        if (all) {
#pragma unroll
          for (int i = 0; i < S; ++i) ov[i] *= BPA_SCALE_FACTOR; }
        if (k == 0)
        {
          uint32_t sc = all ? 1u : 0u;
          if (op.left_scaler  >= 0) sc += Lscaler[(size_t)op.left_scaler*np  + n];
          if (op.right_scaler >= 0) sc += Lscaler[(size_t)op.right_scaler*np + n];
          Lscaler[(size_t)op.parent_scaler*np + n] = sc;
        }
      }
    }
    if (active)
    {
      const gdbl_p out = Lclv + ((((size_t)(op.parent_clv - tips_n)*R) + k)*S)*ld + n;
#pragma unroll
      for (int i = 0; i < S; ++i) { if (NTA) __builtin_nontemporal_store(ov[i], out + (size_t)i*ld); else out[(size_t)i*ld] = ov[i]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next matrices has landed (stores drain with it)
    lds_barrier();                                     // everyone is done reading buffer `cur` and has filled the other
    cur ^= 1u;
  }
  if (!(P.flags & 4u)) return;

  // K2 / K3 (core_likelihood_avx2.c:45-87): every wave its category's term, wave 0 the fma chain over them
  const uint32_t root = ((cu32_p)P.root_clv)[t];
  if (active)
  {
    double c[S];
    if (root == ov_clv) {
#pragma unroll
      for (int s = 0; s < S; ++s) c[s] = ov[s]; }
    else if (root < tips_n)
    {
      const uint32_t code = Ltips[(size_t)root*np + n];
#pragma unroll
      for (int s = 0; s < S; ++s) c[s] = (double)((code >> s) & 1u);
    }
    else
    {
      const gcdbl_p p = Lclv + (((size_t)(root - tips_n)*R + k)*S)*ld + n;
#pragma unroll
      for (int s = 0; s < S; ++s) c[s] = p[(size_t)s*ld];
    }
    const uint32_t m = (uint32_t)par[par_param_idx(R) + k];
    s_x[k*64 + lane] = dot_fma4_s<S>(par + par_matrix(R, S, m) + pm_freqs(S), c);
  }
  lds_barrier();
  if (!active || k) return;
  double term = 0;
  for (uint32_t q = 0; q < R; ++q) term = __builtin_fma(s_x[q*64 + lane], par[par_rate_weights(R) + q], term);
  if (!unphased)
  {
    double lt = log(term);
    const int32_t rsc = ((ci32_p)P.root_scaler)[t];
    if (rsc >= 0)
    {
      const uint32_t sc = Lscaler[(size_t)rsc*np + n];
      if (sc) lt = __builtin_fma((double)sc, BPA_LOG_SCALE_THRESHOLD, lt);
    }
    term = lt*Lwgt[n];
  }
  P.site_term[((cu32_p)P.task_pat_off)[t] + n] = term;
}

