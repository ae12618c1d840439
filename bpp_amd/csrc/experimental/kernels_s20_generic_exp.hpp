// experimental/kernels_s20_generic_exp.hpp — 20-state node-update kernels that are no default and no fall-back any more (round 6: out of
// the default build; compiled with -DBPA_EXPERIMENTAL, where BPA_S20_KERNEL=generic / pipe selects them and
// tests/test_gpu_parity.py::test_20_state_kernel_variants_are_bit_exact still holds them to the reference's bits):
//   partials_lnl_sN_kernel      one lane per pattern, P-matrices read from HBM per use (round 1's first 20-state form)
//   partials_lnl_pipe20_kernel  round 2-4's default: a workgroup barrier per update (superseded by partials_lnl_wave20_kernel)
// Included by kernels.hpp at the places they were cut from (they use the helpers defined above those places).
#pragma once
template <int S>
__global__ void __launch_bounds__(BPA_BLOCK) partials_lnl_sN_kernel(const PlanDev P)
{
  const uint32_t g = blockIdx.x*BPA_BLOCK + threadIdx.x;
  if (g >= P.npatterns) return;
  const uint32_t t = P.thr_task[g];
  const uint32_t n = g - P.task_pat_off[t];
  const LocusDev L = P.loci[P.task_locus[t]];
  const uint32_t R = L.rate_cats, np = L.np, ld = L.ld;

  const uint32_t op_end = P.op_off[t+1];
  for (uint32_t o = P.op_off[t]; o < op_end; ++o)
  {
    const OpDev op = P.ops[o];
    double * out = L.clv + (((size_t)(op.parent_clv - L.tips_n)*R)*S)*ld + n;
    bool all_small = true;
    for (uint32_t k = 0; k < R; ++k)
    {
      double lv[S], rv[S];
      load_childN<S, uint32_t>(L, op.left_clv,  k, n, lv);
      load_childN<S, uint32_t>(L, op.right_clv, k, n, rv);
      const double * lm = L.pmat + ((size_t)op.left_pmatrix*R  + k)*S*S;
      const double * rm = L.pmat + ((size_t)op.right_pmatrix*R + k)*S*S;
      double * dst = out + (size_t)k*S*ld;
      for (int i = 0; i < S; ++i)
      {
        const double x = dot_fma4<S>(lm + i*S, lv);
        const double y = dot_fma4<S>(rm + i*S, rv);
        const double v = x*y;
        all_small = all_small && (v < BPA_SCALE_THRESHOLD);
        dst[(size_t)i*ld] = v;
      }
    }
    if (op.parent_scaler >= 0)
    {
      uint32_t s = 0;
      if (op.left_scaler  >= 0) s += L.scaler[(size_t)op.left_scaler*np  + n];
      if (op.right_scaler >= 0) s += L.scaler[(size_t)op.right_scaler*np + n];
      if (all_small)
      {
        for (uint32_t e = 0; e < R*S; ++e) out[(size_t)e*ld] *= BPA_SCALE_FACTOR;
        s += 1;
      }
      L.scaler[(size_t)op.parent_scaler*np + n] = s;
    }
  }

  // K2 / K3 (core_likelihood_avx2.c:45-87; that file is built with -mfma, so the
  // rate-weight accumulation and the scaler correction are fused there too)
  const uint32_t root = P.root_clv[t];
  const double * par = L.par;
  double term = 0;
  for (uint32_t k = 0; k < R; ++k)
  {
    double c[S];
    load_childN<S, uint32_t>(L, root, k, n, c);
    const uint32_t m = (uint32_t)par[par_param_idx(R) + k];
    const double tr = dot_fma4<S>(par + par_matrix(R, S, m) + pm_freqs(S), c);
    term = __builtin_fma(tr, par[par_rate_weights(R) + k], term);
  }
  if (L.unphased_length)
    P.site_term[g] = term;
  else
  {
    double lt = log(term);
    const int32_t rs = P.root_scaler[t];
    if (rs >= 0)
    {
      const uint32_t sc = L.scaler[(size_t)rs*np + n];
      if (sc) lt = __builtin_fma((double)sc, BPA_LOG_SCALE_THRESHOLD, lt);
    }
    lt *= L.weights[n];
    P.site_term[g] = lt;
  }
}

