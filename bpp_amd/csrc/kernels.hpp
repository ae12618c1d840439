// kernels.hpp — hand-written HIP kernels for gfx950 (MI355X, wave64).
//
// Hot path of BPP's per-locus likelihood:
//   K1  node update            pll_core_update_partial_ii   core_partials.c:585
//   K2  root log-likelihood    pll_core_root_loglikelihood  core_likelihood.c:24
//   K3  site-likelihood vector pll_core_root_likelihood_vector core_likelihood.c:214 (+ locus.c:2586-2615)
//   K4  JC69 P-matrix          locus_update_matrices_jc69   locus.c:2325
//   K5  eigen P-matrix         bpp_core_update_pmatrix      core_pmatrix.c:674 (and :785 library form)
//   K6  eigendecomposition     pll_update_eigen             core_pmatrix.c:239
//
// Execution model.  A *plan* is one proposal step for many loci.  Felsenstein
// pruning is independent per site pattern all the way to the root, so one lane
// owns one (locus, pattern) and walks that locus's whole list of node updates by
// itself: no cross-lane traffic, no barriers, children produced by the lane are
// re-read by the same lane.  Lanes of a wave are consecutive patterns, i.e.
// consecutive 32-B chunks of every CLV plane (coalesced 16-B/lane loads).  The
// per-pattern log-likelihood terms are then summed per locus in pattern order
// (the reference's order) by a second tiny kernel.
//
// Numerics: fp64, compiled with -ffp-contract=off; the only fused operations are
// explicit __builtin_fma calls that mirror the reference's AVX2+FMA 20-state
// back-end.  4-state dot products use the AVX order (p0+p1)+(p2+p3)
// (core_partials_avx.c:461-487); 20-state ones use four FMA lane accumulators
// then (a0+a1)+(a2+a3) (core_partials_avx2.c:666-745).  CLVs are therefore
// bit-identical to the reference's AVX2 build.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"

#define BPA_SCALE_FACTOR     0x1p+256
#define BPA_SCALE_THRESHOLD  0x1p-256
// log(2^-256) as glibc evaluates it (core_likelihood.c:201): -256*ln2, exactly representable scaling of RN(ln2)
#define BPA_LOG_SCALE_THRESHOLD (-0x1.62e42fefa39efp+7)

static constexpr int BPA_BLOCK = 256;

// profiling aid: drain outstanding memory ops, then stamp the 100-MHz wall clock
#define BPA_STAMP(P, b, lane, i) do { if ((P).dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if ((lane) == 0) (P).dbg[(size_t)(b)*8 + (i)] = wall_clock64(); } } while (0)

// the same without draining anything: what the wave has ISSUED by then (explicit waits before it still show)
#define BPA_STAMP_NW(P, b, lane, i) do { if ((P).dbg) { if ((lane) == 0) (P).dbg[(size_t)(b)*8 + (i)] = wall_clock64(); } } while (0)

// ------------------------------------------------------------------ helpers --
__device__ __forceinline__ double dot4_pair(const double m0, const double m1, const double m2,
                                            const double m3, const double * v)
{
  const double p0 = m0*v[0], p1 = m1*v[1], p2 = m2*v[2], p3 = m3*v[3];
  return (p0 + p1) + (p2 + p3);
}

template <int S>
__device__ __forceinline__ double dot_fma4(const double * __restrict__ row, const double * v)
{
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
  for (int j = 0; j < S; j += 4)
  {
    a0 = __builtin_fma(row[j+0], v[j+0], a0);
    a1 = __builtin_fma(row[j+1], v[j+1], a1);
    a2 = __builtin_fma(row[j+2], v[j+2], a2);
    a3 = __builtin_fma(row[j+3], v[j+3], a3);
  }
  return (a0 + a1) + (a2 + a3);
}

// 16-byte accesses in the GLOBAL address space: a pointer that comes out of a record is generic to the compiler, and a generic
// (flat) access also counts against lgkmcnt — every wait for LDS or scalar data then waits for the HBM accesses in flight
typedef double d2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 gld2(const double * p)
{
  const d2v_t v = *reinterpret_cast<const __attribute__((address_space(1))) d2v_t *>(reinterpret_cast<uintptr_t>(p));
  double2 r; r.x = v.x; r.y = v.y; return r;
}
__device__ __forceinline__ void gst2(double * p, const double2 v)
{
  d2v_t w; w.x = v.x; w.y = v.y;
  *reinterpret_cast<__attribute__((address_space(1))) d2v_t *>(reinterpret_cast<uintptr_t>(p)) = w;
}
typedef uint32_t u4v_t __attribute__((ext_vector_type(4)));
typedef uint32_t u2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 gld4(const uint4 * p)
{
  const u4v_t v = *reinterpret_cast<const __attribute__((address_space(1))) u4v_t *>(reinterpret_cast<uintptr_t>(p));
  uint4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r;
}
template <typename T> __device__ __forceinline__ T gld(const T * p)
{ return *reinterpret_cast<const __attribute__((address_space(1))) T *>(reinterpret_cast<uintptr_t>(p)); }
template <typename T> __device__ __forceinline__ void gst(T * p, const T v)
{ *reinterpret_cast<__attribute__((address_space(1))) T *>(reinterpret_cast<uintptr_t>(p)) = v; }

// XCD-aware workgroup -> tile mapping (MI355X: 8 XCDs, each with its own L2; workgroup b is dispatched to XCD b % 8):
// consecutive TILES go to the same XCD, so workgroups that share inputs (the tiles of one locus share its P-matrices)
// find them in that XCD's L2 instead of fetching them once per XCD
__device__ __forceinline__ uint32_t xcd_tile(const uint32_t b, const uint32_t nb)
{
  const uint32_t per = nb >> 3;
  return b < (per << 3) ? (b & 7u)*per + (b >> 3) : b;
}

// a workgroup barrier that orders LDS traffic only: outstanding global loads and stores are NOT waited for (a plain
// __syncthreads() drains vmcnt, i.e. exposes the latency of every store issued before it).  For hand-overs through LDS.
__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 16 bytes per lane, global -> LDS directly (global_load_lds_dwordx4: the wave's 64 lanes land at lds_base + 16 lane), as INLINE
// ASSEMBLY: behind __builtin_amdgcn_global_load_lds the compiler's wait-count pass cannot tell which LDS bytes the load writes
// and puts s_waitcnt vmcnt(0) in front of every later ds_read of the kernel — which then also waits for every global STORE issued
// in between (seen in the disassembly of both round-5 step kernels, tools/disasm.py).  The caller waits for these loads itself
// (s_waitcnt vmcnt); the compiler's own counts stay safe (it waits for a few more than it must: returns are in order).
__device__ __forceinline__ void lds_dma16(const void * gsrc, const void * lds_base)
{
  const uint32_t lds_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)lds_base;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :: "v"((const __attribute__((address_space(1))) void *)gsrc), "s"(lds_off) : "memory", "m0");
}

// ================================================================ K1+K2, S=4 ==
// one lane = one (task, pattern); CLV plane layout [buffer][rate][pattern][4]
__device__ __forceinline__ void load_child4(const LocusDev & L, uint32_t clv_index, uint32_t k,
                                            uint32_t n, double v[4])
{
  if (clv_index < L.tips_n)
  {
    const uint32_t code = L.tips[(size_t)clv_index*L.np + n];
    v[0] = (code & 1u) ? 1.0 : 0.0;
    v[1] = (code & 2u) ? 1.0 : 0.0;
    v[2] = (code & 4u) ? 1.0 : 0.0;
    v[3] = (code & 8u) ? 1.0 : 0.0;
  }
  else
  {
    const double2 * p = reinterpret_cast<const double2 *>(
        L.clv + (((size_t)(clv_index - L.tips_n)*L.rate_cats + k)*L.np + n)*4);
    const double2 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
  }
}

__device__ __forceinline__ void matvec4(const double * __restrict__ m, const double v[4], double x[4])
{
  const double2 * r = reinterpret_cast<const double2 *>(m);
#pragma unroll
  for (int i = 0; i < 4; ++i)
  {
    const double2 a = r[2*i], b = r[2*i+1];
    x[i] = dot4_pair(a.x, a.y, b.x, b.y, v);
  }
}

__device__ __forceinline__ void matvec4_ab(const double a, const double b, const double v[4], double x[4])
{
  // rows (a b b b), (b a b b), (b b a b), (b b b a) in the AVX order (p0+p1)+(p2+p3)
  x[0] = dot4_pair(a, b, b, b, v);
  x[1] = dot4_pair(b, a, b, b, v);
  x[2] = dot4_pair(b, b, a, b, v);
  x[3] = dot4_pair(b, b, b, a, v);
}

// P-matrix (matrix index p, rate k) times v; JC69 loci store the (a, b) pair only
__device__ __forceinline__ void matvec4_p(const double * __restrict__ pmat, const uint32_t pstride,
                                          const size_t pk, const double v[4], double x[4])
{
  if (pstride == 2)
  {
    const double2 ab = *reinterpret_cast<const double2 *>(pmat + pk*2);
    matvec4_ab(ab.x, ab.y, v, x);
  }
  else matvec4(pmat + pk*16, v, x);
}

// node updates of task t for pattern n, then the pattern's root term: returns
// w[n]*(log(site lh) + scaler*log(2^-256))  — or the bare site likelihood for diploid loci
__device__ __forceinline__ double walk_s4(const PlanDev & P, const uint32_t t, const uint32_t n,
                                          const LocusDev & L, const bool do_ops)
{
  const uint32_t R = L.rate_cats, np = L.np;
  if (do_ops)
  {
    const bool ranges = (P.flags & 64u) != 0;        // a (begin, end) pair per task: the device-written steps of the big-tree sampler
    const uint32_t op_end = P.op_off[ranges ? 2*t + 1 : t + 1];
    for (uint32_t o = P.op_off[ranges ? 2*t : t]; o < op_end; ++o)
    {
      const OpDev op = P.ops[o];
      double * out = L.clv + (((size_t)(op.parent_clv - L.tips_n)*R)*np + n)*4;
      bool all_small = true;
      for (uint32_t k = 0; k < R; ++k)
      {
        double lv[4], rv[4], x[4], y[4];
        load_child4(L, op.left_clv,  k, n, lv);
        load_child4(L, op.right_clv, k, n, rv);
        matvec4_p(L.pmat, L.pstride, (size_t)op.left_pmatrix*R  + k, lv, x);
        matvec4_p(L.pmat, L.pstride, (size_t)op.right_pmatrix*R + k, rv, y);
        double2 o0, o1;
        o0.x = x[0]*y[0]; o0.y = x[1]*y[1]; o1.x = x[2]*y[2]; o1.y = x[3]*y[3];
        all_small = all_small && (o0.x < BPA_SCALE_THRESHOLD) && (o0.y < BPA_SCALE_THRESHOLD)
                              && (o1.x < BPA_SCALE_THRESHOLD) && (o1.y < BPA_SCALE_THRESHOLD);
        double2 * dst = reinterpret_cast<double2 *>(out + (size_t)k*np*4);
        dst[0] = o0; dst[1] = o1;
      }
      if (op.parent_scaler >= 0)
      {
        // fill_parent_scaler (core_partials.c:24-46) + per-pattern scaling (core_partials_avx.c:516-529)
        uint32_t s = 0;
        if (op.left_scaler  >= 0) s += L.scaler[(size_t)op.left_scaler*np  + n];
        if (op.right_scaler >= 0) s += L.scaler[(size_t)op.right_scaler*np + n];
        if (all_small)
        {
          for (uint32_t k = 0; k < R; ++k)
          {
            double2 * dst = reinterpret_cast<double2 *>(out + (size_t)k*np*4);
            double2 a = dst[0], b = dst[1];
            a.x *= BPA_SCALE_FACTOR; a.y *= BPA_SCALE_FACTOR; b.x *= BPA_SCALE_FACTOR; b.y *= BPA_SCALE_FACTOR;
            dst[0] = a; dst[1] = b;
          }
          s += 1;
        }
        L.scaler[(size_t)op.parent_scaler*np + n] = s;
      }
    }
  }

  // K2 / K3 at the root (core_likelihood_avx.c:117-150, 251-275)
  const uint32_t root = P.root_clv[t];
  const double * par = L.par;
  double term = 0;
  for (uint32_t k = 0; k < R; ++k)
  {
    double c[4];
    load_child4(L, root, k, n, c);
    const uint32_t m = (uint32_t)par[par_param_idx(R) + k];
    const double * f = par + par_matrix(R, 4, m) + pm_freqs(4);
    const double tr = dot4_pair(f[0], f[1], f[2], f[3], c);
    term += tr*par[par_rate_weights(R) + k];
  }
  if (L.unphased_length) return term;            // K3: likelihood, scalers ignored
  double lt = log(term);
  const int32_t rs = P.root_scaler[t];
  if (rs >= 0)
  {
    const uint32_t sc = L.scaler[(size_t)rs*np + n];
    if (sc) lt += sc*BPA_LOG_SCALE_THRESHOLD;
  }
  return lt*L.weights[n];
}

__global__ void __launch_bounds__(BPA_BLOCK) partials_lnl_s4_kernel(const PlanDev P)
{
  const uint32_t g = blockIdx.x*BPA_BLOCK + threadIdx.x;
  if (g >= P.npatterns) return;
  const uint32_t t = P.thr_task[g];
  const uint32_t n = g - P.task_pat_off[t];
  const LocusDev L = P.loci[P.task_locus[t]];
  P.site_term[g] = walk_s4(P, t, n, L, true);
}

// ============================================================ K1+K2, generic S ==
// state-major planes: clv[((buffer*R + k)*S + s)*Np + n]; one lane = one pattern.
template <int S, typename CODE, bool NTL = false>
__device__ __forceinline__ void load_childN(const LocusDev & L, uint32_t clv_index, uint32_t k,
                                            uint32_t n, double * v)
{
  if (clv_index < L.tips_n)
  {
    const uint32_t code = reinterpret_cast<const CODE *>(L.tips)[(size_t)clv_index*L.np + n];
#pragma unroll
    for (int s = 0; s < S; ++s) v[s] = ((code >> s) & 1u) ? 1.0 : 0.0;
  }
  else
  {
    const double * p = L.clv + (((size_t)(clv_index - L.tips_n)*L.rate_cats + k)*S)*L.ld + n;
#pragma unroll
    for (int s = 0; s < S; ++s) v[s] = NTL ? __builtin_nontemporal_load(p + (size_t)s*L.ld) : p[(size_t)s*L.ld];
  }
}

#ifdef BPA_EXPERIMENTAL
#include "experimental/kernels_s20_generic_exp.hpp"   // partials_lnl_sN_kernel (one lane per pattern, no staging)
#endif

// ============================================== K1+K2, 20 states, LDS-staged P ==
// One workgroup = one tile of TILE consecutive patterns of ONE locus; one lane = one
// pattern.  For every node update the two children's P-matrices (R x 20 x 20 each) are
// staged once per workgroup into LDS with coalesced loads and then read by all lanes at
// the same address (LDS broadcast, conflict-free), so the inner loop is
// v_fma_f64(P from LDS, child CLV in registers): FP64 VALU at LDS-broadcast rate, no
// per-lane P traffic to L1/L2.  Child CLVs come from state-major planes (lanes =
// consecutive patterns: coalesced 8 B/lane).  Summation order = the reference's AVX2
// back-end (four FMA lane accumulators, core_partials_avx2.c:666-745): bit-identical CLVs.
template <int S, int TILE>
__global__ void __launch_bounds__(TILE) partials_lnl_tiled_kernel(const PlanDev P)
{
  extern __shared__ __attribute__((aligned(16))) double s_p[];      // [2][R][S][S]
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  const uint32_t t = P.tile_task[b];
  const uint32_t n = P.tile_n0[b] + lane;
  const LocusDev L = P.loci[P.task_locus[t]];
  const uint32_t R = L.rate_cats, np = L.np, ld = L.ld;
  const bool active = n < np;
  constexpr uint32_t SS = S*S;

  const uint32_t op_end = P.op_off[t+1];
  for (uint32_t o = P.op_off[t]; o < op_end; ++o)
  {
    const OpDev op = P.ops[o];
    __syncthreads();                                   // previous update's LDS reads are done
    {
      const double2 * gl = reinterpret_cast<const double2 *>(L.pmat + (size_t)op.left_pmatrix*R*SS);
      const double2 * gr = reinterpret_cast<const double2 *>(L.pmat + (size_t)op.right_pmatrix*R*SS);
      double2 * sl = reinterpret_cast<double2 *>(s_p);
      double2 * sr = reinterpret_cast<double2 *>(s_p + (size_t)R*SS);
      for (uint32_t i = lane; i < R*SS/2; i += TILE) { sl[i] = gl[i]; sr[i] = gr[i]; }
    }
    __syncthreads();
    if (active)
    {
      double * out = L.clv + (((size_t)(op.parent_clv - L.tips_n)*R)*S)*ld + n;
      bool all_small = true;
      for (uint32_t k = 0; k < R; ++k)
      {
        double lv[S], rv[S];
        load_childN<S, uint32_t>(L, op.left_clv,  k, n, lv);
        load_childN<S, uint32_t>(L, op.right_clv, k, n, rv);
        const double * lm = s_p + (size_t)k*SS;
        const double * rm = s_p + (size_t)(R + k)*SS;
        double * dst = out + (size_t)k*S*ld;
#pragma unroll 4
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
  }
  if (!active || !(P.flags & 4u)) return;

  // K2 / K3 (core_likelihood_avx2.c:45-87)
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
  if (!L.unphased_length)
  {
    double lt = log(term);
    const int32_t rs = P.root_scaler[t];
    if (rs >= 0)
    {
      const uint32_t sc = L.scaler[(size_t)rs*np + n];
      if (sc) lt = __builtin_fma((double)sc, BPA_LOG_SCALE_THRESHOLD, lt);
    }
    term = lt*L.weights[n];
  }
  P.site_term[P.task_pat_off[t] + n] = term;
}

// ================================== K1+K2, 20 states, pipelined (default since round 2) ==
// partials_lnl_tiledk_kernel's arithmetic and workgroup shape (64 patterns x R categories, wave k owns plane k), with
// the serial chain of an update cut down — what the round-1 kernel waited for was not bandwidth (a kernel of this
// access shape with no arithmetic streams the same bytes in 230-300 us, tools/probe_bw.hip) but round trips in series:
//   * everything wave-uniform — tile, locus record, update records — comes through the scalar data path (s_load into
//     SGPRs; was: one vector load per update, waited for with vmcnt(0), which also drained the previous update's stores);
//   * the two children's P-matrices go global -> LDS directly (global_load_lds_dwordx4: no registers, no ds_write),
//     DOUBLE-BUFFERED: update o+1's matrices are requested together with update o's child CLVs (was: four dependent
//     load -> wait -> ds_write rounds between two barriers);
//   * ONE workgroup barrier per update and it orders LDS only (s_waitcnt lgkmcnt(0); s_barrier): CLV stores are never
//     waited for — a lane only ever re-reads what it wrote itself (same pattern, same category);
//   * a parent that is a child of the next update (or the root) is forwarded in registers instead of re-read.
typedef const uint32_t __attribute__((address_space(4))) * cu32_p;
typedef const int32_t  __attribute__((address_space(4))) * ci32_p;
typedef const uint64_t __attribute__((address_space(4))) * cu64_p;
struct OpS { uint32_t parent_clv; int32_t parent_scaler; uint32_t left_clv, left_pmatrix; int32_t left_scaler; uint32_t right_clv, right_pmatrix; int32_t right_scaler; };
__device__ __forceinline__ OpS load_op_scalar(const OpDev * ops, const uint32_t o)
{
  cu32_p q = (cu32_p)(ops + o);
  OpS r;
  r.parent_clv = q[0]; r.parent_scaler = (int32_t)q[1]; r.left_clv = q[2]; r.left_pmatrix = q[3]; r.left_scaler = (int32_t)q[4];
  r.right_clv = q[5]; r.right_pmatrix = q[6]; r.right_scaler = (int32_t)q[7];
  return r;
}
// the 2 x R x S x S doubles of an update's two P-matrix sets, global -> LDS, 1 KB (64 lanes x 16 B) per instruction
template <int S>
__device__ __forceinline__ void stage_pmats_async(double * s_dst, const double * gl, const double * gr, const uint32_t R,
                                                  const uint32_t w, const uint32_t nw, const uint32_t lane)
{
  const uint32_t half = R*S*S/2, total = 2*half;                 // 16-byte units
  for (uint32_t c = w; c*64 < total; c += nw)
  {
    const uint32_t idx = c*64 + lane;
    if (idx < total)
    {
      const double * src = idx < half ? gl + 2*(size_t)idx : gr + 2*(size_t)(idx - half);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(s_dst + (size_t)c*128), 16, 0, 0);
    }
  }
}

typedef double   __attribute__((address_space(1))) * gdbl_p;
typedef const double   __attribute__((address_space(1))) * gcdbl_p;
typedef uint32_t __attribute__((address_space(1))) * gu32_p;
typedef const uint32_t __attribute__((address_space(1))) * gcu32_p;
typedef const double __attribute__((address_space(4))) * cdbl4_p;
template <int S>
__device__ __forceinline__ double dot_fma4_s(cdbl4_p row, const double * v)     // dot_fma4 with the row in SGPRs
{
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
  for (int j = 0; j < S; j += 4)
  {
    a0 = __builtin_fma(row[j+0], v[j+0], a0);
    a1 = __builtin_fma(row[j+1], v[j+1], a1);
    a2 = __builtin_fma(row[j+2], v[j+2], a2);
    a3 = __builtin_fma(row[j+3], v[j+3], a3);
  }
  return (a0 + a1) + (a2 + a3);
}

#ifdef BPA_EXPERIMENTAL
#include "experimental/kernels_s20_pipe_exp.hpp"      // partials_lnl_pipe20_kernel (rounds 2-4's default)
#endif

// ================================== K1+K2, 20 states, waves on their own (default since round 5) ==
// partials_lnl_pipe20_kernel with what still ran in series taken apart.  In that kernel a tile's four waves (one per rate
// category) met at a workgroup barrier every update because the P-matrix staging was dealt out over all of them, every
// update began by requesting its child planes and waiting for them, and ended by waiting for vmcnt(0) — i.e. for its own
// 20 plane stores — before the barrier: load latency, arithmetic and store drain one after the other, on all four SIMDs at
// once, at two waves per SIMD (rocprofv3, round 4: 53 % of the wave-cycles waiting, VALU busy 16 %, LDS 40 %, 0.39 of the
// HBM peak by bytes moved).  Here:
//   * a wave stages ITS category's two matrices itself (7 global_load_lds_dwordx4 per update into its own LDS corner,
//     double-buffered): wave k only ever reads category k, so NO workgroup barrier is left in the update loop — the eight
//     waves of a CU drift apart and one's memory phase lies beside another's arithmetic;
//   * the inputs of update o + 1 — the tip codes and the one inner plane that is neither forwarded nor written earlier in
//     this step — are requested BEFORE update o's arithmetic, into registers of their own (the first update's: in the
//     prologue, together with its matrices; its second inner plane lands in the registers of the not yet computed parent);
//   * the wait at the top of an update is s_waitcnt vmcnt(20): everything older than the previous update's 20 stores —
//     matrices and prefetch — has landed; stores are never waited for (a lane re-reads only what it wrote itself).
// Arithmetic, summation order, scaling rule and stores are pipe20's: same bits (tests/test_gpu_parity.py).
template <int S>
__device__ __forceinline__ void stage_pmats_wave(double * s_dst, const double * gl, const double * gr, const uint32_t lane)
{
  constexpr uint32_t half = S*S/2, total = 2*half;               // 16-byte units of one category's two matrices
#pragma unroll
  for (uint32_t c = 0; c*64 < total; ++c)
  {
    const uint32_t idx = c*64 + lane;
    if (idx < total)
    {
      const double * src = idx < half ? gl + 2*(size_t)idx : gr + 2*(size_t)(idx - half);
      // as inline assembly, NOT __builtin_amdgcn_global_load_lds: the compiler's wait-count pass cannot tell which LDS bytes a
      // global -> LDS load writes and puts s_waitcnt vmcnt(0) in front of EVERY later ds_read — this update's matrix reads would
      // wait for the coming update's matrices, its prefetched plane and the previous update's stores (seen in the disassembly:
      // tools/disasm.py).  The waits for these loads are the kernel's own (vmcnt(20) at the top of an update); the pass still
      // counts its own loads correctly (it then waits for a few more than it must, never fewer: the returns are in order).
      const uint32_t lds_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(s_dst + (size_t)c*128);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                   :: "v"((const __attribute__((address_space(1))) void *)src), "s"(lds_off) : "memory", "m0");
    }
  }
}

// PP patterns per lane (tile = 64 PP patterns): with PP = 2 every matrix element read from LDS feeds two fused multiply-adds per
// lane instead of one.  The kernel's roof is neither HBM nor FP64 but the LDS pipe: a broadcast ds_read_b128 occupies it for 4
// cycles and brings the 2 matrix entries of 2 FMAs per lane — 25 M such reads per config-4 launch = 390 k LDS cycles per CU,
// ~190 us of the launch's 320 (SQ_LDS_IDX_ACTIVE agrees), against ~100 us of FP64 issue.  Two patterns per lane at one wave per
// SIMD (the registers of two: 512 per lane) halve the reads per pattern; the wave's latency cover is the requests-ahead above.
template <bool COHERENT = false> __device__ __forceinline__ void lnl_reduce_wave(const PlanDev & P, const uint32_t t, const uint32_t lane);
template <int S, bool NTA = false, int OCC = 2, int PP = 1, bool RL = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
partials_lnl_wave20_kernel(const PlanDev P)
{
  extern __shared__ __attribute__((aligned(16))) double s_p[];      // [R waves][2 buffers][2 children][S][S], then [PP][R][64] scratch
  constexpr uint32_t SS = S*S;
  static_assert(S == 20 && (PP == 1 || PP == 2), "vmcnt immediates below are written for 20 PP plane stores per update");
  const uint32_t b = ((P.flags & 256u) ? P.blk0 : 0u) + ((P.flags & 32u) ? blockIdx.x : xcd_tile(blockIdx.x, gridDim.x)), lane = threadIdx.x & 63u;
  const uint32_t k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t t = ((cu32_p)P.tile_task)[b];
  const uint32_t n_tile = ((cu32_p)P.tile_n0)[b] + lane;
  const uint32_t lid = ((cu32_p)P.task_locus)[t];
  cu64_p L64 = (cu64_p)(P.loci + lid);
  cu32_p L32 = (cu32_p)(P.loci + lid);
  const gdbl_p   Lclv    = (gdbl_p)L64[0];
  const double * Lpmat   = (const double *)L64[1];
  const gu32_p   Lscaler = (gu32_p)L64[2];
  const gcu32_p  Ltips   = (gcu32_p)L64[3];
  const gcu32_p  Lwgt    = (gcu32_p)L64[4];
  const cdbl4_p  par     = (cdbl4_p)L64[5];
  const uint32_t np = L32[18], tips_n = L32[19], R = L32[20], unphased = L32[25], ld = L32[27];
  const bool wave_on = k < R;                                        // (wave-uniform)
  // The lanes past the locus's last pattern run the LAST pattern again (same loads, same arithmetic, the same values stored to
  // the same addresses) instead of being masked off: no exec-masked region is left in the update loop, so every path through
  // it issues the same vector-memory instructions and the wait counts are the same on all of them (the compiler's wait-count
  // pass merges paths; a path that could skip the stores would turn the vmcnt(20) at the top into vmcnt(0)).
  bool real[PP]; uint32_t n[PP];
#pragma unroll
  for (int q = 0; q < PP; ++q) { const uint32_t nn = n_tile + 64u*q; real[q] = nn < np && wave_on; n[q] = nn < np ? nn : np - 1u; }
  const bool active = wave_on;
  double * myp = s_p + (size_t)k*4*SS;                               // this wave's [2][2][SS]
  double * s_x = s_p + (size_t)4*P.pad*SS;                           // [PP][R][64]

  const bool ranges = (P.flags & 64u) != 0;
  const uint32_t op_begin = ((cu32_p)P.op_off)[ranges ? 2*t : t], op_end = ((cu32_p)P.op_off)[ranges ? 2*t + 1 : t + 1];
  if (ranges && op_begin == op_end) return;
  static const uint32_t none = 0xffffffffu;
  // (round 6) flags bit 11: the step's last update is not stored when it makes the root's planes (step_s4_klane_v3_kernel has the
  // reasons): nothing but the root term below reads them, and it has them in registers — 20 R plane stores per pattern less,
  // a third to a half of what a per-locus step writes (rocprofv3, round 5: 59 % of this kernel's HBM bytes were writes)
  const uint32_t skip_root = (P.flags & 2048u) ? ((cu32_p)P.root_clv)[t] : none;
  double ov[PP][S];                                                  // the parent just computed (forwarded)
  uint32_t ov_clv = none;
  double pfv[PP][S];                                                 // the plane requested ahead for the coming update
  uint32_t pf_clv = none, pf2_clv = none;                            // pf2: the first update's second inner plane, parked in ov
  uint32_t pf_lcode[PP], pf_rcode[PP];
#pragma unroll
  for (int q = 0; q < PP; ++q) { pf_lcode[q] = 1u; pf_rcode[q] = 1u; }
  uint32_t written = 0;                                              // CLV buffers this step has written (5 bits; aliases are taken as written: a late read)
  auto plane = [&](uint32_t c, int q) { return Lclv + (((size_t)(c - tips_n)*R + k)*S)*ld + n[q]; };
  // the requests for update `nx` whose predecessor writes `prevp` (forwarded): codes, one inner plane
  auto request = [&](const OpS & nx, const uint32_t prevp, const bool first)
  {
    const bool lt = nx.left_clv < tips_n, rt = nx.right_clv < tips_n;
#pragma unroll
    for (int q = 0; q < PP; ++q)
    {
      pf_lcode[q] = (active && lt) ? Ltips[(size_t)nx.left_clv*np + n[q]] : 1u;
      pf_rcode[q] = (active && rt) ? Ltips[(size_t)nx.right_clv*np + n[q]] : 1u;
    }
    const bool pl = !lt && nx.left_clv != prevp && !((written >> (nx.left_clv & 31u)) & 1u);
    const bool pr = !rt && nx.right_clv != prevp && !((written >> (nx.right_clv & 31u)) & 1u);
    pf_clv = pl ? nx.left_clv : pr ? nx.right_clv : none;
    pf2_clv = (first && pl && pr) ? nx.right_clv : none;
    if (active && pf_clv != none)
    {
#pragma unroll
      for (int q = 0; q < PP; ++q)
      {
        const gcdbl_p p = plane(pf_clv, q);
#pragma unroll
        for (int s = 0; s < S; ++s) pfv[q][s] = NTA ? __builtin_nontemporal_load(p + (size_t)s*ld) : p[(size_t)s*ld];
      }
    }
    if (active && pf2_clv != none)
    {
#pragma unroll
      for (int q = 0; q < PP; ++q)
      {
        const gcdbl_p p = plane(pf2_clv, q);
#pragma unroll
        for (int s = 0; s < S; ++s) ov[q][s] = NTA ? __builtin_nontemporal_load(p + (size_t)s*ld) : p[(size_t)s*ld];
      }
    }
  };
  uint32_t cur = 0;
  if (op_begin < op_end && wave_on)
  {
    const OpS op0 = load_op_scalar(P.ops, op_begin);
    stage_pmats_wave<S>(myp, Lpmat + ((size_t)op0.left_pmatrix*R + k)*SS, Lpmat + ((size_t)op0.right_pmatrix*R + k)*SS, lane);
    request(op0, none, true);
    // the first update's inputs are waited for HERE, visibly to the compiler (the empty statements below read-modify every
    // requested register): the loop header then merges "nothing pending" with the back edge's "stores behind the requests"
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < PP; ++q)
    {
      asm volatile("" : "+v"(pf_lcode[q]), "+v"(pf_rcode[q]));
#pragma unroll
      for (int s = 0; s < S; ++s) { asm volatile("" : "+v"(pfv[q][s])); asm volatile("" : "+v"(ov[q][s])); }
    }
  }
  bool drain = false;                                                // the wait at the top of the coming update is for everything (after a scaling update)
  for (uint32_t o = op_begin; o < op_end; ++o)
  {
    const OpS op = load_op_scalar(P.ops, o);
    bool all_small[PP];
#pragma unroll
    for (int q = 0; q < PP; ++q) all_small[q] = true;
    if (wave_on)
    {
      if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (PP == 1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");    // all but the previous update's plane stores
      else              asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
      drain = false;
      // (an opaque read-modify of every prefetched register: without it the compiler rotates the loop and copies the requested
      //  plane into the coming update's operand registers right behind the requests — with an s_waitcnt vmcnt(0) in front)
#pragma unroll
      for (int q = 0; q < PP; ++q)
      {
        asm volatile("" : "+v"(pf_lcode[q]), "+v"(pf_rcode[q]));
#pragma unroll
        for (int s = 0; s < S; ++s) asm volatile("" : "+v"(pfv[q][s]));
      }
      const double * lm = myp + (size_t)cur*2*SS;
      const double * rm = lm + SS;
      const bool ltip = op.left_clv < tips_n, rtip = op.right_clv < tips_n;
      const bool lfwd = op.left_clv == ov_clv, rfwd = op.right_clv == ov_clv;
      const bool lpf = !ltip && !lfwd && op.left_clv == pf_clv, rpf = !rtip && !rfwd && !lpf && op.right_clv == pf_clv;
      const bool lp2 = !ltip && !lfwd && !lpf && op.left_clv == pf2_clv, rp2 = !rtip && !rfwd && !rpf && op.right_clv == pf2_clv;
      double lv[PP][S], rv[PP][S];
      // everything requested ahead is taken over HERE, before the coming update's requests reuse the registers
      bool lfast = ltip, rfast = rtip;
      int ls[PP], rs[PP];
#pragma unroll
      for (int q = 0; q < PP; ++q)
      {
        const uint32_t lcode = pf_lcode[q], rcode = pf_rcode[q];
        lfast = lfast && __all(__popc(lcode) == 1); rfast = rfast && __all(__popc(rcode) == 1);
        ls[q] = __ffs(lcode) - 1; rs[q] = __ffs(rcode) - 1;
      }
#pragma unroll
      for (int q = 0; q < PP; ++q)
      {
        const uint32_t lcode = pf_lcode[q], rcode = pf_rcode[q];
        if (ltip && !lfast) {
#pragma unroll
          for (int s = 0; s < S; ++s) lv[q][s] = (double)((lcode >> s) & 1u); }
        if (rtip && !rfast) {
#pragma unroll
          for (int s = 0; s < S; ++s) rv[q][s] = (double)((rcode >> s) & 1u); }
        // what was not requested ahead (a plane an earlier update of this step wrote; a later update's second inner plane)
        if (!ltip && !lfwd && !lpf && !lp2)
        {
          const gcdbl_p p = plane(op.left_clv, q);
#pragma unroll
          for (int s = 0; s < S; ++s) lv[q][s] = NTA ? __builtin_nontemporal_load(p + (size_t)s*ld) : p[(size_t)s*ld];
#pragma unroll
          for (int s = 0; s < S; ++s) asm volatile("" : "+v"(lv[q][s]));      // (waited for inside the branch, every one of them: see the prologue)
        }
        if (!rtip && !rfwd && !rpf && !rp2)
        {
          const gcdbl_p p = plane(op.right_clv, q);
#pragma unroll
          for (int s = 0; s < S; ++s) rv[q][s] = NTA ? __builtin_nontemporal_load(p + (size_t)s*ld) : p[(size_t)s*ld];
#pragma unroll
          for (int s = 0; s < S; ++s) asm volatile("" : "+v"(rv[q][s]));
        }
        if (lfwd || lp2) {
#pragma unroll
          for (int s = 0; s < S; ++s) lv[q][s] = ov[q][s]; }
        if (rfwd || rp2) {
#pragma unroll
          for (int s = 0; s < S; ++s) rv[q][s] = ov[q][s]; }
        if (lpf) {
#pragma unroll
          for (int s = 0; s < S; ++s) lv[q][s] = pfv[q][s]; }
        if (rpf) {
#pragma unroll
          for (int s = 0; s < S; ++s) rv[q][s] = pfv[q][s]; }
      }
      // ---- the coming update's matrices and inputs, before this one's arithmetic
      written |= 1u << (op.parent_clv & 31u);
      if (o + 1 < op_end)
      {
        const OpS nx = load_op_scalar(P.ops, o + 1);
        stage_pmats_wave<S>(myp + (size_t)(cur ^ 1u)*2*SS, Lpmat + ((size_t)nx.left_pmatrix*R + k)*SS, Lpmat + ((size_t)nx.right_pmatrix*R + k)*SS, lane);
        request(nx, op.parent_clv, false);
      }
      else { pf_clv = none; pf2_clv = none; }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < S; ++i)
      {
        // the row's entries come out of LDS once and serve the lane's PP patterns (same 4-accumulator fma order per pattern)
        double x[PP], y[PP];
        if (lfast) {
#pragma unroll
          for (int q = 0; q < PP; ++q) x[q] = lm[i*S + ls[q]]; }
        else
        {
          double a[PP][4];
#pragma unroll
          for (int q = 0; q < PP; ++q) a[q][0] = a[q][1] = a[q][2] = a[q][3] = 0;
          // RL: the row comes out of LDS with ONE read (lane l brings entry l % S) and goes to the multiply-adds as scalar
          // operands through v_readlane — 40 LDS reads per update instead of 400 broadcast reads of two entries each, whose
          // latency (three in flight is what the registers allow) is what the wave waits for
          const double prow = RL ? lm[i*S + (int)(lane % S)] : 0.0;
          const int plo = __double2loint(prow), phi = __double2hiint(prow);
#pragma unroll
          for (int j = 0; j < S; j += 4)
          {
            double m0, m1, m2, m3;
            if (RL)
            {
              m0 = __hiloint2double(__builtin_amdgcn_readlane(phi, j), __builtin_amdgcn_readlane(plo, j));
              m1 = __hiloint2double(__builtin_amdgcn_readlane(phi, j + 1), __builtin_amdgcn_readlane(plo, j + 1));
              m2 = __hiloint2double(__builtin_amdgcn_readlane(phi, j + 2), __builtin_amdgcn_readlane(plo, j + 2));
              m3 = __hiloint2double(__builtin_amdgcn_readlane(phi, j + 3), __builtin_amdgcn_readlane(plo, j + 3));
            }
            else { m0 = lm[i*S + j]; m1 = lm[i*S + j + 1]; m2 = lm[i*S + j + 2]; m3 = lm[i*S + j + 3]; }
#pragma unroll
            for (int q = 0; q < PP; ++q)
            {
              a[q][0] = __builtin_fma(m0, lv[q][j+0], a[q][0]); a[q][1] = __builtin_fma(m1, lv[q][j+1], a[q][1]);
              a[q][2] = __builtin_fma(m2, lv[q][j+2], a[q][2]); a[q][3] = __builtin_fma(m3, lv[q][j+3], a[q][3]);
            }
          }
#pragma unroll
          for (int q = 0; q < PP; ++q) x[q] = (a[q][0] + a[q][1]) + (a[q][2] + a[q][3]);
        }
        if (rfast) {
#pragma unroll
          for (int q = 0; q < PP; ++q) y[q] = rm[i*S + rs[q]]; }
        else
        {
          double a[PP][4];
#pragma unroll
          for (int q = 0; q < PP; ++q) a[q][0] = a[q][1] = a[q][2] = a[q][3] = 0;
          const double prow = RL ? rm[i*S + (int)(lane % S)] : 0.0;
          const int plo = __double2loint(prow), phi = __double2hiint(prow);
#pragma unroll
          for (int j = 0; j < S; j += 4)
          {
            double m0, m1, m2, m3;
            if (RL)
            {
              m0 = __hiloint2double(__builtin_amdgcn_readlane(phi, j), __builtin_amdgcn_readlane(plo, j));
              m1 = __hiloint2double(__builtin_amdgcn_readlane(phi, j + 1), __builtin_amdgcn_readlane(plo, j + 1));
              m2 = __hiloint2double(__builtin_amdgcn_readlane(phi, j + 2), __builtin_amdgcn_readlane(plo, j + 2));
              m3 = __hiloint2double(__builtin_amdgcn_readlane(phi, j + 3), __builtin_amdgcn_readlane(plo, j + 3));
            }
            else { m0 = rm[i*S + j]; m1 = rm[i*S + j + 1]; m2 = rm[i*S + j + 2]; m3 = rm[i*S + j + 3]; }
#pragma unroll
            for (int q = 0; q < PP; ++q)
            {
              a[q][0] = __builtin_fma(m0, rv[q][j+0], a[q][0]); a[q][1] = __builtin_fma(m1, rv[q][j+1], a[q][1]);
              a[q][2] = __builtin_fma(m2, rv[q][j+2], a[q][2]); a[q][3] = __builtin_fma(m3, rv[q][j+3], a[q][3]);
            }
          }
#pragma unroll
          for (int q = 0; q < PP; ++q) y[q] = (a[q][0] + a[q][1]) + (a[q][2] + a[q][3]);
        }
#pragma unroll
        for (int q = 0; q < PP; ++q)
        {
          const double v = x[q]*y[q];
          all_small[q] = all_small[q] && (v < BPA_SCALE_THRESHOLD);
          ov[q][i] = v;
        }
      }
      ov_clv = op.parent_clv;
    }
    if (op.parent_scaler >= 0)                         // uniform: the scaling test couples the categories
    {
#pragma unroll
      for (int q = 0; q < PP; ++q) reinterpret_cast<uint32_t *>(s_x)[(q*P.pad + k)*64 + lane] = all_small[q] ? 1u : 0u;
      lds_barrier();
      if (active)
      {
#pragma unroll
        for (int q = 0; q < PP; ++q)
        {
          bool all = true;
          for (uint32_t c = 0; c < R; ++c) all = all && reinterpret_cast<const uint32_t *>(s_x)[(q*P.pad + c)*64 + lane] != 0u;
          if (all) {
#pragma unroll
            for (int i = 0; i < S; ++i) ov[q][i] *= BPA_SCALE_FACTOR; }
          if (k == 0)
          {
            uint32_t sc = all ? 1u : 0u;
            if (op.left_scaler  >= 0) sc += Lscaler[(size_t)op.left_scaler*np  + n[q]];
            if (op.right_scaler >= 0) sc += Lscaler[(size_t)op.right_scaler*np + n[q]];
            Lscaler[(size_t)op.parent_scaler*np + n[q]] = sc;
          }
        }
      }
      lds_barrier();                                   // (the scratch is free again)
      drain = true;                                    // (more than the plane stores were issued behind the requests)
    }
    // (the skipped store LEAVES the loop: every path around the back edge issues the same stores, so the compiler's wait
    //  counts at the top of an update stay what the explicit vmcnt(20) assumes — see the note on masked regions above)
    if (o + 1u == op_end && op.parent_clv == skip_root) break;
    if (wave_on)
    {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < PP; ++q)
      {
        const gdbl_p out = Lclv + ((((size_t)(op.parent_clv - tips_n)*R) + k)*S)*ld + n[q];
#pragma unroll
        for (int i = 0; i < S; ++i) { if (NTA) __builtin_nontemporal_store(ov[q][i], out + (size_t)i*ld); else out[(size_t)i*ld] = ov[q][i]; }
      }
      asm volatile("" ::: "memory");
    }
    cur ^= 1u;
  }
  if (!(P.flags & 4u)) return;

  // K2 / K3 (core_likelihood_avx2.c:45-87): every wave its category's term, wave 0 the fma chain over them
  const uint32_t root = ((cu32_p)P.root_clv)[t];
  if (active)
  {
    if (root != ov_clv && root >= tips_n) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (a root this step wrote before its last update: its stores first)
#pragma unroll
    for (int q = 0; q < PP; ++q)
    {
      double c[S];
      if (root == ov_clv) {
#pragma unroll
        for (int s = 0; s < S; ++s) c[s] = ov[q][s]; }
      else if (root < tips_n)
      {
        const uint32_t code = Ltips[(size_t)root*np + n[q]];
#pragma unroll
        for (int s = 0; s < S; ++s) c[s] = (double)((code >> s) & 1u);
      }
      else
      {
        const gcdbl_p p = plane(root, q);
#pragma unroll
        for (int s = 0; s < S; ++s) c[s] = p[(size_t)s*ld];
      }
      const uint32_t m = (uint32_t)par[par_param_idx(R) + k];
      s_x[(q*P.pad + k)*64 + lane] = dot_fma4_s<S>(par + par_matrix(R, S, m) + pm_freqs(S), c);
    }
  }
  lds_barrier();
  if (k) return;
#pragma unroll
  for (int q = 0; q < PP; ++q)
  {
    if (!real[q]) continue;
    double term = 0;
    for (uint32_t c = 0; c < R; ++c) term = __builtin_fma(s_x[(q*P.pad + c)*64 + lane], par[par_rate_weights(R) + c], term);
    if (!unphased)
    {
      double lt = log(term);
      const int32_t rsc = ((ci32_p)P.root_scaler)[t];
      if (rsc >= 0)
      {
        const uint32_t sc = Lscaler[(size_t)rsc*np + n[q]];
        if (sc) lt = __builtin_fma((double)sc, BPA_LOG_SCALE_THRESHOLD, lt);
      }
      term = lt*Lwgt[n[q]];
    }
    double * dst = P.site_term + ((cu32_p)P.task_pat_off)[t] + n[q];      // (real lanes: n = the lane's own pattern)
    if (P.flags & 512u) __hip_atomic_store(dst, term, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *dst = term;
  }
  if (P.flags & 512u)
  {
    // the per-locus sum without a launch of its own: the tile that arrives LAST adds the locus's terms up in pattern order
    // (the reference's sequential sum).  Hand-over by 8-byte agent-scope atomics on both sides (MI355X_MICROARCH.md, inter-
    // workgroup visibility: write-through stores, loads past the reader's L1) — NOT a release fence, which writes back
    // every dirty line of the XCD's L2, i.e. the CLV planes this kernel has just stored: measured 72 against 131 it/s —, the
    // wave's stores drained (vmcnt(0)) before its returning arrival atomic.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t old = 0;
    if (lane == 0) old = atomicAdd(P.tile_arrive + t, 1u);
    old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
    const uint32_t ntile = (np + 64u*PP - 1u)/(64u*PP);
    if ((old + 1u) % ntile == 0u) lnl_reduce_wave<true>(P, t, lane);
  }
}

// ======================= K1+K2, 20 states, FP64 MFMA (north_star's matrix-core path), one wave per rate category ==
// The contraction parent[i][n] = (sum_j Pl[i][j] L[j][n]) (sum_j Pr[i][j] R[j][n]) on the matrix cores with
// v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction: blocks = four groups of 4 patterns, A = a 4-row x
// 4-column piece of P replicated over the blocks, B = 4 states x 16 patterns; 20 states = 5 row tiles, no padding in M), ON
// the memory pipeline of partials_lnl_pipe20_kernel: tile / locus / update records through the scalar data path, the two
// children's P-matrices global -> LDS direct and double-buffered, one LDS-only barrier per update, the XCD-aware tile map.
// Lane layout (tools/mfma_layout.hip):
//   A[blk][i][kk] @ lane kk*16+blk*4+i   B[blk][kk][j] @ lane kk*16+blk*4+j   D[blk][i][j] @ lane i*16+blk*4+j
// Bit-exactness: the instruction accumulates kk = 0..3 as an ascending fma chain seeded with C (tools/mfma_order.hip),
// which IS the reference's AVX2 lane accumulator over its first four column blocks: accumulator a takes columns a, a+4,
// a+8, a+12 in one MFMA; its fifth term (column a+16) is one v_fma_f64 in the D layout; then (acc0+acc1)+(acc2+acc3) and
// the product (core_partials_avx2.c:666-745).  The CLV operands of all four 16-pattern groups of the tile are requested
// first (64 loads in flight per lane), then each row tile's A operands come out of LDS ONCE (8 ds_read_b128) and serve
// the four groups: P costs 40 LDS reads per update and wave instead of pipe20's 400.  No register forwarding (D and B
// layouts differ): the next update re-reads the parent through L2.
// NOT the default: on gfx950 the FP64 matrix rate is the FP64 vector rate (tools/f64_rate.hip, measured on the box:
// v_fma_f64 58.5, v_mfma_f64_4x4x4 72.0, v_mfma_f64_16x16x4 49.1 TFLOP/s), so the matrix cores can only save operand
// traffic, and the layout costs more than that saves: config 4, per launch, 499-528 us against pipe20's 353-366 us
// (profiles/r3: SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_LDS 3.3 M against 18.7 M).  An earlier form without pipe20's pipeline
// (partials_lnl_mfma20k_kernel, rounds 1-2) took 575 us.
template <bool NTA = false, int OCC = 2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
partials_lnl_pipemfma20_kernel(const PlanDev P)
{
  constexpr int S = 20, NG = 4;
  constexpr uint32_t SS = 400;
  extern __shared__ __attribute__((aligned(16))) double s_p[];      // [2 buffers][2 children][R][S][S], then [R][64] scratch
  const uint32_t b = (P.flags & 32u) ? blockIdx.x : xcd_tile(blockIdx.x, gridDim.x), lane = threadIdx.x & 63u, nw = blockDim.x >> 6;
  const uint32_t k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t t = ((cu32_p)P.tile_task)[b];
  const uint32_t n0 = ((cu32_p)P.tile_n0)[b];
  const uint32_t lid = ((cu32_p)P.task_locus)[t];
  cu64_p L64 = (cu64_p)(P.loci + lid);
  cu32_p L32 = (cu32_p)(P.loci + lid);
  const gdbl_p   Lclv    = (gdbl_p)L64[0];
  const double * Lpmat   = (const double *)L64[1];
  const gu32_p   Lscaler = (gu32_p)L64[2];
  const gcu32_p  Ltips   = (gcu32_p)L64[3];
  const gcu32_p  Lwgt    = (gcu32_p)L64[4];
  const cdbl4_p  par     = (cdbl4_p)L64[5];
  const uint32_t np = L32[18], tips_n = L32[19], R = L32[20], unphased = L32[25], ld = L32[27];
  const uint32_t kq = lane >> 4, pl = lane & 15u, ri = lane & 3u;
  const bool wave_on = k < R;
  const uint32_t bufsz = 2*P.pad*SS;
  double * s_x = s_p + (size_t)2*bufsz;

  const uint32_t op_begin = ((cu32_p)P.op_off)[t], op_end = ((cu32_p)P.op_off)[t+1];
  uint32_t cur = 0;
  if (op_begin < op_end)
  {
    const OpS op0 = load_op_scalar(P.ops, op_begin);
    stage_pmats_async<S>(s_p, Lpmat + (size_t)op0.left_pmatrix*R*SS, Lpmat + (size_t)op0.right_pmatrix*R*SS, R, k, nw, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
  }
  uint32_t pat[NG]; bool pon[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) { const uint32_t q = n0 + 16u*g + pl; pon[g] = q < np; pat[g] = pon[g] ? q : np - 1; }
  for (uint32_t o = op_begin; o < op_end; ++o)
  {
    const OpS op = load_op_scalar(P.ops, o);
    const double * lm = s_p + (size_t)cur*bufsz + (size_t)k*SS;
    const double * rm = s_p + (size_t)cur*bufsz + (size_t)(R + k)*SS;
    const bool ltip = op.left_clv < tips_n, rtip = op.right_clv < tips_n;
    double bl[NG][4], blt[NG][4], br[NG][4], brt[NG][4];
    // every request of this update first: the B operands of the four pattern groups, then the next update's matrices
    if (wave_on)
    {
      if (ltip)
      {
#pragma unroll
        for (int g = 0; g < NG; ++g)
        {
          const uint32_t code = Ltips[(size_t)op.left_clv*np + pat[g]];
#pragma unroll
          for (int a = 0; a < 4; ++a) { bl[g][a] = ((code >> (a + 4*kq)) & 1u) ? 1.0 : 0.0; blt[g][a] = ((code >> (16 + a)) & 1u) ? 1.0 : 0.0; }
        }
      }
      else
      {
        const gcdbl_p q = Lclv + (((size_t)(op.left_clv - tips_n)*R + k)*S)*ld;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int a = 0; a < 4; ++a)
          {
            const gcdbl_p q0 = q + (size_t)(a + 4*kq)*ld + pat[g], q1 = q + (size_t)(16 + a)*ld + pat[g];
            bl[g][a]  = NTA ? __builtin_nontemporal_load(q0) : *q0;
            blt[g][a] = NTA ? __builtin_nontemporal_load(q1) : *q1;
          }
      }
      if (rtip)
      {
#pragma unroll
        for (int g = 0; g < NG; ++g)
        {
          const uint32_t code = Ltips[(size_t)op.right_clv*np + pat[g]];
#pragma unroll
          for (int a = 0; a < 4; ++a) { br[g][a] = ((code >> (a + 4*kq)) & 1u) ? 1.0 : 0.0; brt[g][a] = ((code >> (16 + a)) & 1u) ? 1.0 : 0.0; }
        }
      }
      else
      {
        const gcdbl_p q = Lclv + (((size_t)(op.right_clv - tips_n)*R + k)*S)*ld;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int a = 0; a < 4; ++a)
          {
            const gcdbl_p q0 = q + (size_t)(a + 4*kq)*ld + pat[g], q1 = q + (size_t)(16 + a)*ld + pat[g];
            br[g][a]  = NTA ? __builtin_nontemporal_load(q0) : *q0;
            brt[g][a] = NTA ? __builtin_nontemporal_load(q1) : *q1;
          }
      }
    }
    if (o + 1 < op_end)
    {
      const OpS nx = load_op_scalar(P.ops, o + 1);
      stage_pmats_async<S>(s_p + (size_t)(cur ^ 1u)*bufsz, Lpmat + (size_t)nx.left_pmatrix*R*SS, Lpmat + (size_t)nx.right_pmatrix*R*SS, R, k, nw, lane);
    }
    const gdbl_p out = Lclv + ((((size_t)(op.parent_clv - tips_n)*R) + k)*S)*ld;
    uint32_t small = 0xfu;                             // bit g: everything this lane produced for group g is < 2^-256
    if (wave_on)
    {
#pragma unroll
      for (int r = 0; r < 5; ++r)
      {
        // A operands of this row tile, once for the four groups: the MFMA piece P[4r+ri][a+4kq] and the tail column P[4r+kq][16+a]
        const double2 * pl0 = reinterpret_cast<const double2 *>(lm + (4*r + ri)*S + 4*kq);
        const double2 * pr0 = reinterpret_cast<const double2 *>(rm + (4*r + ri)*S + 4*kq);
        const double2 * pl1 = reinterpret_cast<const double2 *>(lm + (4*r + kq)*S + 16);
        const double2 * pr1 = reinterpret_cast<const double2 *>(rm + (4*r + kq)*S + 16);
        const double2 l01 = pl0[0], l23 = pl0[1], r01 = pr0[0], r23 = pr0[1];
        const double2 lt01 = pl1[0], lt23 = pl1[1], rt01 = pr1[0], rt23 = pr1[1];
        const double al[4] = {l01.x, l01.y, l23.x, l23.y}, ar[4] = {r01.x, r01.y, r23.x, r23.y};
        const double alt[4] = {lt01.x, lt01.y, lt23.x, lt23.y}, art[4] = {rt01.x, rt01.y, rt23.x, rt23.y};
#pragma unroll
        for (int g = 0; g < NG; ++g)
        {
          double xa[4], ya[4];
#pragma unroll
          for (int a = 0; a < 4; ++a)
          {
            xa[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(al[a], bl[g][a], 0.0, 0, 0, 0);
            ya[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar[a], br[g][a], 0.0, 0, 0, 0);
          }
#pragma unroll
          for (int a = 0; a < 4; ++a)
          {
            xa[a] = __builtin_fma(alt[a], blt[g][a], xa[a]);
            ya[a] = __builtin_fma(art[a], brt[g][a], ya[a]);
          }
          const double x = (xa[0] + xa[1]) + (xa[2] + xa[3]);
          const double y = (ya[0] + ya[1]) + (ya[2] + ya[3]);
          const double v = x*y;
          if (!(v < BPA_SCALE_THRESHOLD)) small &= ~(1u << g);
          if (pon[g]) { const gdbl_p w = out + (size_t)(4*r + kq)*ld + pat[g]; if (NTA) __builtin_nontemporal_store(v, w); else *w = v; }     // D: row = lane >> 4
        }
      }
    }
    if (op.parent_scaler >= 0)                         // uniform: the scaling test couples rows, lanes and categories
    {
      small &= __shfl_xor(small, 16);
      small &= __shfl_xor(small, 32);                  // the 4 lanes of a pattern hold its 20 rows
      if (kq == 0) reinterpret_cast<uint32_t *>(s_x)[k*16 + pl] = wave_on ? small : 0xfu;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the rescaling below re-reads this lane's own stores)
      lds_barrier();
      if (wave_on)
      {
        uint32_t all = 0xfu;
        for (uint32_t q = 0; q < R; ++q) all &= reinterpret_cast<const uint32_t *>(s_x)[q*16 + pl];
#pragma unroll
        for (int g = 0; g < NG; ++g)
        {
          if (!pon[g]) continue;
          const bool rescale = (all >> g) & 1u;
          if (rescale)
#pragma unroll
            for (int r = 0; r < 5; ++r) out[(size_t)(4*r + kq)*ld + pat[g]] *= BPA_SCALE_FACTOR;
          if (k == 0 && kq == 0)
          {
            uint32_t sc = rescale ? 1u : 0u;
            if (op.left_scaler  >= 0) sc += Lscaler[(size_t)op.left_scaler*np  + pat[g]];
            if (op.right_scaler >= 0) sc += Lscaler[(size_t)op.right_scaler*np + pat[g]];
            Lscaler[(size_t)op.parent_scaler*np + pat[g]] = sc;
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next matrices have landed; the next update reads this one's rows through other lanes
    lds_barrier();                                     // everyone is done reading buffer `cur` (and the scratch) and has filled the other
    cur ^= 1u;
  }
  if (!(P.flags & 4u)) return;

  // K2 / K3 (core_likelihood_avx2.c:45-87): every wave its category's term, wave 0 the fma chain over them
  const uint32_t root = ((cu32_p)P.root_clv)[t];
  const uint32_t n = n0 + lane;
  const bool active = wave_on && n < np;
  if (active)
  {
    double c[S];
    if (root < tips_n)
    {
      const uint32_t code = Ltips[(size_t)root*np + n];
#pragma unroll
      for (int s = 0; s < S; ++s) c[s] = (double)((code >> s) & 1u);
    }
    else
    {
      const gcdbl_p q = Lclv + (((size_t)(root - tips_n)*R + k)*S)*ld + n;
#pragma unroll
      for (int s = 0; s < S; ++s) c[s] = q[(size_t)s*ld];
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

// ================================================== batched host -> device set-up ==
// flush_state's uploads (tip codes, weights, parameter blocks of every locus that changed) as ONE staged copy + this scatter:
// record r says where payload bytes [src_off, src_off + bytes) go.  (10 000 loci = 40 000 separate copies otherwise.)
struct UpRec { unsigned char * dst; uint64_t src_off; uint32_t bytes, pad; };
__global__ void __launch_bounds__(64) scatter_upload_kernel(const UpRec * __restrict__ recs, const unsigned char * __restrict__ payload)
{
  const UpRec r = recs[blockIdx.x];
  const unsigned char * src = payload + r.src_off;
  if (((reinterpret_cast<uintptr_t>(r.dst) | reinterpret_cast<uintptr_t>(src)) & 3u) == 0)
  {
    const uint32_t nw = r.bytes >> 2;
    for (uint32_t q = threadIdx.x; q < nw; q += 64) reinterpret_cast<uint32_t *>(r.dst)[q] = reinterpret_cast<const uint32_t *>(src)[q];
    for (uint32_t b = (nw << 2) + threadIdx.x; b < r.bytes; b += 64) r.dst[b] = src[b];
  }
  else
    for (uint32_t b = threadIdx.x; b < r.bytes; b += 64) r.dst[b] = src[b];
}

// ====================================================== per-locus lnL reduction ==
// Sum of the per-pattern terms in pattern order (core_likelihood.c:206-210); the
// diploid branch averages the phase resolutions first (locus.c:2600-2614).
template <typename TERMPTR>
__device__ __forceinline__ double reduce_locus(const LocusDev & L, TERMPTR term)
{
  double logl = 0;
  if (L.unphased_length)
  {
    uint32_t k = 0;
    for (uint32_t u = 0; u < L.unphased_length; ++u)
    {
      double m = 0;
      const uint32_t c = L.dip_count[u];
      for (uint32_t r = 0; r < c; ++r) m += term[L.dip_map[k++]];
      m /= (double)c;
      logl += log(m)*L.dip_weights[u];
    }
  }
  else
  {
    const uint32_t np = L.np;
    for (uint32_t n = 0; n < np; ++n) logl += term[n];
  }
  return logl;
}

// one wave per locus: the terms are fetched 64 at a time (coalesced) and added in pattern order —
// the reference's sequential sum (core_likelihood.c:85) — by broadcasting them lane by lane
// (COHERENT: the terms were written by other workgroups of this launch — 8-byte agent-scope atomic loads, past this CU's L1)
template <bool COHERENT>
__device__ __forceinline__ void lnl_reduce_wave(const PlanDev & P, const uint32_t t, const uint32_t lane)
{
  const LocusDev & L = P.loci[P.task_locus[t]];
  const double * term = P.site_term + P.task_pat_off[t];
  if (L.unphased_length)
  {
    if (lane == 0) P.lnl[t] = P.bfbeta*reduce_locus(L, term);
    return;
  }
  const uint32_t np = L.np;
  double logl = 0;
  for (uint32_t base = 0; base < np; base += 64)
  {
    const double v = base + lane >= np ? 0.0 : COHERENT ? __hip_atomic_load(term + base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : term[base + lane];
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const uint32_t cnt = np - base < 64u ? np - base : 64u;
    for (uint32_t q = 0; q < cnt; ++q)
      logl += __hiloint2double(__builtin_amdgcn_readlane(hi, (int)q), __builtin_amdgcn_readlane(lo, (int)q));
  }
  if (lane == 0) P.lnl[t] = P.bfbeta*logl;
}
__global__ void __launch_bounds__(64) lnl_reduce_wave_kernel(const PlanDev P)
{
  lnl_reduce_wave(P, blockIdx.x + ((P.flags & 256u) ? P.blk0 : 0u), threadIdx.x);       // (bit 8: a half-batch's loci)
}


// sum over the tasks of a plan, deterministic (fixed strided order + LDS tree): the
// quantity threads.c:544-559 / 583-591 reduces over workers for the all-loci
// proposals (TAU: logl_diff, MIX: lnacceptance) and that is all-reduced across GPUs.
__global__ void __launch_bounds__(1024) lnl_sum_kernel(const double * __restrict__ lnl, uint32_t n,
                                                       double * __restrict__ out)
{
  __shared__ double sh[1024];
  double acc = 0;
  for (uint32_t i = threadIdx.x; i < n; i += 1024) acc += lnl[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t w = 512; w > 0; w >>= 1)
  {
    if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

// ================================================================== K4 / K5 ==
// P(t) for one (branch, rate category).  4-state: one lane does the 16 entries.
__device__ __forceinline__ void pmatrix_identity(double * p, int S)
{
  for (int j = 0; j < S; ++j)
    for (int c = 0; c < S; ++c) p[j*S + c] = (j == c) ? 1.0 : 0.0;
}

// library_form: expm1(lambda*rate*t), identity iff t == 0 (core_pmatrix.c:826,842)
// inference   : expm1(lambda*(t*rate)), identity iff t*rate < 1e-100 (core_pmatrix.c:731,738,754)
template <int S>
__device__ __forceinline__ void pmatrix_eigen_row(double * __restrict__ prow, int j, double t, double rate,
                                                  const double * __restrict__ evals,
                                                  const double * __restrict__ evecs,
                                                  const double * __restrict__ ievecs, bool library_form)
{
  double tmp[S];
  const double bl = t*rate;
#pragma unroll
  for (int m = 0; m < S; ++m)
  {
    const double e = library_form ? expm1(evals[m]*rate*t) : expm1(evals[m]*bl);
    tmp[m] = ievecs[j*S + m]*e;
  }
  for (int c = 0; c < S; ++c)
  {
    double acc = (j == c) ? 1.0 : 0.0;
#pragma unroll
    for (int m = 0; m < S; ++m) acc += tmp[m]*evecs[m*S + c];
    prow[c] = acc;
  }
}

// all four rows of a 4-state P(t), inference form: the four expm1 are taken once (not once per row); same products,
// same order of additions as pmatrix_eigen_row<4>
__device__ __forceinline__ void pmatrix_eigen_4x4(double * __restrict__ q, double t, double rate,
                                                  const double * __restrict__ evals,
                                                  const double * __restrict__ evecs,
                                                  const double * __restrict__ ievecs)
{
  const double bl = t*rate;
  double ex[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) ex[m] = expm1(evals[m]*bl);
#pragma unroll
  for (int j = 0; j < 4; ++j)
  {
    double tmp[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) tmp[m] = ievecs[j*4 + m]*ex[m];
#pragma unroll
    for (int c = 0; c < 4; ++c)
    {
      double acc = (j == c) ? 1.0 : 0.0;
#pragma unroll
      for (int m = 0; m < 4; ++m) acc += tmp[m]*evecs[m*4 + c];
      q[4*j + c] = acc;
    }
  }
}

// K7: closed-form P(t) of K80 / F81 / HKY / T92 / TN93 / F84 (locus.c:1981-2323), state
// order A,C,G,T, written P = I + (...)expm1(...) like the reference and evaluated in its order.
// model = BPA_DNA_MODEL_* (1..6); f = frequencies, q = substitution parameters of the locus.
__device__ __forceinline__ void pmatrix_dna_closed(const uint32_t model, const double * __restrict__ f,
                                                   const double * __restrict__ q, const double bl, double * p)
{
  if (model == 1)
  {
    const double kappa = q[1]/q[0];
    const double e1 = expm1(-4*bl/(kappa + 2));
    if (fabs(kappa - 1) < 1e-20)
    {
#pragma unroll
      for (int i = 0; i < 16; ++i) p[i] = ((i >> 2) == (i & 3)) ? 1. + 3/4.*e1 : -e1/4;
    }
    else
    {
      const double e2 = expm1(-2*bl*(kappa + 1)/(kappa + 2));
#pragma unroll
      for (int i = 0; i < 16; ++i)
        p[i] = ((i >> 2) == (i & 3)) ? 1 + (e1 + 2*e2)/4 : ((((i >> 2) ^ (i & 3)) == 2) ? (e1 - 2*e2)/4 : -e1/4);
    }
  }
  else if (model == 2)
  {
    double beta = 1;
    for (int j = 0; j < 4; ++j) beta -= f[j]*f[j];
    beta = 1./beta;
    const double e = exp(-beta*bl), em1 = expm1(-beta*bl);
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = ((i >> 2) == (i & 3)) ? e - f[i & 3]*em1 : -f[i & 3]*em1;
  }
  else if (model == 4)
  {
    const double GC = f[3] + f[2];
    const double e1 = expm1(-bl);
    const double e2 = expm1(-(q[0]/q[1] + 1)*bl/2);
    const double a = -(1 - GC)/2*e1, g = -GC/2*e1;
    p[0]  = a;                           p[1]  = GC/2*e1 - GC*e2;           p[2]  = g;                 p[3]  = 1 + 0.5*(1 - GC)*e1 + GC*e2;
    p[4]  = a;                           p[5]  = 1 + GC/2*e1 + (1 - GC)*e2; p[6]  = g;                 p[7]  = (1 - GC)/2*e1 - (1 - GC)*e2;
    p[8]  = 1 + 0.5*(1 - GC)*e1 + GC*e2; p[9]  = g;                         p[10] = GC/2*e1 - GC*e2;   p[11] = a;
    p[12] = (1 - GC)/2*e1 - (1 - GC)*e2; p[13] = g;                         p[14] = 1 + GC/2*e1 + (1 - GC)*e2; p[15] = a;
  }
  else
  {
    const double A = f[0], C = f[1], G = f[2], T = f[3], Y = T + C, R = A + G;
    double bt, a1t, a2t;
    if (model == 3)
    {
      const double kappa = q[1]/q[0];
      const double mr = 1/(2*T*C*kappa + 2*A*G*kappa + 2*Y*R);
      bt = bl*mr; a1t = a2t = kappa*bt;
    }
    else if (model == 6)
    {
      const double kappa = q[0]/q[1];
      const double mr = 1/(2*T*C*kappa + 2*A*G*kappa + 2*Y*R);
      bt = bl*mr; a1t = (1 + kappa/Y)*bt; a2t = (1 + kappa/R)*bt;
    }
    else
    {
      const double mr = 1/(2*T*C*q[0] + 2*A*G*q[1] + 2*Y*R);
      bt = bl*mr; a1t = (q[0]/q[2])*bt; a2t = (q[1]/q[2])*bt;
    }
    const double e1 = expm1(-bt), e2 = expm1(-(R*a2t + Y*bt)), e3 = expm1(-(Y*a1t + R*bt));
    p[0]  = 1 + Y*A/R*e1 + G/R*e2;  p[1]  = -C*e1;                 p[2]  = Y*G/R*e1 - G/R*e2;     p[3]  = -T*e1;
    p[4]  = -A*e1;                  p[5]  = 1 + (R*C*e1 + T*e3)/Y; p[6]  = -G*e1;                 p[7]  = (R*e1 - e3)*T/Y;
    p[8]  = Y*A/R*e1 - A/R*e2;      p[9]  = -C*e1;                 p[10] = 1 + Y*G/R*e1 + A/R*e2; p[11] = -T*e1;
    p[12] = -A*e1;                  p[13] = (R*e1 - e3)*C/Y;       p[14] = -G*e1;                 p[15] = 1 + (R*T*e1 + C*e3)/Y;
  }
}

__device__ __forceinline__ void pmatrix_s4_entry(const PlanDev & P, const uint32_t e, const uint32_t k)
{
  if (P.mat_task[e] == 0xffffffffu) return;          // a hole of a device-written step (bigsampler.hpp)
  const LocusDev & L = P.loci[P.task_locus[P.mat_task[e]]];
  const uint32_t R = L.rate_cats;
  if (k >= R) return;
  const double * par = L.par;
  const double t = P.mat_length[e];
  const double rate = par[par_rates(R) + k];
  double * p = L.pmat + ((size_t)P.mat_pmatrix[e]*R + k)*L.pstride;
  double q[16];
  if (L.model == 0 /* JC69, locus.c:2342-2414: stored as the pair (a, b) */)
  {
    const double bl = t*rate;
    double2 ab; ab.x = 1.0; ab.y = 0.0;
    if (!(bl < 1e-100))
    {
      ab.x = (1 + 3*exp(-4*bl/3))/4;
      ab.y = (1 - ab.x)/3;
    }
    *reinterpret_cast<double2 *>(p) = ab;
    return;
  }
  else if (L.model >= 1 && L.model <= 6)
  {
    const uint32_t m = (uint32_t)par[par_param_idx(R) + k];
    const double * pm = par + par_matrix(R, 4, m);
    pmatrix_dna_closed(L.model, pm + pm_freqs(4), pm + pm_subst(4), t*rate, q);
  }
  else
  {
    const double bl = t*rate;
    if (bl < 1e-100)
      pmatrix_identity(q, 4);
    else
    {
      const uint32_t m = (uint32_t)par[par_param_idx(R) + k];
      const double * pm = par + par_matrix(R, 4, m);
      pmatrix_eigen_4x4(q, t, rate, pm + pm_evals(4), pm + pm_evecs(4), pm + pm_ievecs(4));
    }
  }
  double2 * dst = reinterpret_cast<double2 *>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) { double2 v; v.x = q[2*i]; v.y = q[2*i+1]; dst[i] = v; }
}

__global__ void __launch_bounds__(BPA_BLOCK) pmatrix_s4_kernel(const PlanDev P, const uint32_t rmax)
{
  const uint32_t tid = blockIdx.x*BPA_BLOCK + threadIdx.x;
  const uint32_t e = tid / rmax, k = tid % rmax;
  if (e >= P.nmat) return;
  pmatrix_s4_entry(P, e, k);
}

// ==================================================== fused proposal step, S=4 ==
// One launch per proposal step: a workgroup owns whole loci.
//   phase A  lanes produce the step's P-matrices (K4/K5) into the loci's HBM buffers;
//            workgroup barrier (stores are write-through; readers are on the same CU);
//   phase B  one lane per site pattern walks the node updates and forms its root term
//            (K1+K2); a parent that is the next update's child is forwarded in registers;
//   phase C  one lane per locus adds the terms in pattern order (the reference's order,
//            core_likelihood.c:206-210) from LDS.
// Latency is the enemy here (config 2: 5 patterns x 2 updates per locus), so the
// descriptors are flattened (TaskRec/MatRec) and every load that does not depend on
// phase A is issued before it.
// one (branch, rate category) P-matrix from the category's rate and the parameter block of its matrix (device_types.hpp:
// freqs | substitution parameters | eigenvalues | eigenvectors | inverse eigenvectors; not read for JC69)
__device__ __forceinline__ void pmatrix_s4_core(double * dst_base, const uint32_t model, const double rate, const double * __restrict__ pm,
                                                const double t, const uint32_t k)
{
  double q[16];
  const double bl = t*rate;
  if (model == 0 /* JC69, locus.c:2342-2414: stored as the pair (a, b) */)
  {
    double2 ab; ab.x = 1.0; ab.y = 0.0;
    if (!(bl < 1e-100))
    {
      ab.x = (1 + 3*exp(-4*bl/3))/4;
      ab.y = (1 - ab.x)/3;
    }
    d2v_t abv; abv.x = ab.x; abv.y = ab.y;
    *reinterpret_cast<__attribute__((address_space(1))) d2v_t *>(reinterpret_cast<uintptr_t>(dst_base + (size_t)k*2)) = abv;
    return;
  }
  else if (model >= 1 && model <= 6)
    pmatrix_dna_closed(model, pm + pm_freqs(4), pm + pm_subst(4), bl, q);
  else if (bl < 1e-100)
    pmatrix_identity(q, 4);
  else
    pmatrix_eigen_4x4(q, t, rate, pm + pm_evals(4), pm + pm_evecs(4), pm + pm_ievecs(4));
  // (the P-matrix buffers are device memory: a global store, not a flat one that the wave's next LDS wait would wait for)
  auto * dst = reinterpret_cast<__attribute__((address_space(1))) d2v_t *>(reinterpret_cast<uintptr_t>(dst_base + (size_t)k*16));
#pragma unroll
  for (int i = 0; i < 8; ++i) { d2v_t v; v.x = q[2*i]; v.y = q[2*i+1]; dst[i] = v; }
}

__device__ __forceinline__ void pmatrix_s4_rec_t(const MatRec & m, const double t, const uint32_t k)
{
  const uint32_t R = m.rate_cats;
  const double * par = m.par;
  const double rate = par[par_rates(R) + k];
  const double * pm = par;
  if (m.model != 0) pm = par + par_matrix(R, 4, (uint32_t)par[par_param_idx(R) + k]);
  pmatrix_s4_core(m.dst, m.model, rate, pm, t, k);
}
__device__ __forceinline__ void pmatrix_s4_rec(const MatRec & m, const double * __restrict__ mat_length, const uint32_t k)
{
  pmatrix_s4_rec_t(m, mat_length[m.entry], k);
}

__device__ __forceinline__ void load_vec4(const TaskRec & T, const uint32_t clv_index, const uint32_t k,
                                          const uint32_t n, double v[4])
{
  if (clv_index < T.tips_n)
  {
    const uint32_t code = T.tips[(size_t)clv_index*T.np + n];
    v[0] = (code & 1u) ? 1.0 : 0.0;
    v[1] = (code & 2u) ? 1.0 : 0.0;
    v[2] = (code & 4u) ? 1.0 : 0.0;
    v[3] = (code & 8u) ? 1.0 : 0.0;
  }
  else
  {
    const double2 * p = reinterpret_cast<const double2 *>(
        T.clv + (((size_t)(clv_index - T.tips_n)*T.rate_cats + k)*T.np + n)*4);
    const double2 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
  }
}

// RT > 0: compile-time rate-category count, parent CLVs forwarded in registers; RT == 0: runtime.
// PH = phases compiled in (1 = A, 6 = B+C, 7 = all): the eigen/closed-form P-matrix code of phase A is
// what sets the kernel's register count, so loci that take that path run A and B+C as two launches
// and B+C keeps twice the waves in flight.
template <int BS, int RT, int PH = 7>
__global__ void __launch_bounds__(BS) step_s4_fused_kernel(const PlanDev P)
{
  __shared__ double s_term[BS];
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  const uint32_t gl = b*BS + lane;
  // ---- loads that do not depend on phase A, issued first
  const uint32_t t0 = P.blk_task_off[b], t1 = P.blk_task_off[b+1];
  const uint32_t ro = P.lane_rec[gl];
  const bool active = ro != 0xffffffffu;
  TaskRec T{};
  OpDev op0{}, op1{};
  const uint4 * rp = P.recs + (active ? ro : 0u);
  if (active)
  {
    uint4 * dst = reinterpret_cast<uint4 *>(&T);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(TaskRec)/16); ++i) dst[i] = rp[i];
    // the first two node updates ride along with the header
    uint4 * o0 = reinterpret_cast<uint4 *>(&op0), * o1 = reinterpret_cast<uint4 *>(&op1);
    o0[0] = rp[6]; o0[1] = rp[7]; o1[0] = rp[9]; o1[1] = rp[10];      // 48-byte op slots
  }
  uint32_t sum_rec = 0xffffffffu;
  if ((P.flags & 4u) && lane < t1 - t0) sum_rec = P.task_rec[t0 + lane];

  // ---- phase A: P-matrices of this workgroup's loci
  if ((PH & 1) && (P.flags & 1u))
  {
    const uint32_t e0 = P.mat_off[t0], e1 = P.mat_off[t1];
    if (RT == 1)
    {
      for (uint32_t e = e0 + lane; e < e1; e += BS) pmatrix_s4_rec(P.mat_recs[e], P.mat_length, 0);
    }
    else
    {
      const uint32_t rmax = RT ? (uint32_t)RT : P.pad;
      const uint32_t cnt = (e1 - e0)*rmax;
      for (uint32_t i = lane; i < cnt; i += BS)
      {
        const MatRec m = P.mat_recs[e0 + i/rmax];
        const uint32_t k = i % rmax;
        if (k < m.rate_cats) pmatrix_s4_rec(m, P.mat_length, k);
      }
    }
    __syncthreads();
  }
  if (!(PH & 6) || !(P.flags & 6u)) return;

  // ---- phase B: node updates + root term of this lane's pattern
  double term = 0;
  if (active)
  {
    const uint32_t n = gl - T.lane0, np = T.np;
    const uint32_t R = RT ? (uint32_t)RT : T.rate_cats;
    constexpr int RF = RT ? RT : 1;
    double fwd[RF][4];                // last parent CLV of this pattern (RT > 0 only)
    uint32_t fwd_clv = 0xffffffffu;
    if (P.flags & 2u)
    {
      for (uint32_t o = 0; o < T.nops; ++o)
      {
        OpDev op;
        if (o == 0) op = op0;
        else if (o == 1) op = op1;
        else
        {
          uint4 * d = reinterpret_cast<uint4 *>(&op);
          d[0] = rp[6 + 3*o]; d[1] = rp[7 + 3*o];
        }
        double * out = T.clv + (((size_t)(op.parent_clv - T.tips_n)*R)*np + n)*4;
        bool all_small = true;
        double res[RF][4];
#pragma unroll
        for (uint32_t k = 0; k < (RT ? (uint32_t)RT : R); ++k)
        {
          double lv[4], rv[4], x[4], y[4];
          if (RT && op.left_clv == fwd_clv) { lv[0] = fwd[RT ? k : 0][0]; lv[1] = fwd[RT ? k : 0][1]; lv[2] = fwd[RT ? k : 0][2]; lv[3] = fwd[RT ? k : 0][3]; }
          else load_vec4(T, op.left_clv, k, n, lv);
          if (RT && op.right_clv == fwd_clv) { rv[0] = fwd[RT ? k : 0][0]; rv[1] = fwd[RT ? k : 0][1]; rv[2] = fwd[RT ? k : 0][2]; rv[3] = fwd[RT ? k : 0][3]; }
          else load_vec4(T, op.right_clv, k, n, rv);
          matvec4_p(T.pmat, T.pstride, (size_t)op.left_pmatrix*R  + k, lv, x);
          matvec4_p(T.pmat, T.pstride, (size_t)op.right_pmatrix*R + k, rv, y);
          double2 o0, o1;
          o0.x = x[0]*y[0]; o0.y = x[1]*y[1]; o1.x = x[2]*y[2]; o1.y = x[3]*y[3];
          all_small = all_small && (o0.x < BPA_SCALE_THRESHOLD) && (o0.y < BPA_SCALE_THRESHOLD)
                                && (o1.x < BPA_SCALE_THRESHOLD) && (o1.y < BPA_SCALE_THRESHOLD);
          if (RT) { res[RT ? k : 0][0] = o0.x; res[RT ? k : 0][1] = o0.y; res[RT ? k : 0][2] = o1.x; res[RT ? k : 0][3] = o1.y; }
          else
          {
            double2 * dst = reinterpret_cast<double2 *>(out + (size_t)k*np*4);
            dst[0] = o0; dst[1] = o1;
          }
        }
        bool rescale = false;
        if (op.parent_scaler >= 0)
        {
          uint32_t s = 0;
          if (op.left_scaler  >= 0) s += T.scaler[(size_t)op.left_scaler*np  + n];
          if (op.right_scaler >= 0) s += T.scaler[(size_t)op.right_scaler*np + n];
          if (all_small) { rescale = true; s += 1; }
          T.scaler[(size_t)op.parent_scaler*np + n] = s;
        }
        if (RT)
        {
#pragma unroll
          for (int k = 0; k < RF; ++k)
          {
            if (rescale) { res[k][0] *= BPA_SCALE_FACTOR; res[k][1] *= BPA_SCALE_FACTOR; res[k][2] *= BPA_SCALE_FACTOR; res[k][3] *= BPA_SCALE_FACTOR; }
            double2 * dst = reinterpret_cast<double2 *>(out + (size_t)k*np*4);
            double2 a, c; a.x = res[k][0]; a.y = res[k][1]; c.x = res[k][2]; c.y = res[k][3];
            dst[0] = a; dst[1] = c;
            fwd[k][0] = res[k][0]; fwd[k][1] = res[k][1]; fwd[k][2] = res[k][2]; fwd[k][3] = res[k][3];
          }
          fwd_clv = op.parent_clv;
        }
        else if (rescale)
        {
          for (uint32_t k = 0; k < R; ++k)
          {
            double2 * dst = reinterpret_cast<double2 *>(out + (size_t)k*np*4);
            double2 a = dst[0], c = dst[1];
            a.x *= BPA_SCALE_FACTOR; a.y *= BPA_SCALE_FACTOR; c.x *= BPA_SCALE_FACTOR; c.y *= BPA_SCALE_FACTOR;
            dst[0] = a; dst[1] = c;
          }
        }
      }
    }
    // K2 / K3 at the root (core_likelihood_avx.c:117-150, 251-275)
    const double * par = T.par;
#pragma unroll
    for (uint32_t k = 0; k < (RT ? (uint32_t)RT : R); ++k)
    {
      double c[4];
      if (RT && T.root_clv == fwd_clv) { c[0] = fwd[RT ? k : 0][0]; c[1] = fwd[RT ? k : 0][1]; c[2] = fwd[RT ? k : 0][2]; c[3] = fwd[RT ? k : 0][3]; }
      else load_vec4(T, T.root_clv, k, n, c);
      const uint32_t m = (uint32_t)par[par_param_idx(R) + k];
      const double * f = par + par_matrix(R, 4, m) + pm_freqs(4);
      const double tr = dot4_pair(f[0], f[1], f[2], f[3], c);
      term += tr*par[par_rate_weights(R) + k];
    }
    if (!T.unphased_length)
    {
      double lt = log(term);
      if (T.root_scaler >= 0)
      {
        const uint32_t sc = T.scaler[(size_t)T.root_scaler*np + n];
        if (sc) lt += sc*BPA_LOG_SCALE_THRESHOLD;
      }
      term = lt*T.weights[n];
    }
    P.site_term[T.pat_off + n] = term;
  }
  if (!(P.flags & 4u)) return;

  // ---- phase C: per-locus sum in pattern order
  s_term[lane] = term;
  __syncthreads();
  if (sum_rec != 0xffffffffu)
  {
    const TaskRec * S = reinterpret_cast<const TaskRec *>(P.recs + sum_rec);
    const uint32_t np = S->np, l0 = S->lane0 - b*BS;
    double logl = 0;
    if (S->unphased_length)
      logl = reduce_locus(P.loci[S->locus], s_term + l0);
    else
      for (uint32_t n = 0; n < np; ++n) logl += s_term[l0 + n];
    P.lnl[S->task] = P.bfbeta*logl;
  }
}

// ================================= fused proposal step, S=4, one lane per (pattern, rate) ==
// step_s4_fused_kernel gives a lane all R rate categories of its pattern: R x (two CLVs, two 4x4
// matrices, the result) live at once, ~170 VGPRs, two waves per SIMD, and config 3 (29 patterns x 4
// categories per locus) is latency-bound at a third of the HBM rate.  Here a lane owns ONE plane of ONE
// pattern (lane = k*np + n inside its locus, so a wave reads contiguous CLV planes): a quarter of the
// state per lane, R times the lanes to hide the P-matrix and CLV latencies.  The categories of a
// pattern meet only in the root term — combined through LDS in category order (same additions as the
// one-lane version) — and in the scaling test: loci with scalers stay on step_s4_fused_kernel.
// WITH_A: phase A (the P-matrix code, which alone needs ~160 VGPRs) compiled in; the engine launches
// <BS, true> restricted to phase A and then <BS, false> for B + C with twice the waves in flight
template <int BS, bool WITH_A>
__global__ void __launch_bounds__(BS) step_s4_klane_kernel(const PlanDev P)
{
  __shared__ double s_term[BS], s_tr[BS];
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  const uint32_t gl = b*BS + lane;
  const uint32_t t0 = P.blk_task_off[b], t1 = P.blk_task_off[b+1];
  const uint32_t ro = P.lane_rec[gl];
  const bool active = ro != 0xffffffffu;
  TaskRec T{};
  const uint4 * rp = P.recs + (active ? ro : 0u);
  if (active)
  {
    uint4 * dst = reinterpret_cast<uint4 *>(&T);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(TaskRec)/16); ++i) dst[i] = rp[i];
  }
  uint32_t sum_rec = 0xffffffffu;
  if ((P.flags & 4u) && lane < t1 - t0) sum_rec = P.task_rec[t0 + lane];

  // ---- phase A: P-matrices of this workgroup's loci
  if (WITH_A && (P.flags & 1u))
  {
    const uint32_t e0 = P.mat_off[t0], e1 = P.mat_off[t1];
    const uint32_t rmax = P.pad;
    const uint32_t cnt = (e1 - e0)*rmax;
    for (uint32_t i = lane; i < cnt; i += BS)
    {
      const MatRec m = P.mat_recs[e0 + i/rmax];
      const uint32_t k = i % rmax;
      if (k < m.rate_cats) pmatrix_s4_rec(m, P.mat_length, k);
    }
    __syncthreads();
  }
  if (WITH_A || !(P.flags & 6u)) return;

  // ---- phase B: node updates of this lane's (pattern, category) + its root term
  double tr = 0;
  uint32_t n = 0, k = 0, np = 0, R = 1;
  if (active)
  {
    np = T.np; R = T.rate_cats;
    const uint32_t q = gl - T.lane0;
    k = q/np; n = q - k*np;
    double fwd[4] = {0, 0, 0, 0};
    uint32_t fwd_clv = 0xffffffffu;
    if (P.flags & 2u)
    {
      for (uint32_t o = 0; o < T.nops; ++o)
      {
        OpDev op;
        uint4 * d = reinterpret_cast<uint4 *>(&op);
        d[0] = rp[6 + 3*o]; d[1] = rp[7 + 3*o];
        double lv[4], rv[4], x[4], y[4];
        if (op.left_clv == fwd_clv) { lv[0] = fwd[0]; lv[1] = fwd[1]; lv[2] = fwd[2]; lv[3] = fwd[3]; }
        else load_vec4(T, op.left_clv, k, n, lv);
        if (op.right_clv == fwd_clv) { rv[0] = fwd[0]; rv[1] = fwd[1]; rv[2] = fwd[2]; rv[3] = fwd[3]; }
        else load_vec4(T, op.right_clv, k, n, rv);
        matvec4_p(T.pmat, T.pstride, (size_t)op.left_pmatrix*R  + k, lv, x);
        matvec4_p(T.pmat, T.pstride, (size_t)op.right_pmatrix*R + k, rv, y);
        double2 o0, o1;
        o0.x = x[0]*y[0]; o0.y = x[1]*y[1]; o1.x = x[2]*y[2]; o1.y = x[3]*y[3];
        double2 * dst = reinterpret_cast<double2 *>(T.clv + ((((size_t)(op.parent_clv - T.tips_n)*R) + k)*np + n)*4);
        dst[0] = o0; dst[1] = o1;
        fwd[0] = o0.x; fwd[1] = o0.y; fwd[2] = o1.x; fwd[3] = o1.y;
        fwd_clv = op.parent_clv;
      }
    }
    // K2 at the root (core_likelihood_avx.c:117-150): this category's frequency-weighted sum
    const double * par = T.par;
    double c[4];
    if (T.root_clv == fwd_clv) { c[0] = fwd[0]; c[1] = fwd[1]; c[2] = fwd[2]; c[3] = fwd[3]; }
    else load_vec4(T, T.root_clv, k, n, c);
    const uint32_t m = (uint32_t)par[par_param_idx(R) + k];
    const double * f = par + par_matrix(R, 4, m) + pm_freqs(4);
    tr = dot4_pair(f[0], f[1], f[2], f[3], c);
  }
  s_tr[lane] = tr;
  __syncthreads();
  double term = 0;
  if (active && k == 0)
  {
    const double * par = T.par;
    for (uint32_t q = 0; q < R; ++q) term += s_tr[lane + q*np]*par[par_rate_weights(R) + q];
    term = log(term)*T.weights[n];
    P.site_term[T.pat_off + n] = term;
  }
  if (!(P.flags & 4u)) return;

  // ---- phase C: per-locus sum in pattern order (the k = 0 lanes are the first np lanes of a locus)
  s_term[lane] = term;
  __syncthreads();
  if (sum_rec != 0xffffffffu)
  {
    const TaskRec * S = reinterpret_cast<const TaskRec *>(P.recs + sum_rec);
    const uint32_t snp = S->np, l0 = S->lane0 - b*BS;
    double logl = 0;
    for (uint32_t q = 0; q < snp; ++q) logl += s_term[l0 + q];
    P.lnl[S->task] = P.bfbeta*logl;
  }
}

// ============================================ fused proposal step, JC69, R = 1 ==
// Configs 1/2/5 (JC69, one rate category, a handful of patterns per locus) are pure
// latency: what matters is the length of one lane's dependent-load chain.  Under JC69 a
// P-matrix is two numbers, a (diagonal) and b, so the lanes that need a branch updated in
// this very step compute (a, b) themselves from the branch length carried in the op slot
// (locus.c:2342-2414, same expression => same bits) instead of waiting for phase A and a
// workgroup barrier; matrices not touched in this step are read back as their (a, b) pair
// (16 B instead of 128).  All inputs of the first three node updates that do not depend on
// an earlier update of the step are loaded up front, in one wave of requests; the updates
// then run out of registers (results forwarded).  The full 4x4 matrices are still written
// to HBM for later steps, off the critical path, at the end of the kernel.  Arithmetic is
// the same mat-vec on the same 16 values in the same order: CLVs stay bit-identical.
struct OpSlot { OpDev op; int32_t left_e, right_e, pad0, pad1; };   // 48 B; *_e: entry of mat_length[] when the child's
                                                                    // P-matrix is updated in this very step, else -1

// x / 3, correctly rounded, in three instructions instead of the division's ~35 (Markstein: q0 = x c with c = RN(1/3), the
// exact residual r = x - 3 q0 by one fma, q = q0 + r c by another; 0 differences from x / 3.0 on 4e8 random doubles over
// 200 binades) — JC69's P(t) has two of them per branch, and the sampler kernels make P(t) for every proposal
__device__ __forceinline__ double div3(const double x)
{
  const double c = 1.0/3.0, q0 = x*c;
  return __builtin_fma(__builtin_fma(-3.0, q0, x), c, q0);
}
__device__ __forceinline__ void jc69_ab(const double len, const double rate, double & a, double & b)
{
  const double bl = len*rate;
  a = 1.0; b = 0.0;
  if (!(bl < 1e-100))
  {
    a = (1 + 3*exp(div3(-4*bl)))/4;
    b = div3(1 - a);
  }
}

__device__ __forceinline__ void expand_code(const uint32_t code, double v[4])
{
  v[0] = (code & 1u) ? 1.0 : 0.0; v[1] = (code & 2u) ? 1.0 : 0.0;
  v[2] = (code & 4u) ? 1.0 : 0.0; v[3] = (code & 8u) ? 1.0 : 0.0;
}

// EARLY: the step's fresh (a, b) pairs are produced by the first lanes of the workgroup (one branch each, from the
// MatRec and branch length they fetched at kernel entry) while the update lanes are still waiting for their records,
// handed over through LDS and written to HBM right there: the exponentials (up to six per lane before) and the old
// tail leave the update lanes' critical path for one workgroup barrier that overlaps the record fetch.
template <int BS, bool EARLY = true>
__global__ void __launch_bounds__(BS) step_jc69_kernel(const PlanDev P)
{
  constexpr int NPRE = 3;                       // node updates whose inputs are preloaded
  constexpr uint32_t NAB = 2*BS;                // fresh (a, b) pairs a workgroup can hand over through LDS
  __shared__ double s_term[BS];
  __shared__ double2 s_ab[EARLY ? NAB : 1];
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  const uint32_t gl = b*BS + lane;
  BPA_STAMP(P, b, lane, 0);
  const uint32_t t0 = P.blk_task_off[b], t1 = P.blk_task_off[b+1];
  const uint32_t ro = P.lane_rec[gl];
  const bool active = ro != 0xffffffffu;
  BPA_STAMP(P, b, lane, 1);
  // the tail's first P-matrix record (K4), requested now
  const bool do_mats = (P.flags & 1u) != 0;
  const uint32_t e0 = do_mats ? P.mat_off[t0] : 0u, e1 = do_mats ? P.mat_off[t1] : 0u;
  MatRec m0{};
  double m0_len = 0, m0_rate = 0;
  const bool have_m0 = e0 + lane < e1;
  if (have_m0)
  {
    const uint4 * mp = reinterpret_cast<const uint4 *>(P.mat_recs + e0 + lane);
    uint4 * md = reinterpret_cast<uint4 *>(&m0);
    md[0] = mp[0]; md[1] = mp[1];
    m0_len = P.mat_length[e0 + lane];          // entry == its own index
  }
  // phase C bookkeeping, fetched now so that it has long arrived when needed
  uint32_t c_np = 0, c_l0 = 0, c_task = 0, c_unph = 0, c_locus = 0;
  const bool summer = (P.flags & 4u) && lane < t1 - t0;
  if (summer)
  {
    const TaskRec * S = reinterpret_cast<const TaskRec *>(P.recs + P.task_rec[t0 + lane]);
    c_np = S->np; c_l0 = S->lane0 - b*BS; c_task = S->task; c_unph = S->unphased_length; c_locus = S->locus;
  }

  if (have_m0) m0_rate = m0.par[par_rates(1)];

  const bool work = active && (P.flags & 6u);
  // (tried: the workgroup's contiguous records staged in LDS by all lanes, in parallel with the lane -> record lookup,
  //  to save one dependent hop: 12.0 us instead of 10.85 — the barrier then waits for the slowest lane's share)
  const uint4 * rp = P.recs + (active ? ro : 0u);
  TaskRec T{};
  OpSlot sl[NPRE];
  if (work)
  {
    uint4 * dst = reinterpret_cast<uint4 *>(&T);
#pragma unroll
    for (int i = 0; i < 6; ++i) dst[i] = rp[i];
    uint4 * ds = reinterpret_cast<uint4 *>(sl);
#pragma unroll
    for (int i = 0; i < 3*NPRE; ++i) ds[i] = rp[6 + i];
  }
  const bool use_lds = EARLY && do_mats;
  if (EARLY)
  {
    // K4 for this step's branches (locus.c:2342-2414), one branch per lane, while the records are in flight
    if (have_m0)
    {
      double2 ab;
      jc69_ab(m0_len, m0_rate, ab.x, ab.y);
      *reinterpret_cast<double2 *>(m0.dst) = ab;
      s_ab[lane] = ab;
      for (uint32_t e = e0 + lane + BS; e < e1; e += BS)
      {
        const MatRec m = P.mat_recs[e];
        double2 ab2;
        jc69_ab(P.mat_length[m.entry], m.par[par_rates(1)], ab2.x, ab2.y);
        *reinterpret_cast<double2 *>(m.dst) = ab2;
        if (e - e0 < NAB) s_ab[e - e0] = ab2;
      }
    }
    if (do_mats) __syncthreads();
  }

  double term = 0;
  if (work)
  {
    BPA_STAMP(P, b, lane, 2);
    const uint32_t n = gl - T.lane0, np = T.np, tips = T.tips_n;
    const uint32_t nops = (P.flags & 2u) ? T.nops : 0u;
    const double * par = T.par;

    // ---- one wave of independent input loads
    const double rate = par[par_rates(1)];
    const double rw = par[par_rate_weights(1)];
    const double2 f01 = *reinterpret_cast<const double2 *>(par + par_matrix(1, 4, 0) + pm_freqs(4));   // param_idx 0 (R = 1)
    const double2 f23 = *reinterpret_cast<const double2 *>(par + par_matrix(1, 4, 0) + pm_freqs(4) + 2);
    const uint32_t wgt = T.weights[n];
    double inl[NPRE][4], inr[NPRE][4];          // child vectors (tips expanded / inner from HBM)
    double abl[NPRE][2], abr[NPRE][2];          // (a, b) of the two branches
    uint32_t fwl[NPRE], fwr[NPRE];              // 0..NPRE-1: forwarded from that earlier update; 0xff: in inl/inr
#pragma unroll
    for (int i = 0; i < NPRE; ++i)
    {
      fwl[i] = fwr[i] = 0xffu;
      if ((uint32_t)i < nops)
      {
        const OpDev & op = sl[i].op;
#pragma unroll
        for (int j = 0; j < i; ++j)
        {
          if (sl[j].op.parent_clv == op.left_clv)  fwl[i] = j;
          if (sl[j].op.parent_clv == op.right_clv) fwr[i] = j;
        }
        if (fwl[i] == 0xffu)
        {
          if (op.left_clv < tips) expand_code(T.tips[(size_t)op.left_clv*np + n], inl[i]);
          else
          {
            const double2 * p = reinterpret_cast<const double2 *>(T.clv + ((size_t)(op.left_clv - tips)*np + n)*4);
            const double2 u = p[0], w = p[1];
            inl[i][0] = u.x; inl[i][1] = u.y; inl[i][2] = w.x; inl[i][3] = w.y;
          }
        }
        if (fwr[i] == 0xffu)
        {
          if (op.right_clv < tips) expand_code(T.tips[(size_t)op.right_clv*np + n], inr[i]);
          else
          {
            const double2 * p = reinterpret_cast<const double2 *>(T.clv + ((size_t)(op.right_clv - tips)*np + n)*4);
            const double2 u = p[0], w = p[1];
            inr[i][0] = u.x; inr[i][1] = u.y; inr[i][2] = w.x; inr[i][3] = w.y;
          }
        }
        // abl/abr hold either the stored (a, b) pair or, in [0], the fresh branch length
        if (sl[i].left_e < 0)
        {
          const double2 ab = *reinterpret_cast<const double2 *>(T.pmat + (size_t)op.left_pmatrix*2);
          abl[i][0] = ab.x; abl[i][1] = ab.y;
        }
        else if (use_lds && (uint32_t)sl[i].left_e - e0 < NAB) { const double2 ab = s_ab[(uint32_t)sl[i].left_e - e0]; abl[i][0] = ab.x; abl[i][1] = ab.y; }
        else abl[i][0] = P.mat_length[sl[i].left_e];
        if (sl[i].right_e < 0)
        {
          const double2 ab = *reinterpret_cast<const double2 *>(T.pmat + (size_t)op.right_pmatrix*2);
          abr[i][0] = ab.x; abr[i][1] = ab.y;
        }
        else if (use_lds && (uint32_t)sl[i].right_e - e0 < NAB) { const double2 ab = s_ab[(uint32_t)sl[i].right_e - e0]; abr[i][0] = ab.x; abr[i][1] = ab.y; }
        else abr[i][0] = P.mat_length[sl[i].right_e];
      }
    }

    BPA_STAMP(P, b, lane, 3);
    // ---- node updates out of registers
    double res[NPRE][4];
    uint32_t last_clv = 0xffffffffu;
    double last[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NPRE; ++i)
    {
      if ((uint32_t)i < nops)
      {
        const OpDev & op = sl[i].op;
        double lv[4], rv[4], x[4], y[4], al, bl_, ar, br;
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
          lv[q] = inl[i][q]; rv[q] = inr[i][q];
#pragma unroll
          for (int j = 0; j < i; ++j)
          {
            if (fwl[i] == (uint32_t)j) lv[q] = res[j][q];
            if (fwr[i] == (uint32_t)j) rv[q] = res[j][q];
          }
        }
        if (sl[i].left_e < 0 || (use_lds && (uint32_t)sl[i].left_e - e0 < NAB))   { al = abl[i][0]; bl_ = abl[i][1]; } else jc69_ab(abl[i][0], rate, al, bl_);
        if (sl[i].right_e < 0 || (use_lds && (uint32_t)sl[i].right_e - e0 < NAB)) { ar = abr[i][0]; br = abr[i][1]; } else jc69_ab(abr[i][0], rate, ar, br);
        matvec4_ab(al, bl_, lv, x);
        matvec4_ab(ar, br, rv, y);
        res[i][0] = x[0]*y[0]; res[i][1] = x[1]*y[1]; res[i][2] = x[2]*y[2]; res[i][3] = x[3]*y[3];
        if (op.parent_scaler >= 0)
        {
          uint32_t s = 0;
          if (op.left_scaler  >= 0) s += T.scaler[(size_t)op.left_scaler*np  + n];
          if (op.right_scaler >= 0) s += T.scaler[(size_t)op.right_scaler*np + n];
          if (res[i][0] < BPA_SCALE_THRESHOLD && res[i][1] < BPA_SCALE_THRESHOLD &&
              res[i][2] < BPA_SCALE_THRESHOLD && res[i][3] < BPA_SCALE_THRESHOLD)
          {
            res[i][0] *= BPA_SCALE_FACTOR; res[i][1] *= BPA_SCALE_FACTOR; res[i][2] *= BPA_SCALE_FACTOR; res[i][3] *= BPA_SCALE_FACTOR;
            s += 1;
          }
          T.scaler[(size_t)op.parent_scaler*np + n] = s;
        }
        double2 * dst = reinterpret_cast<double2 *>(T.clv + ((size_t)(op.parent_clv - tips)*np + n)*4);
        double2 u, w; u.x = res[i][0]; u.y = res[i][1]; w.x = res[i][2]; w.y = res[i][3];
        dst[0] = u; dst[1] = w;
        last_clv = op.parent_clv;
        last[0] = res[i][0]; last[1] = res[i][1]; last[2] = res[i][2]; last[3] = res[i][3];
      }
    }
    // ---- further node updates (deeper trees): same arithmetic, inputs loaded at use
    for (uint32_t o = NPRE; o < nops; ++o)
    {
      OpSlot s;
      uint4 * d = reinterpret_cast<uint4 *>(&s);
      d[0] = rp[6 + 3*o]; d[1] = rp[7 + 3*o]; d[2] = rp[8 + 3*o];
      const OpDev & op = s.op;
      double lv[4], rv[4], x[4], y[4], al, bl_, ar, br;
      if (op.left_clv == last_clv) { lv[0] = last[0]; lv[1] = last[1]; lv[2] = last[2]; lv[3] = last[3]; }
      else load_vec4(T, op.left_clv, 0, n, lv);
      if (op.right_clv == last_clv) { rv[0] = last[0]; rv[1] = last[1]; rv[2] = last[2]; rv[3] = last[3]; }
      else load_vec4(T, op.right_clv, 0, n, rv);
      if (s.left_e < 0)  { const double2 ab = *reinterpret_cast<const double2 *>(T.pmat + (size_t)op.left_pmatrix*2);  al = ab.x; bl_ = ab.y; }
      else if (use_lds && (uint32_t)s.left_e - e0 < NAB) { const double2 ab = s_ab[(uint32_t)s.left_e - e0]; al = ab.x; bl_ = ab.y; }
      else jc69_ab(P.mat_length[s.left_e], rate, al, bl_);
      if (s.right_e < 0) { const double2 ab = *reinterpret_cast<const double2 *>(T.pmat + (size_t)op.right_pmatrix*2); ar = ab.x; br = ab.y; }
      else if (use_lds && (uint32_t)s.right_e - e0 < NAB) { const double2 ab = s_ab[(uint32_t)s.right_e - e0]; ar = ab.x; br = ab.y; }
      else jc69_ab(P.mat_length[s.right_e], rate, ar, br);
      matvec4_ab(al, bl_, lv, x);
      matvec4_ab(ar, br, rv, y);
      double r0 = x[0]*y[0], r1 = x[1]*y[1], r2 = x[2]*y[2], r3 = x[3]*y[3];
      if (op.parent_scaler >= 0)
      {
        uint32_t sc = 0;
        if (op.left_scaler  >= 0) sc += T.scaler[(size_t)op.left_scaler*np  + n];
        if (op.right_scaler >= 0) sc += T.scaler[(size_t)op.right_scaler*np + n];
        if (r0 < BPA_SCALE_THRESHOLD && r1 < BPA_SCALE_THRESHOLD && r2 < BPA_SCALE_THRESHOLD && r3 < BPA_SCALE_THRESHOLD)
        { r0 *= BPA_SCALE_FACTOR; r1 *= BPA_SCALE_FACTOR; r2 *= BPA_SCALE_FACTOR; r3 *= BPA_SCALE_FACTOR; sc += 1; }
        T.scaler[(size_t)op.parent_scaler*np + n] = sc;
      }
      double2 * dst = reinterpret_cast<double2 *>(T.clv + ((size_t)(op.parent_clv - tips)*np + n)*4);
      double2 u, w; u.x = r0; u.y = r1; w.x = r2; w.y = r3;
      dst[0] = u; dst[1] = w;
      last_clv = op.parent_clv; last[0] = r0; last[1] = r1; last[2] = r2; last[3] = r3;
    }

    BPA_STAMP(P, b, lane, 4);
    // ---- K2 / K3 at the root (core_likelihood_avx.c:117-150)
    double c[4];
    if (T.root_clv == last_clv) { c[0] = last[0]; c[1] = last[1]; c[2] = last[2]; c[3] = last[3]; }
    else load_vec4(T, T.root_clv, 0, n, c);
    const double tr = dot4_pair(f01.x, f01.y, f23.x, f23.y, c);
    term = 0 + tr*rw;
    if (!T.unphased_length)
    {
      double lt = log(term);
      if (T.root_scaler >= 0)
      {
        const uint32_t sc = T.scaler[(size_t)T.root_scaler*np + n];
        if (sc) lt += sc*BPA_LOG_SCALE_THRESHOLD;
      }
      term = lt*wgt;
    }
    P.site_term[T.pat_off + n] = term;
  }

  BPA_STAMP(P, b, lane, 5);
  // ---- phase C: per-locus sum in pattern order
  if (P.flags & 4u)
  {
    s_term[lane] = term;
    __syncthreads();
    if (summer)
    {
      double logl = 0;
      if (c_unph) logl = reduce_locus(P.loci[c_locus], s_term + c_l0);
      else for (uint32_t q = 0; q < c_np; ++q) logl += s_term[c_l0 + q];
      P.lnl[c_task] = P.bfbeta*logl;
    }
  }
  BPA_STAMP(P, b, lane, 6);

  // ---- tail: the step's P-matrices (their (a, b) pairs) go to HBM for later steps (K4)
  if (!EARLY && have_m0)
  {
    double2 ab;
    jc69_ab(m0_len, m0_rate, ab.x, ab.y);
    *reinterpret_cast<double2 *>(m0.dst) = ab;
    for (uint32_t e = e0 + lane + BS; e < e1; e += BS) pmatrix_s4_rec(P.mat_recs[e], P.mat_length, 0);
  }
  BPA_STAMP(P, b, lane, 7);
}

// The same step on the compact records (device_types.hpp): hop 1 is the engine's lane table (L2-resident, shared by
// all plans: weight, tip codes, position), hop 2 the slot's static entry (L2-resident) next to the step's 16 + 16*ops
// bytes from HBM, hop 3 the inner CLVs — tip states and weights need no load at all, the record address is slot *
// stride.  Workgroups are the ENGINE's packing (the same loci in the same workgroup in every plan: a locus's CLVs
// stay in the L2 of the XCD that works on it); loci that are not part of the plan leave their lanes idle.
// Arithmetic: the very statements of step_jc69_kernel.
//
// Split in two so that a chain of steps (step_jc69_v2_chain_kernel) can have the NEXT step's records in flight while
// this step computes: what a lane reads is (1) per launch — its entries of the engine's tables and the locus's
// parameters (Jc69Lane), (2) per step — the step's records (Jc69Recs: jc69_v2_fetch), then the step itself
// (jc69_v2_compute).
constexpr int JC69_NPRE = 3;
struct Jc69Lane
{
  LaneStatic ls;
  SlotStatic S;                   // valid when the lane has a slot
  uint32_t s0, s1;                // the workgroup's slots
  uint32_t c_np, c_l0, c_unph, c_locus;       // the slot this lane sums in phase C (lane < s1 - s0)
  double rate, rw;
  double2 f01, f23;
};
struct Jc69Recs
{
  uint32_t e0, e1;                // the workgroup's fresh P-matrices of this step
  MatRec2 m0; double m0_len;      // the one this lane computes (e0 + lane < e1)
  uint32_t c_task;
  StepRec hdr;
  StepOp sl[JC69_NPRE];
};

template <int BS>
__device__ __forceinline__ void jc69_v2_lane(const PlanDev & P, Jc69Lane & L)
{
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  L.ls = P.lane_tab[b*BS + lane];
  L.s0 = P.blk_slot_off[b]; L.s1 = P.blk_slot_off[b+1];
  L.c_np = L.c_l0 = L.c_unph = L.c_locus = 0;
  if (lane < L.s1 - L.s0)
  {
    const SlotStatic & C = P.slot_tab[L.s0 + lane];
    L.c_np = C.np; L.c_l0 = C.lane0 - b*BS; L.c_unph = C.unphased_length; L.c_locus = C.locus;
  }
  L.rate = 1; L.rw = 0; L.f01 = double2{0, 0}; L.f23 = double2{0, 0};
  if (L.ls.slot != 0xffffffffu)
  {
    const uint4 * sp = reinterpret_cast<const uint4 *>(P.slot_tab + L.ls.slot);
    uint4 * sd = reinterpret_cast<uint4 *>(&L.S);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(SlotStatic)/16); ++i) sd[i] = sp[i];
    const double * par = L.S.par;
    L.rate = par[par_rates(1)];
    L.rw = par[par_rate_weights(1)];
    L.f01 = *reinterpret_cast<const double2 *>(par + par_matrix(1, 4, 0) + pm_freqs(4));   // param_idx 0 (R = 1)
    L.f23 = *reinterpret_cast<const double2 *>(par + par_matrix(1, 4, 0) + pm_freqs(4) + 2);
  }
}

// the step's records of this lane: requested here, waited for where they are first used
template <int BS>
__device__ __forceinline__ void jc69_v2_fetch(const PlanDev & P, const Jc69Lane & L, Jc69Recs & R)
{
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  const bool has_slot = L.ls.slot != 0xffffffffu;
  const bool do_mats = (P.flags & 1u) != 0;
  R.e0 = do_mats ? gld(P.blk_mat_off + b) : 0u; R.e1 = do_mats ? gld(P.blk_mat_off + b + 1) : 0u;
  R.m0 = MatRec2{0, 0}; R.m0_len = 0;
  if (R.e0 + lane < R.e1)
  {
    const u2v_t m = *reinterpret_cast<const __attribute__((address_space(1))) u2v_t *>(reinterpret_cast<uintptr_t>(P.mat2 + R.e0 + lane));
    R.m0.slot = m.x; R.m0.pmatrix = m.y; R.m0_len = gld(P.mat_length + R.e0 + lane);
  }
  R.c_task = 0xffffffffu;
  if ((P.flags & 4u) && lane < L.s1 - L.s0)
    R.c_task = gld(reinterpret_cast<const uint32_t *>(P.recs2 + (size_t)(L.s0 + lane)*P.rec2_units));
  R.hdr = StepRec{};
  R.hdr.task = 0xffffffffu;
  if (has_slot && (P.flags & 6u))
  {
    const uint4 * rp = P.recs2 + (size_t)L.ls.slot*P.rec2_units;
    *reinterpret_cast<uint4 *>(&R.hdr) = gld4(rp);
    uint4 * ds = reinterpret_cast<uint4 *>(R.sl);
#pragma unroll
    for (int i = 0; i < JC69_NPRE; ++i) ds[i] = gld4(rp + 1 + i);
  }
}

template <int BS>
__device__ __forceinline__ void jc69_v2_compute(const PlanDev & P, const Jc69Lane & L, const Jc69Recs & R,
                                                double * s_term, double2 * s_ab, double * s_lnl)
{
  constexpr int NPRE = JC69_NPRE;
  constexpr uint32_t NAB = 2*BS;
  static_assert(sizeof(LaneStatic) == 16 && sizeof(SlotStatic) == 80 && sizeof(StepRec) == 16 && sizeof(StepOp) == 16 && sizeof(MatRec2) == 8, "compact records");
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  const LaneStatic & ls = L.ls;
  const SlotStatic & S = L.S;
  const uint32_t s0 = L.s0, s1 = L.s1;
  const bool has_slot = ls.slot != 0xffffffffu;
  const bool do_mats = (P.flags & 1u) != 0;
  const uint32_t e0 = R.e0, e1 = R.e1;
  // K4 for this step's branches (locus.c:2342-2414), one branch per lane
  // (an entry with slot 0xffffffff is a hole: the device-written step images of gsampler.hpp keep a fixed number of
  //  entries per locus)
  const bool have_m0 = e0 + lane < e1;
  const bool summer = (P.flags & 4u) && lane < s1 - s0;
  double c_lnl = 0;
  const uint32_t c_np = L.c_np, c_l0 = L.c_l0, c_task = R.c_task, c_unph = L.c_unph, c_locus = L.c_locus;
  const StepRec & hdr = R.hdr;
  const StepOp * sl = R.sl;
  const uint4 * rp = P.recs2 + (size_t)(has_slot ? ls.slot : 0u)*P.rec2_units;
  if (have_m0)
  {
    if (R.m0.slot != 0xffffffffu)
    {
      // the slot entry carries the locus's rate (engine_pack): one hop from the matrix record to the exponential
      const SlotStatic & M = P.slot_tab[R.m0.slot];
      double2 ab;
      jc69_ab(R.m0_len, M.rate0, ab.x, ab.y);
      gst2(M.pmat + (size_t)R.m0.pmatrix*2, ab);
      s_ab[lane] = ab;
    }
    for (uint32_t e = e0 + lane + BS; e < e1; e += BS)
    {
      const MatRec2 m = P.mat2[e];
      if (m.slot == 0xffffffffu) continue;
      const SlotStatic & M2 = P.slot_tab[m.slot];
      double2 ab2;
      jc69_ab(P.mat_length[e], M2.rate0, ab2.x, ab2.y);
      gst2(M2.pmat + (size_t)m.pmatrix*2, ab2);
      if (e - e0 < NAB) s_ab[e - e0] = ab2;
    }
  }

  const bool work = has_slot && hdr.task != 0xffffffffu && (P.flags & 6u);
  const uint32_t n = ls.n_np_tips & 511u, np = (ls.n_np_tips >> 9) & 511u, tips = S.tips_n;      // (the lane entry's own tips field is 5 bits)
  const uint32_t nops = (work && (P.flags & 2u)) ? hdr.nops : 0u;
  double inl[NPRE][4], inr[NPRE][4];          // child vectors (tips expanded / inner from HBM)
  uint32_t fwl[NPRE], fwr[NPRE];              // 0..NPRE-1: forwarded from that earlier update; 0xff: in inl/inr
  const double rate = L.rate, rw = L.rw;
  const double2 f01 = L.f01, f23 = L.f23;
  if (work)
  {
    // ---- one wave of independent input loads (before the workgroup waits for the fresh (a, b) pairs)
#pragma unroll
    for (int i = 0; i < NPRE; ++i)
    {
      fwl[i] = fwr[i] = 0xffu;
      if ((uint32_t)i < nops)
      {
        const StepOp & op = sl[i];
#pragma unroll
        for (int j = 0; j < i; ++j)
        {
          if (sl[j].parent_clv == op.left_clv)  fwl[i] = j;
          if (sl[j].parent_clv == op.right_clv) fwr[i] = j;
        }
        if (fwl[i] == 0xffu)
        {
          if (op.left_clv < tips) expand_code(tips <= 8 ? (ls.tipcodes >> (4*op.left_clv)) & 15u : (uint32_t)S.tips[(size_t)op.left_clv*np + n], inl[i]);
          else
          {
            const double * p = S.clv + ((size_t)(op.left_clv - tips)*np + n)*4;
            const double2 u = gld2(p), w = gld2(p + 2);
            inl[i][0] = u.x; inl[i][1] = u.y; inl[i][2] = w.x; inl[i][3] = w.y;
          }
        }
        if (fwr[i] == 0xffu)
        {
          if (op.right_clv < tips) expand_code(tips <= 8 ? (ls.tipcodes >> (4*op.right_clv)) & 15u : (uint32_t)S.tips[(size_t)op.right_clv*np + n], inr[i]);
          else
          {
            const double * p = S.clv + ((size_t)(op.right_clv - tips)*np + n)*4;
            const double2 u = gld2(p), w = gld2(p + 2);
            inr[i][0] = u.x; inr[i][1] = u.y; inr[i][2] = w.x; inr[i][3] = w.y;
          }
        }
      }
    }
  }
  if (do_mats) __syncthreads();                 // the fresh pairs: handed over through LDS, and their HBM copies (read by later steps of a
                                                // chain, by lanes other than the writer) are complete before anyone goes on

  double term = 0;
  if (work)
  {
    // (a, b) of a branch: fresh from this step (LDS, or computed here when the workgroup has more than NAB of them)
    // or the stored pair
    auto pair_of = [&](int32_t e, uint32_t pm, double & a_, double & b_)
    {
      if (e < 0) { const double2 ab = gld2(S.pmat + (size_t)pm*2); a_ = ab.x; b_ = ab.y; }
      else if (do_mats && (uint32_t)e - e0 < NAB) { const double2 ab = s_ab[(uint32_t)e - e0]; a_ = ab.x; b_ = ab.y; }
      else jc69_ab(gld(P.mat_length + e), rate, a_, b_);
    };
    double res[NPRE][4];
    uint32_t last_clv = 0xffffffffu;
    double last[4] = {0, 0, 0, 0};
    auto finish = [&](const StepOp & op, double & r0, double & r1, double & r2, double & r3)
    {
      if (op.parent_scaler >= 0)
      {
        uint32_t sc = 0;
        if (op.left_scaler  >= 0) sc += gld(S.scaler + (size_t)op.left_scaler*np  + n);
        if (op.right_scaler >= 0) sc += gld(S.scaler + (size_t)op.right_scaler*np + n);
        if (r0 < BPA_SCALE_THRESHOLD && r1 < BPA_SCALE_THRESHOLD && r2 < BPA_SCALE_THRESHOLD && r3 < BPA_SCALE_THRESHOLD)
        { r0 *= BPA_SCALE_FACTOR; r1 *= BPA_SCALE_FACTOR; r2 *= BPA_SCALE_FACTOR; r3 *= BPA_SCALE_FACTOR; sc += 1; }
        gst(S.scaler + (size_t)op.parent_scaler*np + n, sc);
      }
      double * dst = S.clv + ((size_t)(op.parent_clv - tips)*np + n)*4;
      double2 u, w; u.x = r0; u.y = r1; w.x = r2; w.y = r3;
      gst2(dst, u); gst2(dst + 2, w);
      last_clv = op.parent_clv; last[0] = r0; last[1] = r1; last[2] = r2; last[3] = r3;
    };
#pragma unroll
    for (int i = 0; i < NPRE; ++i)
    {
      if ((uint32_t)i < nops)
      {
        const StepOp & op = sl[i];
        double lv[4], rv[4], x[4], y[4], al, bl_, ar, br;
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
          lv[q] = inl[i][q]; rv[q] = inr[i][q];
#pragma unroll
          for (int j = 0; j < i; ++j)
          {
            if (fwl[i] == (uint32_t)j) lv[q] = res[j][q];
            if (fwr[i] == (uint32_t)j) rv[q] = res[j][q];
          }
        }
        pair_of(op.left_e, op.left_pmatrix, al, bl_);
        pair_of(op.right_e, op.right_pmatrix, ar, br);
        matvec4_ab(al, bl_, lv, x);
        matvec4_ab(ar, br, rv, y);
        res[i][0] = x[0]*y[0]; res[i][1] = x[1]*y[1]; res[i][2] = x[2]*y[2]; res[i][3] = x[3]*y[3];
        finish(op, res[i][0], res[i][1], res[i][2], res[i][3]);
      }
    }
    // ---- further node updates (deeper trees): same arithmetic, inputs loaded at use
    auto vec_of = [&](uint32_t c, double v[4])
    {
      if (c == last_clv) { v[0] = last[0]; v[1] = last[1]; v[2] = last[2]; v[3] = last[3]; }
      else if (c < tips) expand_code(tips <= 8 ? (ls.tipcodes >> (4*c)) & 15u : (uint32_t)S.tips[(size_t)c*np + n], v);
      else
      {
        const double * p = S.clv + ((size_t)(c - tips)*np + n)*4;
        const double2 u = gld2(p), w = gld2(p + 2);
        v[0] = u.x; v[1] = u.y; v[2] = w.x; v[3] = w.y;
      }
    };
    for (uint32_t o = NPRE; o < nops; ++o)
    {
      StepOp op;
      *reinterpret_cast<uint4 *>(&op) = gld4(rp + 1 + o);
      double lv[4], rv[4], x[4], y[4], al, bl_, ar, br;
      vec_of(op.left_clv, lv);
      vec_of(op.right_clv, rv);
      pair_of(op.left_e, op.left_pmatrix, al, bl_);
      pair_of(op.right_e, op.right_pmatrix, ar, br);
      matvec4_ab(al, bl_, lv, x);
      matvec4_ab(ar, br, rv, y);
      double r0 = x[0]*y[0], r1 = x[1]*y[1], r2 = x[2]*y[2], r3 = x[3]*y[3];
      finish(op, r0, r1, r2, r3);
    }

    // ---- K2 / K3 at the root (core_likelihood_avx.c:117-150)
    double c[4];
    vec_of(hdr.root_clv, c);
    const double tr = dot4_pair(f01.x, f01.y, f23.x, f23.y, c);
    term = 0 + tr*rw;
    if (!S.unphased_length)
    {
      double lt = log(term);
      if (hdr.root_scaler >= 0)
      {
        const uint32_t sc = gld(S.scaler + (size_t)hdr.root_scaler*np + n);
        if (sc) lt += sc*BPA_LOG_SCALE_THRESHOLD;
      }
      term = lt*ls.wgt;
    }
    gst(P.site_term + hdr.pat_off + n, term);
  }

  // ---- phase C: per-locus sum in pattern order
  if (P.flags & 4u)
  {
    s_term[lane] = term;
    lds_barrier();
    if (summer && c_task != 0xffffffffu)
    {
      double logl = 0;
      if (c_unph) logl = reduce_locus(P.loci[c_locus], s_term + c_l0);
      else for (uint32_t q = 0; q < c_np; ++q) logl += s_term[c_l0 + q];
      gst(P.lnl + c_task, P.bfbeta*logl);
      c_lnl = P.bfbeta*logl;
    }
    // ---- partial sums of the plan's total (bpa_plan_enable_partial_sums): this workgroup's loci in slot order; the
    // consumer (a decision kernel, or the host after the all-reduce over the GPUs) adds the workgroups' values up
    if (P.flags & 8u)
    {
      s_lnl[lane] = c_lnl;
      lds_barrier();
      if (lane == 0)
      {
        double part = 0;
        for (uint32_t q = 0; q < s1 - s0; ++q) part += s_lnl[q];
        gst(P.wg_part + b, part);
      }
    }
  }
}

template <int BS>
__global__ void __launch_bounds__(BS) step_jc69_v2_kernel(const PlanDev P)
{
  __shared__ double s_term[BS], s_lnl[BS];
  __shared__ double2 s_ab[2*BS];
  Jc69Lane L; Jc69Recs R;
  jc69_v2_lane<BS>(P, L);
  jc69_v2_fetch<BS>(P, L, R);
  jc69_v2_compute<BS>(P, L, R, s_term, s_ab, s_lnl);
}

// A CHAIN of steps in one launch: the per-locus proposals of an iteration (GAGE, GSPR: gtree.c:4585, 6531) need nothing
// from other loci, so a workgroup walks its loci through all of them without coming back to the host — the shape of
// threads.c:87-200, where a worker walks its loci's proposals without a barrier.  Step k + 1 of a locus reads what step k
// wrote: CLVs by the same lane, (a, b) pairs by lanes of the same workgroup (the engine's packing never splits a locus),
// so a workgroup barrier between steps orders them; the tables' entries and the parameters are loaded once, and the
// records of step k + 1 are in flight while step k computes.  Each step keeps its own records and its own result arrays
// (ChainStep), i.e. a chain of resident plans is launched, not a new kind of plan.
struct ChainStep
{
  const uint4 *    recs2;
  const MatRec2 *  mat2;
  const double *   mat_length;
  const uint32_t * blk_mat_off;
  double *         site_term;
  double *         lnl;
  double *         wg_part;
  uint32_t         rec2_units, flags;
};
constexpr int BPA_CHAIN_MAX = 24;
struct ChainDev
{
  PlanDev   base;                       // the engine's tables (lane_tab, slot_tab, blk_slot_off, loci, bfbeta)
  uint32_t  nsteps, pad;
  ChainStep st[BPA_CHAIN_MAX];
};

__device__ __forceinline__ void chain_plan(const ChainDev & C, uint32_t k, PlanDev & P)
{
  const ChainStep & st = C.st[k];
  P.recs2 = st.recs2; P.mat2 = st.mat2; P.mat_length = st.mat_length; P.blk_mat_off = st.blk_mat_off;
  P.site_term = st.site_term; P.lnl = st.lnl; P.wg_part = st.wg_part; P.rec2_units = st.rec2_units; P.flags = st.flags;
}

template <int BS>
__global__ void __launch_bounds__(BS) step_jc69_v2_chain_kernel(const ChainDev C)
{
  __shared__ double s_term[BS], s_lnl[BS];
  __shared__ double2 s_ab[2*BS];
  Jc69Lane L;
  jc69_v2_lane<BS>(C.base, L);
  PlanDev P = C.base, Pn = C.base;
  Jc69Recs R, Rn;
  chain_plan(C, 0, P);
  jc69_v2_fetch<BS>(P, L, R);
  for (uint32_t k = 0; k < C.nsteps; ++k)
  {
    if (k + 1 < C.nsteps) { chain_plan(C, k + 1, Pn); jc69_v2_fetch<BS>(Pn, L, Rn); }    // next step's records: in flight
    jc69_v2_compute<BS>(P, L, R, s_term, s_ab, s_lnl);
    lds_barrier();                      // the next step reuses the LDS arrays.  No wait for this step's stores: CLVs are re-read only by the
                                        // lane that wrote them (program order), the (a, b) pairs were completed at the barrier inside the step
    P = Pn; R = Rn;
  }
}

// step_s4_klane_kernel on the compact records over the engine's packing (device_types.hpp; see step_jc69_v2_kernel):
// several rate categories, any 4-state model, no scalers, no phase averaging.  WITH_A: the P-matrix phase as its own
// launch (its eigen / closed-form code needs twice the registers of the node updates).
// A/B switch of the staging below: flags bit 4 = every lane fetches its matrices itself (BPA_KLANE_DIRECT=1)
__device__ __forceinline__ bool getenv_klane_direct(const PlanDev & P) { return (P.flags & 16u) != 0; }

// The P-matrix phase of the compact-record path as a DENSE grid: one lane per (fresh branch entry, rate category).
// step_s4_klane_v2_kernel<.., true> walks the entries workgroup by workgroup of the engine's packing — a workgroup there
// is 256 (pattern, category) lanes = two or three loci, i.e. ~40 entries for 256 lanes, and every mostly-empty wave still
// runs the whole eigen / closed-form code: config 3 measured 27 us per step at 16 % lane use.  Nothing in this phase needs
// the packing (entries carry their slot), so the lanes are dealt out over the entries themselves.
__global__ void __launch_bounds__(256) pmatrix_s4_dense_kernel(const PlanDev P, const uint32_t nent)
{
  const uint32_t rmax = P.pad ? P.pad : 1u, i = blockIdx.x*256u + threadIdx.x;
  const uint32_t e = P.ent0 + i/rmax, k = i % rmax;
  if (e >= nent) return;
  const u2v_t mm = *reinterpret_cast<const __attribute__((address_space(1))) u2v_t *>(reinterpret_cast<uintptr_t>(P.mat2 + e));
  if (mm.x == 0xffffffffu) return;                           // a hole of a device-written step image (gsampler.hpp)
  const SlotStatic & M = P.slot_tab[mm.x];
  const uint32_t R = M.rate_cats;
  if (k >= R) return;
  MatRec m;
  m.dst = M.pmat + (size_t)mm.y*R*M.pstride; m.par = M.par; m.rate_cats = R; m.model = M.model; m.entry = e; m.pad = 0;
  pmatrix_s4_rec(m, P.mat_length, k);
}

template <int N>
__device__ __forceinline__ void update_eigen_regs(const double * __restrict__ freqs, const double * __restrict__ subst,
                                                  double * __restrict__ evals, double * __restrict__ evecs, double * __restrict__ ievecs);
// FUSE_A: the block's P-matrix entries first, a workgroup barrier, then its node updates — ONE launch for a step of a small
// set (a strong-scaling rank's share), where the dense P-matrix launch is 6.6 us + a launch boundary for microseconds of work;
// the registers of both phases in one kernel cost occupancy, so large sets keep the two launches
template <int BS, bool WITH_A, int OCC = 0, bool FUSE_A = false>        // OCC: waves per SIMD the register allocation is held to (0: the compiler's choice)
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(OCC ? OCC : 1, OCC ? OCC : 8)))
step_s4_klane_v2_kernel(const PlanDev P)
{
  __shared__ double s_term[BS], s_tr[BS];
  const uint32_t b = P.blk0 + blockIdx.x, lane = threadIdx.x, gl = b*BS + lane;
  if (FUSE_A && (P.flags & 32u))
  {
    // K6 first: the block's loci had their frequencies / exchangeabilities moved since their eigensystems were made
    // (a parameter step of the device sampler: the launch of eigen_kernel it replaces was 16 us + a launch boundary)
    const uint32_t q0 = P.blk_slot_off[b], q1 = P.blk_slot_off[b+1];
    if (lane < q1 - q0)
    {
      const LocusDev & L = P.loci[P.slot_tab[q0 + lane].locus];
      // (a JC69 locus — the other loci of a composite's packing — has no eigensystem.  NOT "only the loci with a record in this
      //  step": a locus whose rejected parameter proposal was just rolled back may sit this step out, and the refresh is owed once)
      const bool mine = L.model != 0u;
      for (uint32_t m = 0; mine && m < L.rate_matrices; ++m)
      {
        double * pm = L.par + par_matrix(L.rate_cats, 4, m);
        update_eigen_regs<4>(pm + pm_freqs(4), pm + pm_subst(4), pm + pm_evals(4), pm + pm_evecs(4), pm + pm_ievecs(4));
      }
    }
    __syncthreads();
  }
  if (WITH_A || (FUSE_A && (P.flags & 1u)))
  {
    if (!(P.flags & 1u)) return;
    const uint32_t e0 = P.blk_mat_off[b], e1 = P.blk_mat_off[b+1], rmax = P.pad;
    for (uint32_t i = lane; i < (e1 - e0)*rmax; i += BS)
    {
      const uint32_t e = e0 + i/rmax, k = i % rmax;
      const MatRec2 m2 = P.mat2[e];
      if (m2.slot == 0xffffffffu) continue;                  // a hole of a device-written step image (gsampler.hpp)
      const SlotStatic & M = P.slot_tab[m2.slot];
      if (k >= M.rate_cats) continue;
      MatRec m;
      m.dst = M.pmat + (size_t)m2.pmatrix*M.rate_cats*M.pstride; m.par = M.par; m.rate_cats = M.rate_cats; m.model = M.model; m.entry = e; m.pad = 0;
      pmatrix_s4_rec(m, P.mat_length, k);
    }
    if (WITH_A) return;
    __syncthreads();                                         // the block's fresh P-matrices are out (written and read on this CU)
  }
  const uint32_t s0 = P.blk_slot_off[b], s1 = P.blk_slot_off[b+1];
  const LaneStatic ls = P.lane_tab[gl];
  const bool has_slot = ls.slot != 0xffffffffu;
  const bool summer = (P.flags & 4u) && lane < s1 - s0;
  uint32_t c_np = 0, c_l0 = 0, c_task = 0xffffffffu;
  double c_lnl = 0;
  if (summer)
  {
    const SlotStatic & C = P.slot_tab[s0 + lane];
    c_np = C.np; c_l0 = C.lane0 - b*BS;
    c_task = reinterpret_cast<const StepRec *>(P.recs2 + (size_t)(s0 + lane)*P.rec2_units)->task;
  }
  SlotStatic S{};
  StepRec hdr{};
  hdr.task = 0xffffffffu;
  const uint4 * rp = P.recs2 + (size_t)(has_slot ? ls.slot : 0u)*P.rec2_units;
  if (has_slot && (P.flags & 6u))
  {
    const uint4 * sp = reinterpret_cast<const uint4 *>(P.slot_tab + ls.slot);
    uint4 * sd = reinterpret_cast<uint4 *>(&S);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(SlotStatic)/16); ++i) sd[i] = sp[i];
    *reinterpret_cast<uint4 *>(&hdr) = rp[0];
  }
  const bool work = has_slot && hdr.task != 0xffffffffu && (P.flags & 6u);
  const uint32_t n = ls.n_np_tips & 511u, np = (ls.n_np_tips >> 9) & 511u, tips = S.tips_n;
  const uint32_t k = (ls.n_np_tips >> 23) & 7u, R = ls.n_np_tips >> 26;

  // ---- node updates of this lane's (pattern, category) + its root term
  // The two 4x4 P-matrices of an update are the same 256 bytes for all lanes of one (locus, category) — np lanes in a
  // row — yet fetched by every lane they are 32 16-byte load instructions per update through the CU's one
  // texture-address path (64 B/clk): with 20 waves per CU that path, not HBM, set the kernel's time.  Here each WAVE
  // stages the matrices of the (at most four) groups it spans in its own LDS corner — ONE 16-byte load per lane per
  // update: lane j brings chunk j % 16 of group j / 16 — and every lane reads its group's rows back as LDS broadcasts.
  // No workgroup barrier: writer and readers are the same wave, whose LDS instructions execute in order.  Loci of one
  // wave have different update lists, so a staging lane follows the list of the locus it stages FOR.
  __shared__ double2 s_pm[BS/64][4][16];
  const uint32_t wave = lane >> 6, wl = lane & 63u;
  bool use_lds;
  uint32_t my_g = 0;
  const uint4 * st_rp = nullptr; const double * st_pmat = nullptr; uint32_t st_R = 0, st_k = 0, st_nops = 0;
  {
    static const uint32_t none = 0xffffffffu;
    const bool grouped = work && (P.flags & 2u);
    const uint32_t gkey = grouped ? (ls.slot << 3 | k) : none;
    const uint32_t prevkey = __shfl_up(gkey, 1);
    const bool first = grouped && (wl == 0 || gkey != prevkey);
    const unsigned long long fmask = __ballot(first);
    const unsigned long long odd = __ballot(grouped && S.pstride != 16u);      // (a, b) pairs of JC69 loci: nothing to stage
    const uint32_t ngroups = (uint32_t)__popcll(fmask);
    use_lds = ngroups >= 1 && ngroups <= 4 && odd == 0 && !getenv_klane_direct(P);
    my_g = (uint32_t)__popcll(fmask & ((2ull << wl) - 1ull)) - 1u;
    // the group this lane stages for: its first lane holds slot, category, R and the buffer address
    const uint32_t gs = wl >> 4;
    uint32_t src = 0;
    { unsigned long long m = fmask; for (uint32_t i = 0; i < gs; ++i) m &= m - 1ull; src = m ? (uint32_t)__ffsll((long long)m) - 1u : 0u; }
    const uint32_t s_slot = __shfl(ls.slot, (int)src), s_kR = __shfl(k | R << 3, (int)src);
    const unsigned long long s_pm_lo = __shfl((uint32_t)reinterpret_cast<uintptr_t>(S.pmat), (int)src);
    const unsigned long long s_pm_hi = __shfl((uint32_t)(reinterpret_cast<uintptr_t>(S.pmat) >> 32), (int)src);
    const uint32_t s_n = __shfl((uint32_t)hdr.nops, (int)src);
    if (use_lds && gs < ngroups)
    {
      st_rp = P.recs2 + (size_t)s_slot*P.rec2_units;
      st_pmat = reinterpret_cast<const double *>(s_pm_lo | s_pm_hi << 32);
      st_k = s_kR & 7u; st_R = s_kR >> 3; st_nops = s_n;
    }
  }
  double tr = 0;
  {
    double fwd[4] = {0, 0, 0, 0};
    uint32_t fwd_clv = 0xffffffffu;
    auto vec_of = [&](uint32_t c, double v[4])
    {
      if (c == fwd_clv) { v[0] = fwd[0]; v[1] = fwd[1]; v[2] = fwd[2]; v[3] = fwd[3]; }
      else if (c < tips) expand_code(tips <= 8 ? (ls.tipcodes >> (4*c)) & 15u : (uint32_t)S.tips[(size_t)c*np + n], v);
      else
      {
        typedef double d2v __attribute__((ext_vector_type(2)));
        const __attribute__((address_space(1))) d2v * p = reinterpret_cast<const __attribute__((address_space(1))) d2v *>(
            reinterpret_cast<uintptr_t>(S.clv + ((((size_t)(c - tips)*R) + k)*np + n)*4));
        const d2v uu = __builtin_nontemporal_load(p), ww = __builtin_nontemporal_load(p + 1);
        double2 u, w; u.x = uu.x; u.y = uu.y; w.x = ww.x; w.y = ww.y;
        v[0] = u.x; v[1] = u.y; v[2] = w.x; v[3] = w.y;
      }
    };
    if (P.flags & 2u)
    {
      const uint32_t nops = work ? hdr.nops : 0u;
      // the wave walks update index o together: its lanes' loci may have lists of different lengths
      // (requesting the records of update o + 1 while update o runs was tried: no gain, 106 vs 105 us)
      for (uint32_t o = 0; __any(o < nops); ++o)
      {
        if (use_lds)
        {
          if (o < st_nops)
          {
            StepOp so;
            *reinterpret_cast<uint4 *>(&so) = gld4(st_rp + 1 + o);
            const uint32_t c = wl & 15u, pm = c < 8u ? so.left_pmatrix : so.right_pmatrix;
            s_pm[wave][wl >> 4][c] = gld2(st_pmat + ((size_t)pm*st_R + st_k)*16 + (size_t)(c & 7u)*2);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (o < nops)
        {
          StepOp op;
          *reinterpret_cast<uint4 *>(&op) = gld4(rp + 1 + o);
          double lv[4], rv[4], x[4], y[4];
          vec_of(op.left_clv, lv);
          vec_of(op.right_clv, rv);
          if (use_lds)
          {
            // one matrix at a time (16 doubles live, not 32: the kernel keeps 5 waves per SIMD)
            const double2 * r = &s_pm[wave][my_g][0];
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
              const double2 a0 = r[2*i], b0 = r[2*i+1];
              x[i] = dot4_pair(a0.x, a0.y, b0.x, b0.y, lv);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
              const double2 a1 = r[8 + 2*i], b1 = r[8 + 2*i+1];
              y[i] = dot4_pair(a1.x, a1.y, b1.x, b1.y, rv);
            }
          }
          else
          {
            matvec4_p(S.pmat, S.pstride, (size_t)op.left_pmatrix*R  + k, lv, x);
            matvec4_p(S.pmat, S.pstride, (size_t)op.right_pmatrix*R + k, rv, y);
          }
          double2 o0, o1;
          o0.x = x[0]*y[0]; o0.y = x[1]*y[1]; o1.x = x[2]*y[2]; o1.y = x[3]*y[3];
          { typedef double d2v __attribute__((ext_vector_type(2))); d2v a0, a1; a0.x = o0.x; a0.y = o0.y; a1.x = o1.x; a1.y = o1.y;
            __attribute__((address_space(1))) d2v * dst = reinterpret_cast<__attribute__((address_space(1))) d2v *>(
                reinterpret_cast<uintptr_t>(S.clv + ((((size_t)(op.parent_clv - tips)*R) + k)*np + n)*4));
            __builtin_nontemporal_store(a0, dst); __builtin_nontemporal_store(a1, dst + 1); }
          fwd[0] = o0.x; fwd[1] = o0.y; fwd[2] = o1.x; fwd[3] = o1.y;
          fwd_clv = op.parent_clv;
        }
        if (use_lds)
        {
          // the next round's staging must not overtake this round's reads (compiler: LDS order of one wave is the hardware's)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
    }
    if (work)
    {
      // K2 at the root (core_likelihood_avx.c:117-150): this category's frequency-weighted sum
      const double * par = S.par;
      double c[4];
      vec_of(hdr.root_clv, c);
      const uint32_t m = (uint32_t)par[par_param_idx(R) + k];
      const double * f = par + par_matrix(R, 4, m) + pm_freqs(4);
      tr = dot4_pair(f[0], f[1], f[2], f[3], c);
    }
  }
  s_tr[lane] = tr;
  __syncthreads();
  double term = 0;
  if (work && k == 0)
  {
    const double * par = S.par;
    for (uint32_t q = 0; q < R; ++q) term += s_tr[lane + q*np]*par[par_rate_weights(R) + q];
    term = log(term)*ls.wgt;
    gst(P.site_term + hdr.pat_off + n, term);
  }
  if (!(P.flags & 4u)) return;

  // ---- per-locus sum in pattern order (the k = 0 lanes are the first np lanes of a locus)
  s_term[lane] = term;
  __syncthreads();
  if (summer && c_task != 0xffffffffu)
  {
    double logl = 0;
    for (uint32_t q = 0; q < c_np; ++q) logl += s_term[c_l0 + q];
    P.lnl[c_task] = P.bfbeta*logl;
    c_lnl = P.bfbeta*logl;
  }
  if (P.flags & 8u)
  {
    __shared__ double s_lnl[BS];
    s_lnl[lane] = c_lnl;
    __syncthreads();
    if (lane == 0)
    {
      double part = 0;
      for (uint32_t q = 0; q < s1 - s0; ++q) part += s_lnl[q];
      P.wg_part[b] = part;
    }
  }
}

// ================================================================ step_s4_klane_v3_kernel (round 5) ==
// step_s4_klane_v2_kernel with its dependent round trips cut from ~2 per node update to 3 per STEP.  v2 walks a locus's update
// list one record at a time: record -> (child CLV, the wave's P-matrix chunks) -> compute -> next record ..., i.e. 2 nops + 2
// global round trips in series with ~150 cycles of arithmetic between them — rocprofv3 had 82 % of its wave-cycles waiting and
// 32 % of its LDS cycles in bank conflicts (profiles/r4/profile_c3.json).  Here (a step of nops updates: 2 + ceil(nops / CH) trips):
//   trip 1  the lane's table entry (as before);
//   trip 2  the slot record and header AND every record of the step for the (at most four) groups the wave spans: lanes
//           16 g .. 16 g + 15 bring the 16 sixteen-byte units of group g's record (header + up to 15 updates) into LDS — the
//           update list is then read from LDS by index, no global load per update;
//   trip 3  (once per CH = 2 updates) their P-matrices, global -> LDS directly (global_load_lds_dwordx4 from inline assembly,
//           lds_dma16: no registers, and no s_waitcnt vmcnt(0) of the compiler's in front of every later ds_read) and their
//           children that can be read ahead: inner nodes that no update of this step writes (a bit mask of the step's parents
//           decides; a child that is the previous update's parent is forwarded in registers as before, a child written earlier
//           in the step is read when it is used, after the lane's own store).  ONE explicit wait for all of it, shown to the
//           compiler (empty read-modify statements on the registers read ahead), so that no update waits for an earlier
//           update's stores.
// Bank conflicts: the chunk j of group g sits at position 16 g + (j ^ g) of the wave's 1-KB corner, so lanes of different
// groups reading "their" chunk j hit different banks (v2: all four groups on the same banks whenever a 16-lane service group
// of ds_read_b128 spans two groups).  Arithmetic, operation order and stores are v2's: same bits.
#ifndef BPA_KLANE_OCC
#define BPA_KLANE_OCC 4
#endif
#ifndef BPA_KLANE_CH
#define BPA_KLANE_CH 2
#endif
template <int BS, bool FUSE_A = false>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(FUSE_A ? 3 : BPA_KLANE_OCC, 8)))
step_s4_klane_v3_kernel(const PlanDev P)
{
  extern __shared__ __attribute__((aligned(16))) double2 s_pmx[];    // [BS/64][rec2_units - 1][64]: every update's two matrices, per wave
  __shared__ double s_term[BS], s_tr[BS];
  __shared__ uint4 s_rec[BS/64][4][16];                               // per wave: the step records of the groups it spans
  __shared__ double2 s_par[BS/64][4][16];                             // per wave and group: what the root term and the site term read of the parameter block (round 6)
  __shared__ double2 s_ring[2][2][BS];                                // per lane: the parents of the updates two and three back (round 6)
  const uint32_t b = P.blk0 + blockIdx.x, lane = threadIdx.x, gl = b*BS + lane;
  if (FUSE_A && (P.flags & 32u))
  {
    const uint32_t q0 = P.blk_slot_off[b], q1 = P.blk_slot_off[b+1];
    if (lane < q1 - q0)
    {
      const LocusDev & L = P.loci[P.slot_tab[q0 + lane].locus];
      // (a JC69 locus — the other loci of a composite's packing — has no eigensystem.  NOT "only the loci with a record in this
      //  step": a locus whose rejected parameter proposal was just rolled back may sit this step out, and the refresh is owed once)
      const bool mine = L.model != 0u;
      for (uint32_t m = 0; mine && m < L.rate_matrices; ++m)
      {
        double * pm = L.par + par_matrix(L.rate_cats, 4, m);
        update_eigen_regs<4>(pm + pm_freqs(4), pm + pm_subst(4), pm + pm_evals(4), pm + pm_evecs(4), pm + pm_ievecs(4));
      }
    }
    __syncthreads();
  }
  if (FUSE_A && (P.flags & 1u))
  {
    const uint32_t e0 = P.blk_mat_off[b], e1 = P.blk_mat_off[b+1], rmax = P.pad;
    for (uint32_t i = lane; i < (e1 - e0)*rmax; i += BS)
    {
      const uint32_t e = e0 + i/rmax, k = i % rmax;
      const MatRec2 m2 = P.mat2[e];
      if (m2.slot == 0xffffffffu) continue;
      const SlotStatic & M = P.slot_tab[m2.slot];
      if (k >= M.rate_cats) continue;
      MatRec m;
      m.dst = M.pmat + (size_t)m2.pmatrix*M.rate_cats*M.pstride; m.par = M.par; m.rate_cats = M.rate_cats; m.model = M.model; m.entry = e; m.pad = 0;
      pmatrix_s4_rec(m, P.mat_length, k);
    }
    __syncthreads();                                         // the block's fresh P-matrices are out (written and read on this CU)
  }
  BPA_STAMP_NW(P, b, lane, 0);
  const uint32_t s0 = P.blk_slot_off[b], s1 = P.blk_slot_off[b+1];
  const LaneStatic ls = P.lane_tab[gl];
  const bool has_slot = ls.slot != 0xffffffffu;
  const bool summer = (P.flags & 4u) && lane < s1 - s0;
  const uint32_t n = ls.n_np_tips & 511u, np = (ls.n_np_tips >> 9) & 511u;
  const uint32_t k = (ls.n_np_tips >> 23) & 7u, R = ls.n_np_tips >> 26;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(lane >> 6), wl = lane & 63u;
  const uint32_t units = P.rec2_units;
  static const uint32_t none = 0xffffffffu;

  // ---- the groups of this wave (a group = the lanes of one (locus, category)), from the lane table alone
  const bool grouped = has_slot && (P.flags & 2u);
  const uint32_t gkey = grouped ? (ls.slot << 3 | k) : none;
  const uint32_t prevkey = __shfl_up(gkey, 1);
  const bool first = grouped && (wl == 0 || gkey != prevkey);
  const unsigned long long fmask = __ballot(first);
  const uint32_t ngroups = (uint32_t)__popcll(fmask);
  const uint32_t my_g = (uint32_t)__popcll(fmask & ((2ull << wl) - 1ull)) - 1u;
  const uint32_t gs = wl >> 4, gu = wl & 15u;
  uint32_t src = 0;
  { unsigned long long m = fmask; for (uint32_t i = 0; i < gs; ++i) m &= m - 1ull; src = m ? (uint32_t)__ffsll((long long)m) - 1u : 0u; }
  const uint32_t s_slot = __shfl(ls.slot, (int)src), s_kR = __shfl(k | R << 3, (int)src);
  const bool stager = ngroups >= 1 && ngroups <= 4 && gs < ngroups;

  BPA_STAMP_NW(P, b, lane, 1);
  // ---- trip 2: slot record, header, the step's records of the wave's groups, the summing lanes' task
  uint32_t c_np = 0, c_l0 = 0, c_task = none;
  double c_lnl = 0;
  if (summer)
  {
    const SlotStatic & C = P.slot_tab[s0 + lane];
    c_np = C.np; c_l0 = C.lane0 - b*BS;
    c_task = reinterpret_cast<const StepRec *>(P.recs2 + (size_t)(s0 + lane)*units)->task;
  }
  SlotStatic S{};
  StepRec hdr{};
  hdr.task = none;
  const uint4 * rp = P.recs2 + (size_t)(has_slot ? ls.slot : 0u)*units;
  uint4 st_unit = make_uint4(none, 0, 0, 0);
  if (stager && gu < units) st_unit = gld4(P.recs2 + (size_t)s_slot*units + gu);
  if (has_slot && (P.flags & 6u))
  {
    const uint4 * sp = reinterpret_cast<const uint4 *>(P.slot_tab + ls.slot);
    uint4 * sd = reinterpret_cast<uint4 *>(&S);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(SlotStatic)/16); ++i) sd[i] = sp[i];
    *reinterpret_cast<uint4 *>(&hdr) = rp[0];
  }
  if (stager) s_rec[wave][gs][gu] = st_unit;
  const bool work = has_slot && hdr.task != none && (P.flags & 6u);
  const uint32_t tips = S.tips_n;
  const unsigned long long odd = __ballot(grouped && work && S.pstride != 16u);      // (a, b) pairs of JC69 loci: nothing to stage
  const bool use_lds = ngroups >= 1 && ngroups <= 4 && odd == 0 && !getenv_klane_direct(P);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  BPA_STAMP_NW(P, b, lane, 2);
  // ---- trip 3, part 1: every update's two matrices, global -> LDS (lane 16 g + p brings chunk p ^ g of group g)
  // (the matrices of CH updates at a time — CH = 2, measured on config 3's full set: 87 us; CH = 4: 96 us with 10 spilled registers at
  //  the 128 of four waves per SIMD; everything at once, round 5's first form: 90 us, 38 KB of LDS per workgroup; five or six waves per
  //  SIMD spill 30-86 registers: 139-199 us — tools/ab_klane_occ.sh)
  constexpr int CH = BPA_KLANE_CH;
  double2 * pmw = s_pmx + (size_t)wave*CH*64u;
  const double * st_pmat = nullptr;
  uint32_t st_k = 0, st_R = 0, st_nops = 0;
  const uint32_t st_c = gu ^ gs;
  if (use_lds)
  {
    const unsigned long long s_pm_lo = __shfl((uint32_t)reinterpret_cast<uintptr_t>(S.pmat), (int)src);
    const unsigned long long s_pm_hi = __shfl((uint32_t)(reinterpret_cast<uintptr_t>(S.pmat) >> 32), (int)src);
    st_pmat = reinterpret_cast<const double *>(s_pm_lo | s_pm_hi << 32);
    st_k = s_kR & 7u; st_R = s_kR >> 3;
    if (stager)
    {
      const StepRec sh = *reinterpret_cast<const StepRec *>(&s_rec[wave][gs][0]);
      st_nops = sh.task != none ? sh.nops : 0u;
    }
  }
  // ---- (round 6) what the root term and the site term read of the locus's parameter block — matrix 0's frequencies, the
  // categories' matrix indices, the category weights: 10 sixteen-byte units per group — is requested HERE, with trip 3, by the
  // group's staging lanes and handed over through LDS.  Before, the root term began with a dependent round trip (index, then
  // frequencies) and the site term with another (the weights): ~1.5 of a workgroup's 11 us (profiles/r5/probe_klane.txt).
  // Global -> LDS directly (lds_dma16: lane 16 g + u lands at unit u of group g's corner, no registers); the units are 16-byte
  // aligned when the category count is even (the block begins rates[R] | weights[R] | indices[R] | matrix 0), else the old way.
  const bool have_par = use_lds && (P.flags & 2u) && __ballot(grouped && (R & 1u)) == 0ull;
  if (have_par)                                    // (wave-uniform: every lane takes part in the shuffles)
  {
    const unsigned long long s_pa_lo = __shfl((uint32_t)reinterpret_cast<uintptr_t>(S.par), (int)src);
    const unsigned long long s_pa_hi = __shfl((uint32_t)(reinterpret_cast<uintptr_t>(S.par) >> 32), (int)src);
    const double * st_par = reinterpret_cast<const double *>(s_pa_lo | s_pa_hi << 32);
    if (stager && st_nops != 0u && gu < 10u)
    {
      const uint32_t off = gu < 2u ? par_matrix(st_R, 4, 0) + pm_freqs(4) + 2u*gu : gu < 6u ? par_param_idx(st_R) + 2u*(gu - 2u) : par_rate_weights(st_R) + 2u*(gu - 6u);
      lds_dma16(st_par + off, &s_par[wave][0][0]);
    }
  }

  double tr = 0;
  {
    double fwd[4] = {0, 0, 0, 0};
    uint32_t fwd_clv = none;
    typedef double d2v __attribute__((ext_vector_type(2)));
    auto clv_ptr = [&](uint32_t c)
    {
      return reinterpret_cast<const __attribute__((address_space(1))) d2v *>(
          reinterpret_cast<uintptr_t>(S.clv + ((((size_t)(c - tips)*R) + k)*np + n)*4));
    };
    auto code_of = [&](uint32_t c) { return tips <= 8 ? (ls.tipcodes >> (4*c)) & 15u : (uint32_t)S.tips[(size_t)c*np + n]; };
    // (round 6) flags bit 11: the step's LAST update is not stored when it makes the root's CLV — nothing reads a root's
    // buffer but the root term below, which has it in registers; the device samplers set it (gsampler_host.hpp: gs_eval) and
    // bring the buffers level before anything else may read them (gs_download).  A third of a step's parent stores.
    const bool skip_root = (P.flags & 2048u) != 0;
    auto store_parent = [&](uint32_t pc, const double x[4], const double y[4], const bool keep)
    {
      double2 o0, o1;
      o0.x = x[0]*y[0]; o0.y = x[1]*y[1]; o1.x = x[2]*y[2]; o1.y = x[3]*y[3];
      d2v a0, a1; a0.x = o0.x; a0.y = o0.y; a1.x = o1.x; a1.y = o1.y;
      __attribute__((address_space(1))) d2v * dst = reinterpret_cast<__attribute__((address_space(1))) d2v *>(
          reinterpret_cast<uintptr_t>(S.clv + ((((size_t)(pc - tips)*R) + k)*np + n)*4));
      if (keep) { __builtin_nontemporal_store(a0, dst); __builtin_nontemporal_store(a1, dst + 1); }
      fwd[0] = o0.x; fwd[1] = o0.y; fwd[2] = o1.x; fwd[3] = o1.y;
      fwd_clv = pc;
    };
    const uint32_t nops = (work && (P.flags & 2u)) ? hdr.nops : 0u;
    if (use_lds)
    {
      uint32_t written = 0;                                    // the CLV buffers (< 32: byte indices of <= 8-tip loci, 5 bits) this lane's step has written
      bool dma_waited = false;
      // (round 6) the parents of the last three updates stay with the lane: the last one in registers (fwd, as before), the two
      // before it in the lane's own LDS words (s_ring[update & 1]; hist: their buffer indices, a byte each).  A prune-and-regraft
      // step walks TWO root paths whose updates alternate in age order, so a child is often the parent of the update before
      // the last: read back from HBM "when used, after the lane's own store" that was a store -> load round trip (~2 us) in
      // one locus-step in four (config 3; tools/opstat.py: distances 1 / 2 / 3 / more = 52 752 / 6 665 / 2 313 / 1 172).
      uint32_t hist = 0xffffffu;
      for (uint32_t o0 = 0; __any(o0 < nops); o0 += (uint32_t)CH)
      {
        // ---- trip 3, part 1 (once per CH updates): these updates' matrices, global -> LDS
        // (lane 16 g + p brings chunk p ^ g of group g; the reads of the previous four are behind this wave in program order)
#pragma unroll
        for (int j = 0; j < CH; ++j)
          if (o0 + j < st_nops)
          {
            const StepOp so = *reinterpret_cast<const StepOp *>(&s_rec[wave][gs][1 + o0 + j]);
            const uint32_t pm = st_c < 8u ? so.left_pmatrix : so.right_pmatrix;
            lds_dma16(st_pmat + ((size_t)pm*st_R + st_k)*16 + (size_t)(st_c & 7u)*2, pmw + (size_t)j*64u);
          }
        // ---- trip 3, part 2: their children that can be read ahead
        uint2 opw[CH];
        d2v ch[CH][2], chb[2];
        uint32_t pf[CH];
        uint32_t w2 = written, prevp = fwd_clv;
#pragma unroll
        for (int j = 0; j < CH; ++j)
        {
          pf[j] = 0; opw[j] = make_uint2(0, 0);
          if (o0 + j < nops)
          {
            opw[j] = *reinterpret_cast<const uint2 *>(&s_rec[wave][my_g][1 + o0 + j]);
            const uint32_t pc = opw[j].x & 255u, lc = (opw[j].x >> 8) & 255u, rc = (opw[j].x >> 16) & 255u;
            const bool pl = lc >= tips && lc != prevp && !((w2 >> (lc & 31u)) & 1u);
            const bool pr = rc >= tips && rc != prevp && !((w2 >> (rc & 31u)) & 1u);
            if (pl)
            {
              const auto p = clv_ptr(lc);
              ch[j][0] = __builtin_nontemporal_load(p); ch[j][1] = __builtin_nontemporal_load(p + 1);
              pf[j] = 1;
              if (j == 0 && pr) { const auto q = clv_ptr(rc); chb[0] = __builtin_nontemporal_load(q); chb[1] = __builtin_nontemporal_load(q + 1); pf[j] = 3; }
            }
            else if (pr)
            {
              const auto p = clv_ptr(rc);
              ch[j][0] = __builtin_nontemporal_load(p); ch[j][1] = __builtin_nontemporal_load(p + 1);
              pf[j] = 2;
            }
            w2 |= 1u << (pc & 31u);
            prevp = pc;
          }
        }
        // the matrices have landed (this wave's own requests: vmcnt; the children read ahead come back with them).  The wait is
        // made HERE, once per CH updates, and shown to the compiler (the empty statements read-modify every register read
        // ahead): left to its wait-count pass the first use of a child in update j is an s_waitcnt vmcnt(0) at a branch merge,
        // i.e. every update would wait for the previous update's STORES to be acknowledged
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < CH; ++j) { asm volatile("" : "+v"(ch[j][0])); asm volatile("" : "+v"(ch[j][1])); }
        asm volatile("" : "+v"(chb[0])); asm volatile("" : "+v"(chb[1]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!dma_waited) { dma_waited = true; BPA_STAMP_NW(P, b, lane, 3); }
#pragma unroll
        for (int j = 0; j < CH; ++j)
        {
          if (o0 + j < nops)
          {
            const uint32_t pc = opw[j].x & 255u, lc = (opw[j].x >> 8) & 255u, rc = (opw[j].x >> 16) & 255u;
            const uint32_t oi = o0 + (uint32_t)j;
            // the parents two / three updates back and where they lie (three back shares the LDS word of the last one, which
            // is still in registers and moves out below, after this update's reads)
            const uint32_t h2 = (hist >> 8) & 255u, h3 = (hist >> 16) & 255u;
            double lv[4], rv[4], x[4], y[4];
            if (lc == fwd_clv) { lv[0] = fwd[0]; lv[1] = fwd[1]; lv[2] = fwd[2]; lv[3] = fwd[3]; }
            else if (lc < tips) expand_code(code_of(lc), lv);
            else if (lc == h2 || lc == h3)
            {
              const uint32_t lvl = lc == h2 ? (oi & 1u) : ((oi + 1u) & 1u);
              const double2 uu = s_ring[lvl][0][lane], ww = s_ring[lvl][1][lane];
              lv[0] = uu.x; lv[1] = uu.y; lv[2] = ww.x; lv[3] = ww.y;
            }
            else
            {
              d2v uu, ww;
              if (pf[j] & 1u) { uu = ch[j][0]; ww = ch[j][1]; }
              else { const auto p = clv_ptr(lc); uu = __builtin_nontemporal_load(p); ww = __builtin_nontemporal_load(p + 1);
                     asm volatile("" : "+v"(uu)); asm volatile("" : "+v"(ww)); }      // (a read at use: waited for inside its branch)
              lv[0] = uu.x; lv[1] = uu.y; lv[2] = ww.x; lv[3] = ww.y;
            }
            if (rc == fwd_clv) { rv[0] = fwd[0]; rv[1] = fwd[1]; rv[2] = fwd[2]; rv[3] = fwd[3]; }
            else if (rc < tips) expand_code(code_of(rc), rv);
            else if (rc == h2 || rc == h3)
            {
              const uint32_t lvl = rc == h2 ? (oi & 1u) : ((oi + 1u) & 1u);
              const double2 uu = s_ring[lvl][0][lane], ww = s_ring[lvl][1][lane];
              rv[0] = uu.x; rv[1] = uu.y; rv[2] = ww.x; rv[3] = ww.y;
            }
            else
            {
              d2v uu, ww;
              if (pf[j] == 2u) { uu = ch[j][0]; ww = ch[j][1]; }
              else if (j == 0 && pf[j] == 3u) { uu = chb[0]; ww = chb[1]; }
              else { const auto p = clv_ptr(rc); uu = __builtin_nontemporal_load(p); ww = __builtin_nontemporal_load(p + 1);
                     asm volatile("" : "+v"(uu)); asm volatile("" : "+v"(ww)); }
              rv[0] = uu.x; rv[1] = uu.y; rv[2] = ww.x; rv[3] = ww.y;
            }
            // the last update's parent leaves the registers for its LDS word (level (oi - 1) & 1: the update three back has been
            // read if it was wanted).  HERE, between the children and the arithmetic: at the end of the update, with x and y
            // live, the same two writes cost 200 spilled registers
            if (oi != 0u)
            {
              double2 m0, m1; m0.x = fwd[0]; m0.y = fwd[1]; m1.x = fwd[2]; m1.y = fwd[3];
              s_ring[(oi + 1u) & 1u][0][lane] = m0; s_ring[(oi + 1u) & 1u][1][lane] = m1;
            }
            const double2 * r = pmw + (size_t)j*64u + my_g*16u;
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
              const double2 a0 = r[(2*i) ^ my_g], b0 = r[(2*i + 1) ^ my_g];
              x[i] = dot4_pair(a0.x, a0.y, b0.x, b0.y, lv);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
              const double2 a1 = r[(8 + 2*i) ^ my_g], b1 = r[(8 + 2*i + 1) ^ my_g];
              y[i] = dot4_pair(a1.x, a1.y, b1.x, b1.y, rv);
            }
            store_parent(pc, x, y, !(skip_root && oi + 1u == nops && pc == (uint32_t)hdr.root_clv));
            hist = (hist << 8) | pc;
            written |= 1u << (pc & 31u);
          }
        }
      }
    }
    else
    {
      // more than four groups in the wave (loci of a handful of patterns) or (a, b) pairs: v2's direct form, a record at a time
      for (uint32_t o = 0; __any(o < nops); ++o)
      {
        if (o < nops)
        {
          StepOp op;
          *reinterpret_cast<uint4 *>(&op) = gld4(rp + 1 + o);
          double lv[4], rv[4], x[4], y[4];
          auto vec_of = [&](uint32_t c, double v[4])
          {
            if (c == fwd_clv) { v[0] = fwd[0]; v[1] = fwd[1]; v[2] = fwd[2]; v[3] = fwd[3]; }
            else if (c < tips) expand_code(code_of(c), v);
            else { const auto p = clv_ptr(c); const d2v uu = __builtin_nontemporal_load(p), ww = __builtin_nontemporal_load(p + 1); v[0] = uu.x; v[1] = uu.y; v[2] = ww.x; v[3] = ww.y; }
          };
          vec_of(op.left_clv, lv);
          vec_of(op.right_clv, rv);
          matvec4_p(S.pmat, S.pstride, (size_t)op.left_pmatrix*R  + k, lv, x);
          matvec4_p(S.pmat, S.pstride, (size_t)op.right_pmatrix*R + k, rv, y);
          store_parent(op.parent_clv, x, y, !(skip_root && o + 1u == nops && op.parent_clv == hdr.root_clv));
        }
      }
    }
    BPA_STAMP_NW(P, b, lane, 4);
    if (work)
    {
      // K2 at the root (core_likelihood_avx.c:117-150): this category's frequency-weighted sum
      const double * par = S.par;
      double c[4];
      const uint32_t rc = hdr.root_clv;
      if (rc == fwd_clv) { c[0] = fwd[0]; c[1] = fwd[1]; c[2] = fwd[2]; c[3] = fwd[3]; }
      else if (rc < tips) expand_code(code_of(rc), c);
      else { const auto p = clv_ptr(rc); const d2v uu = __builtin_nontemporal_load(p), ww = __builtin_nontemporal_load(p + 1); c[0] = uu.x; c[1] = uu.y; c[2] = ww.x; c[3] = ww.y; }
      // (the category's matrix index and — in the same round trip — matrix 0's frequencies, which it names nearly always;
      //  round 6: both out of the group's LDS corner, requested with the step's matrices)
      double md, fr[4];
      if (have_par && nops != 0u)
      {
        const double2 mi = s_par[wave][my_g][2u + (k >> 1)], fa = s_par[wave][my_g][0], fb = s_par[wave][my_g][1];
        md = (k & 1u) ? mi.y : mi.x; fr[0] = fa.x; fr[1] = fa.y; fr[2] = fb.x; fr[3] = fb.y;
      }
      else
      {
        md = par[par_param_idx(R) + k];
        const double * f0 = par + par_matrix(R, 4, 0) + pm_freqs(4);
        fr[0] = f0[0]; fr[1] = f0[1]; fr[2] = f0[2]; fr[3] = f0[3];
      }
      const uint32_t m = (uint32_t)md;
      if (m != 0u) { const double * f = par + par_matrix(R, 4, m) + pm_freqs(4); fr[0] = f[0]; fr[1] = f[1]; fr[2] = f[2]; fr[3] = f[3]; }
      tr = dot4_pair(fr[0], fr[1], fr[2], fr[3], c);
    }
  }
  s_tr[lane] = tr;
  lds_barrier();                                   // (LDS only: a plain __syncthreads() would wait for this wave's CLV stores to be acknowledged)
  BPA_STAMP_NW(P, b, lane, 5);
  double term = 0;
  if (work && k == 0)
  {
    // (the category weights in ONE round trip: a load per turn of a loop with a run-time bound waits for each in turn)
    const double * par = S.par;
    double rwv[8];
    if (have_par && hdr.nops != 0u)
    {
#pragma unroll
      for (int q = 0; q < 8; q += 2) { const double2 w2 = s_par[wave][my_g][6 + (q >> 1)]; rwv[q] = w2.x; rwv[q + 1] = w2.y; }
    }
    else
    {
#pragma unroll
      for (int q = 0; q < 8; ++q) rwv[q] = (uint32_t)q < R ? par[par_rate_weights(R) + q] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) if ((uint32_t)q < R) term += s_tr[lane + q*np]*rwv[q];
    term = log(term)*ls.wgt;
    gst(P.site_term + hdr.pat_off + n, term);
  }
  if (!(P.flags & 4u)) return;

  // ---- per-locus sum in pattern order (the k = 0 lanes are the first np lanes of a locus)
  s_term[lane] = term;
  lds_barrier();
  BPA_STAMP_NW(P, b, lane, 6);
  if (summer && c_task != none)
  {
    // (pattern order, the reference's sequential sum; the terms come out of LDS eight at a time — one at a time every add
    //  waited for its own read: ~100 cycles x np)
    double logl = 0;
    uint32_t q = 0;
    for (; q + 8u <= c_np; q += 8u)
    {
      double t8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t8[j] = s_term[c_l0 + q + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) logl += t8[j];
    }
    for (; q < c_np; ++q) logl += s_term[c_l0 + q];
    P.lnl[c_task] = P.bfbeta*logl;
    c_lnl = P.bfbeta*logl;
  }
  if (P.flags & 8u)
  {
    __shared__ double s_lnl[BS];
    s_lnl[lane] = c_lnl;
    lds_barrier();
    if (lane == 0)
    {
      double part = 0;
      for (uint32_t q = 0; q < s1 - s0; ++q) part += s_lnl[q];
      P.wg_part[b] = part;
    }
  }
  BPA_STAMP_NW(P, b, lane, 7);
}

// pmatrix_wg_kernel with its serial chain cut down the way partials_lnl_pipe20_kernel's was: the wave-uniform chain
// (branch entry -> task -> locus record -> parameter block -> matrix index) runs through the scalar data path, the two
// eigenvector matrices go global -> LDS directly (no registers, no ds_write), barriers order LDS only and the output
// stores are not waited for.  Same arithmetic, same bits.
template <int S>
__global__ void __launch_bounds__(256) pmatrix_wg2_kernel(const PlanDev P)
{
  static_assert((S*S*8) % 16 == 0, "16-byte staging units");
  __shared__ __attribute__((aligned(16))) double s_evs[2*S*S], s_tmp[4][S*S], s_e[4][S];
  double * const s_ev = s_evs, * const s_iev = s_evs + S*S;
  const uint32_t e = blockIdx.x + ((P.flags & 256u) ? P.ent0 : 0u), tid = threadIdx.x, lane = tid & 63u;      // (bit 8: a half-batch's entries)
  const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (((cu32_p)P.mat_task)[e] == 0xffffffffu) return;            // a hole of a device-written step (gsampler.hpp)
  const uint32_t lid = ((cu32_p)P.task_locus)[((cu32_p)P.mat_task)[e]];
  cu64_p L64 = (cu64_p)(P.loci + lid);
  cu32_p L32 = (cu32_p)(P.loci + lid);
  const uint32_t R = L32[20];
  const cdbl4_p par = (cdbl4_p)L64[5];
  const double * parg = (const double *)L64[5];
  const double t = ((cdbl4_p)P.mat_length)[e];
  const gdbl_p pbase = (gdbl_p)L64[1] + (size_t)((cu32_p)P.mat_pmatrix)[e]*R*S*S;
  for (uint32_t k0 = 0; k0 < R; )
  {
    const uint32_t m = (uint32_t)par[par_param_idx(R) + k0];
    uint32_t k1 = k0 + 1;
    while (k1 < R && k1 - k0 < 4 && (uint32_t)par[par_param_idx(R) + k1] == m) ++k1;
    const uint32_t nk = k1 - k0;
    const uint32_t pmo = par_matrix(R, S, m);
    if (k0) lds_barrier();                                     // the previous group's LDS reads are done
    {
      // evecs | ievecs are adjacent in the parameter block: 2*S*S doubles = S*S 16-byte units, 64 per instruction
      const double * src = parg + pmo + pm_evecs(S);
      constexpr uint32_t total = S*S;                          // 16-byte units
      if (((pmo + pm_evecs(S)) & 1u) == 0)                    // 16-byte aligned in the parameter block (R even)
        for (uint32_t c = w; c*64 < total; c += 4)
        {
          const uint32_t idx = c*64 + lane;
          if (idx < total)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 2*(size_t)idx),
                                             (__attribute__((address_space(3))) void *)(s_evs + (size_t)c*128), 16, 0, 0);
        }
      else
        for (uint32_t i = tid; i < 2*S*S; i += 256) s_evs[i] = ((gcdbl_p)src)[i];
    }
    if (tid < nk*S)
    {
      const uint32_t k = k0 + tid/S, mm = tid % S;
      s_e[tid/S][mm] = expm1(((gcdbl_p)parg)[pmo + pm_evals(S) + mm]*(t*((gcdbl_p)parg)[par_rates(R) + k]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    // temp = inv_eigenvecs * expd (core_pmatrix.c:741-747), once per element
    for (uint32_t i = tid; i < nk*S*S; i += 256) s_tmp[i/(S*S)][i % (S*S)] = s_iev[i % (S*S)]*s_e[i/(S*S)][i % S];
    lds_barrier();
    // pmat = I + temp * eigenvecs (core_pmatrix.c:749-756): one lane = 4 consecutive columns of a row
    constexpr uint32_t Q = S/4;
    for (uint32_t idx = tid; idx < nk*S*Q; idx += 256)
    {
      const uint32_t kk = idx/(S*Q), j = (idx % (S*Q))/Q, c0 = 4*(idx % Q), k = k0 + kk;
      double acc[4] = {j == c0 ? 1.0 : 0.0, j == c0 + 1 ? 1.0 : 0.0, j == c0 + 2 ? 1.0 : 0.0, j == c0 + 3 ? 1.0 : 0.0};
      if (!(t*par[par_rates(R) + k] < 1e-100))
      {
        const double * tr = &s_tmp[kk][j*S];
#pragma unroll
        for (int mm = 0; mm < S; ++mm)
        {
          const double tv = tr[mm];
          const double2 * ev = reinterpret_cast<const double2 *>(&s_ev[mm*S + c0]);
          const double2 a = ev[0], c = ev[1];
          acc[0] += tv*a.x; acc[1] += tv*a.y; acc[2] += tv*c.x; acc[3] += tv*c.y;
        }
      }
      typedef double d2v __attribute__((ext_vector_type(2)));
      d2v o0, o1; o0.x = acc[0]; o0.y = acc[1]; o1.x = acc[2]; o1.y = acc[3];
      __attribute__((address_space(1))) d2v * dst = (__attribute__((address_space(1))) d2v *)(pbase + (size_t)k*S*S + j*S + c0);
      dst[0] = o0; dst[1] = o1;
    }
    k0 = k1;
  }
}

// pmatrix_wg2_kernel for step images that keep a locus's branch entries together (the generic sampler's: entries
// [i*group, (i + 1)*group) are locus i's, unused ones holes): ONE workgroup per locus.  The wave-uniform chain (entry -> task ->
// locus record -> parameter block) and the two eigenvector matrices' trip into LDS are paid once per locus instead of once per
// branch — they are most of a workgroup's life in pmatrix_wg2_kernel (config 4: 3 fresh branches per locus and step) —, the
// eigenvector matrices come through lds_dma16 (no s_waitcnt vmcnt(0) of the compiler's in front of the LDS reads: an entry's
// output stores are never waited for).  Same arithmetic per element, same bits.
template <int S>
__global__ void __launch_bounds__(256) pmatrix_wg2_group_kernel(const PlanDev P, const uint32_t group)
{
  static_assert((S*S*8) % 16 == 0, "16-byte staging units");
  __shared__ __attribute__((aligned(16))) double s_evs[2*S*S], s_tmp[4][S*S], s_e[4][S];
  double * const s_ev = s_evs, * const s_iev = s_evs + S*S;
  const uint32_t e0 = ((P.flags & 256u) ? P.ent0 : 0u) + blockIdx.x*group, tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t task = 0xffffffffu;
  for (uint32_t q = 0; q < group && task == 0xffffffffu; ++q) task = ((cu32_p)P.mat_task)[e0 + q];
  if (task == 0xffffffffu) return;                              // the locus proposed nothing this step
  const uint32_t lid = ((cu32_p)P.task_locus)[task];
  cu64_p L64 = (cu64_p)(P.loci + lid);
  cu32_p L32 = (cu32_p)(P.loci + lid);
  const uint32_t R = L32[20];
  const cdbl4_p par = (cdbl4_p)L64[5];
  const double * parg = (const double *)L64[5];
  const gdbl_p pmat = (gdbl_p)L64[1];
  for (uint32_t k0 = 0; k0 < R; )
  {
    const uint32_t m = (uint32_t)par[par_param_idx(R) + k0];
    uint32_t k1 = k0 + 1;
    while (k1 < R && k1 - k0 < 4 && (uint32_t)par[par_param_idx(R) + k1] == m) ++k1;
    const uint32_t nk = k1 - k0;
    const uint32_t pmo = par_matrix(R, S, m);
    if (k0) lds_barrier();                                     // the previous group's LDS reads are done
    {
      const double * src = parg + pmo + pm_evecs(S);
      constexpr uint32_t total = S*S;                          // 16-byte units
      if (((pmo + pm_evecs(S)) & 1u) == 0)                    // 16-byte aligned in the parameter block (R even)
        for (uint32_t c = w; c*64 < total; c += 4)
        {
          const uint32_t idx = c*64 + lane;
          if (idx < total) lds_dma16(src + 2*(size_t)idx, s_evs + (size_t)c*128);
        }
      else
        for (uint32_t i = tid; i < 2*S*S; i += 256) s_evs[i] = ((gcdbl_p)src)[i];
    }
    bool first = true;
    for (uint32_t q = 0; q < group; ++q)
    {
      const uint32_t e = e0 + q;
      if (((cu32_p)P.mat_task)[e] != task) continue;           // a hole (or, in a foreign layout, another locus's entry: not this kernel's)
      const double t = ((cdbl4_p)P.mat_length)[e];
      const gdbl_p pbase = pmat + (size_t)((cu32_p)P.mat_pmatrix)[e]*R*S*S;
      // (s_e is read between the two barriers below only, s_tmp after the second: the next entry's writes of s_e may start
      //  while other waves still multiply, its writes of s_tmp come after its first barrier)
      if (tid < nk*S)
      {
        const uint32_t k = k0 + tid/S, mm = tid % S;
        s_e[tid/S][mm] = expm1(((gcdbl_p)parg)[pmo + pm_evals(S) + mm]*(t*((gcdbl_p)parg)[par_rates(R) + k]));
      }
      if (first) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); first = false; }      // the eigenvector matrices are in LDS
      lds_barrier();
      // temp = inv_eigenvecs * expd (core_pmatrix.c:741-747), once per element
      for (uint32_t i = tid; i < nk*S*S; i += 256) s_tmp[i/(S*S)][i % (S*S)] = s_iev[i % (S*S)]*s_e[i/(S*S)][i % S];
      lds_barrier();
      // pmat = I + temp * eigenvecs (core_pmatrix.c:749-756): one lane = 4 consecutive columns of a row
      constexpr uint32_t Q = S/4;
      for (uint32_t idx = tid; idx < nk*S*Q; idx += 256)
      {
        const uint32_t kk = idx/(S*Q), j = (idx % (S*Q))/Q, c0 = 4*(idx % Q), k = k0 + kk;
        double acc[4] = {j == c0 ? 1.0 : 0.0, j == c0 + 1 ? 1.0 : 0.0, j == c0 + 2 ? 1.0 : 0.0, j == c0 + 3 ? 1.0 : 0.0};
        if (!(t*par[par_rates(R) + k] < 1e-100))
        {
          const double * tr = &s_tmp[kk][j*S];
#pragma unroll
          for (int mm = 0; mm < S; ++mm)
          {
            const double tv = tr[mm];
            const double2 * ev = reinterpret_cast<const double2 *>(&s_ev[mm*S + c0]);
            const double2 a = ev[0], c = ev[1];
            acc[0] += tv*a.x; acc[1] += tv*a.y; acc[2] += tv*c.x; acc[3] += tv*c.y;
          }
        }
        d2v_t o0, o1; o0.x = acc[0]; o0.y = acc[1]; o1.x = acc[2]; o1.y = acc[3];
        __attribute__((address_space(1))) d2v_t * dst = (__attribute__((address_space(1))) d2v_t *)(pbase + (size_t)k*S*S + j*S + c0);
        dst[0] = o0; dst[1] = o1;
      }
    }
    k0 = k1;
  }
}

// pll_core_update_pmatrix (core_pmatrix.c:785) over staged host arrays:
// one lane per (matrix i, rate k, row j).
template <int S>
__global__ void __launch_bounds__(BPA_BLOCK) pmatrix_lib_kernel(double * __restrict__ out, uint32_t count, uint32_t R,
                                                                const double * __restrict__ rates,
                                                                const double * __restrict__ bl,
                                                                const uint32_t * __restrict__ param_idx,
                                                                const double * __restrict__ evals,
                                                                const double * __restrict__ evecs,
                                                                const double * __restrict__ ievecs)
{
  const uint32_t tid = blockIdx.x*BPA_BLOCK + threadIdx.x;
  const uint32_t j = tid % S, ik = tid / S;
  const uint32_t i = ik / R, k = ik % R;
  if (i >= count) return;
  double * prow = out + ((size_t)i*R + k)*S*S + j*S;
  const double t = bl[i];
  if (t == 0.0)
  {
    for (int c = 0; c < S; ++c) prow[c] = ((int)j == c) ? 1.0 : 0.0;
    return;
  }
  const uint32_t m = param_idx[k];
  pmatrix_eigen_row<S>(prow, (int)j, t, rates[k], evals + (size_t)m*S, evecs + (size_t)m*S*S,
                       ievecs + (size_t)m*S*S, true);
}

// ======================================================================= K6 ==
// Symmetrised rate matrix -> Householder tridiagonalisation -> implicit QL, in
// the operation order of core_pmatrix.c:28-297 (so eigenvectors are bit-identical
// to the reference's).  One lane per rate matrix; the matrix lives in that lane's
// private memory (4x4: registers; 20x20: 3.2 KB scratch, run once per AA locus).
//
// PROVENANCE.  eigen_sym / eigen_sym_static below are the textbook tred2 / tqli pair (Householder reduction to
// tridiagonal form, then QL with implicit shifts: Wilkinson & Reinsch's EISPACK routines as printed in Numerical
// Recipes), which is also what the reference carries as mytred2 / mytqli (core_pmatrix.c:28-182).  They are written
// here statement by statement in the reference's operation order, re-indexed to 0-based arrays, ON PURPOSE: SURVEY A.5
// makes only the P-matrices a contract (eigenvalue order and eigenvector signs are free), but an eigensystem equal to
// the reference's to the bit makes every P-matrix of a GTR / amino-acid locus equal to <= 8 ulp and lets the golden
// `eigenvals` / `eigenvecs` vectors be compared with ==.  What is this repo's own around them: the symmetrisation and
// scaling (update_eigen_regs), the compile-time-indexed form that keeps a 4x4 system in registers (eigen_sym_static:
// every loop bound and index a constant, no scratch memory), and the kernels that call them.
template <int N>
__device__ void eigen_sym(double (&a)[N][N], double (&d)[N], double (&e)[N])
{
  // --- tridiagonalise (column-oriented Householder) ---
  for (int i = N - 1; i >= 1; --i)
  {
    const int l = i;
    double h = 0, scale = 0;
    if (l > 1)
    {
      for (int k = 0; k < l; ++k) scale += fabs(a[k][i]);
      if (scale == 0.0)
        e[i] = a[l-1][i];
      else
      {
        for (int k = 0; k < l; ++k) { a[k][i] /= scale; h += a[k][i]*a[k][i]; }
        double f = a[l-1][i];
        double g = (f > 0) ? -sqrt(h) : sqrt(h);
        e[i] = scale*g;
        h -= f*g;
        a[l-1][i] = f - g;
        f = 0.0;
        for (int j = 0; j < l; ++j)
        {
          a[i][j] = a[j][i]/h;
          g = 0.0;
          for (int k = 0; k <= j; ++k)    g += a[k][j]*a[k][i];
          for (int k = j + 1; k < l; ++k) g += a[j][k]*a[k][i];
          e[j] = g/h;
          f += e[j]*a[j][i];
        }
        const double hh = f/(h + h);
        for (int j = 0; j < l; ++j)
        {
          f = a[j][i];
          g = e[j] - hh*f;
          e[j] = g;
          for (int k = 0; k <= j; ++k) a[k][j] -= (f*e[k] + g*a[k][i]);
        }
      }
    }
    else
      e[i] = a[l-1][i];
    d[i] = h;
  }
  d[0] = 0.0; e[0] = 0.0;
  for (int i = 0; i < N; ++i)
  {
    const int l = i;
    if (d[i] != 0.0)
      for (int j = 0; j < l; ++j)
      {
        double g = 0.0;
        for (int k = 0; k < l; ++k) g += a[k][i]*a[j][k];
        for (int k = 0; k < l; ++k) a[j][k] -= g*a[i][k];
      }
    d[i] = a[i][i];
    a[i][i] = 1.0;
    for (int j = 0; j < l; ++j) a[i][j] = a[j][i] = 0.0;
  }
  // --- implicit-shift QL on (d, e), rotating the rows of a ---
  for (int i = 1; i < N; ++i) e[i-1] = e[i];
  e[N-1] = 0.0;
  for (int l = 0; l < N; ++l)
  {
    for (int iter = 0; iter < 60; ++iter)
    {
      int m;
      for (m = l; m + 1 < N; ++m)
      {
        const double dd = fabs(d[m]) + fabs(d[m+1]);
        if (fabs(e[m]) + dd == dd) break;
      }
      if (m == l) break;
      double g = (d[l+1] - d[l])/(2.0*e[l]);
      double r = sqrt(g*g + 1.0);
      g = d[m] - d[l] + e[l]/(g + ((g < 0) ? -fabs(r) : fabs(r)));
      double s = 1.0, c = 1.0, p = 0.0;
      for (int i = m - 1; i >= l; --i)
      {
        double f = s*e[i];
        const double b = c*e[i];
        if (fabs(f) >= fabs(g))
        {
          c = g/f;
          r = sqrt(c*c + 1.0);
          e[i+1] = f*r;
          c *= (s = 1.0/r);
        }
        else
        {
          s = f/g;
          r = sqrt(s*s + 1.0);
          e[i+1] = g*r;
          s *= (c = 1.0/r);
        }
        g = d[i+1] - p;
        r = (d[i] - g)*s + 2.0*c*b;
        p = s*r;
        d[i+1] = g + p;
        g = c*r - b;
        for (int k = 0; k < N; ++k)
        {
          f = a[i+1][k];
          a[i+1][k] = s*a[i][k] + c*f;
          a[i][k]   = c*a[i][k] - s*f;
        }
      }
      d[l] = d[l] - p;
      e[l] = g;
      e[m] = 0.0;
    }
  }
}

// The same operations in the same order with every array index a compile-time constant (all loops unrolled, the QL sweep's
// run-time bounds l <= i < m as predicates, d[m] / e[m] as selects): the 4x4 system then lives in REGISTERS.  With run-time
// indices it was LDS (or scratch) — a round trip per access on a chain of ~600 dependent accesses, 35 us a launch for any
// number of loci; the chain of the arithmetic alone is a quarter of that.
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F && f)
{
  if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
template <int N>
__device__ __forceinline__ void eigen_sym_static(double (&a)[N][N], double (&d)[N], double (&e)[N])
{
  static_for<0, N - 1>([&](auto T_)
  {
    constexpr int i = N - 1 - decltype(T_)::value;
    constexpr int l = i;
    double h = 0, scale = 0;
    if constexpr (l > 1)
    {
#pragma unroll
      for (int k = 0; k < l; ++k) scale += fabs(a[k][i]);
      if (scale == 0.0)
        e[i] = a[l-1][i];
      else
      {
#pragma unroll
        for (int k = 0; k < l; ++k) { a[k][i] /= scale; h += a[k][i]*a[k][i]; }
        double f = a[l-1][i];
        double g = (f > 0) ? -sqrt(h) : sqrt(h);
        e[i] = scale*g;
        h -= f*g;
        a[l-1][i] = f - g;
        f = 0.0;
#pragma unroll
        for (int j = 0; j < l; ++j)
        {
          a[i][j] = a[j][i]/h;
          g = 0.0;
#pragma unroll
          for (int k = 0; k <= j; ++k)    g += a[k][j]*a[k][i];
#pragma unroll
          for (int k = j + 1; k < l; ++k) g += a[j][k]*a[k][i];
          e[j] = g/h;
          f += e[j]*a[j][i];
        }
        const double hh = f/(h + h);
#pragma unroll
        for (int j = 0; j < l; ++j)
        {
          f = a[j][i];
          g = e[j] - hh*f;
          e[j] = g;
#pragma unroll
          for (int k = 0; k <= j; ++k) a[k][j] -= (f*e[k] + g*a[k][i]);
        }
      }
    }
    else
      e[i] = a[l-1][i];
    d[i] = h;
  });
  d[0] = 0.0; e[0] = 0.0;
  static_for<0, N>([&](auto I_)
  {
    constexpr int i = decltype(I_)::value;
    constexpr int l = i;
    if (d[i] != 0.0)
    {
#pragma unroll
      for (int j = 0; j < l; ++j)
      {
        double g = 0.0;
#pragma unroll
        for (int k = 0; k < l; ++k) g += a[k][i]*a[j][k];
#pragma unroll
        for (int k = 0; k < l; ++k) a[j][k] -= g*a[i][k];
      }
    }
    d[i] = a[i][i];
    a[i][i] = 1.0;
#pragma unroll
    for (int j = 0; j < l; ++j) a[i][j] = a[j][i] = 0.0;
  });
#pragma unroll
  for (int i = 1; i < N; ++i) e[i-1] = e[i];
  e[N-1] = 0.0;
  static_for<0, N - 1>([&](auto L_)               // (l = N - 1: nothing to do, m = l at once)
  {
    constexpr int l = decltype(L_)::value;
    for (int iter = 0; iter < 60; ++iter)
    {
      int m = N - 1;
      bool found = false;
#pragma unroll
      for (int mm = 0; mm + 1 < N; ++mm)
        if (mm >= l && !found)
        {
          const double dd = fabs(d[mm]) + fabs(d[mm+1]);
          if (fabs(e[mm]) + dd == dd) { m = mm; found = true; }
        }
      if (m == l) break;
      double g = (d[l+1] - d[l])/(2.0*e[l]);
      double r = sqrt(g*g + 1.0);
      double dm = d[N-1];
#pragma unroll
      for (int mm = 0; mm + 1 < N; ++mm) if (mm == m) dm = d[mm];
      g = dm - d[l] + e[l]/(g + ((g < 0) ? -fabs(r) : fabs(r)));
      double s = 1.0, c = 1.0, p = 0.0;
#pragma unroll
      for (int i = N - 2; i >= 0; --i)
        if (i <= m - 1 && i >= l)
        {
          double f = s*e[i];
          const double b = c*e[i];
          if (fabs(f) >= fabs(g))
          {
            c = g/f;
            r = sqrt(c*c + 1.0);
            e[i+1] = f*r;
            c *= (s = 1.0/r);
          }
          else
          {
            s = f/g;
            r = sqrt(s*s + 1.0);
            e[i+1] = g*r;
            s *= (c = 1.0/r);
          }
          g = d[i+1] - p;
          r = (d[i] - g)*s + 2.0*c*b;
          p = s*r;
          d[i+1] = g + p;
          g = c*r - b;
#pragma unroll
          for (int k = 0; k < N; ++k)
          {
            f = a[i+1][k];
            a[i+1][k] = s*a[i][k] + c*f;
            a[i][k]   = c*a[i][k] - s*f;
          }
        }
      d[l] = d[l] - p;
      e[l] = g;
#pragma unroll
      for (int mm = 0; mm < N; ++mm) if (mm == m) e[mm] = 0.0;
    }
  });
}

// freqs/subst -> eigenvals, eigenvecs (u_m[k] sqrt(pi_k)), inv_eigenvecs (u_m[j]/sqrt(pi_j))
// work space of one lane's eigensystem: the solver indexes it with run-time indices, so as private arrays it lives in scratch
// memory (a round trip through the memory hierarchy per access); the 4-state kernels keep one per lane in LDS instead
template <int N> struct EigenWork { double a[N][N], d[N], e[N]; double pad_; };
// the 4-state system in registers (eigen_sym_static), inlined into its kernels
template <int N>
__device__ __forceinline__ void update_eigen_regs(const double * __restrict__ freqs, const double * __restrict__ subst,
                                                  double * __restrict__ evals, double * __restrict__ evecs, double * __restrict__ ievecs)
{
  {
    double a[N][N], d[N], e[N], fr[N];
    constexpr int NP = N*(N-1)/2;
    const double last = subst[NP-1];
#pragma unroll
    for (int i = 0; i < N; ++i) { fr[i] = freqs[i];
#pragma unroll
      for (int j = 0; j < N; ++j) a[i][j] = 0.0; }
    int k = 0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = i + 1; j < N; ++j)
      {
        double x = subst[k++];
        if (last > 0.0) x /= last;                 // core_pmatrix.c:198-202
        a[i][j] = a[j][i] = x*sqrt(fr[i]*fr[j]);
        a[i][i] -= x*fr[j];
        a[j][j] -= x*fr[i];
      }
    double mean = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) mean += fr[i]*(-a[i][i]);
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < N; ++j) a[i][j] /= mean;
    eigen_sym_static<N>(a, d, e);
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
      evals[i] = d[i];
#pragma unroll
      for (int j = 0; j < N; ++j)
      {
        const double sq = sqrt(fr[j]);
        ievecs[j*N + i] = a[i][j]/sq;
        evecs[i*N + j]  = a[i][j]*sq;
      }
    }
  }
}
template <int N>
__device__ void update_eigen_dev(const double * __restrict__ freqs, const double * __restrict__ subst,
                                 double * __restrict__ evals, double * __restrict__ evecs,
                                 double * __restrict__ ievecs, EigenWork<N> & W)
{
  if constexpr (N == 4)
  {
    update_eigen_regs<4>(freqs, subst, evals, evecs, ievecs);
    return;
  }
  else
  {
  double (&a)[N][N] = W.a; double (&d)[N] = W.d; double (&e)[N] = W.e;
  constexpr int NP = N*(N-1)/2;
  const double last = subst[NP-1];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) a[i][j] = 0.0;
  int k = 0;
  for (int i = 0; i < N; ++i)
    for (int j = i + 1; j < N; ++j)
    {
      double x = subst[k++];
      if (last > 0.0) x /= last;                 // core_pmatrix.c:198-202
      a[i][j] = a[j][i] = x*sqrt(freqs[i]*freqs[j]);
      a[i][i] -= x*freqs[j];
      a[j][j] -= x*freqs[i];
    }
  double mean = 0;
  for (int i = 0; i < N; ++i) mean += freqs[i]*(-a[i][i]);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) a[i][j] /= mean;

  eigen_sym<N>(a, d, e);

  for (int i = 0; i < N; ++i)
  {
    evals[i] = d[i];
    for (int j = 0; j < N; ++j)
    {
      const double sq = sqrt(freqs[j]);
      ievecs[j*N + i] = a[i][j]/sq;
      evecs[i*N + j]  = a[i][j]*sq;
    }
  }
  }
}

template <int N>
__device__ void update_eigen_dev(const double * __restrict__ freqs, const double * __restrict__ subst,
                                 double * __restrict__ evals, double * __restrict__ evecs, double * __restrict__ ievecs)
{
  EigenWork<N> w;
  update_eigen_dev<N>(freqs, subst, evals, evecs, ievecs, w);
}

// refresh the eigensystems of the listed loci (locus.c:2462-2476)
// SK = 4 / 20: every listed locus has that many states (the instance carries only that eigensolver: the 20-state one needs
// 10 KB of scratch per lane, which the all-in-one kernel reserved for 4-state loci too); SK = 0: mixed list
template <int SK>
__global__ void __launch_bounds__(64) eigen_kernel(const LocusDev * loci, const uint32_t * list, uint32_t count)
{
  const uint32_t i = blockIdx.x*64 + threadIdx.x;
  if (i >= count) return;
  const LocusDev & L = loci[list[i]];
  const uint32_t R = L.rate_cats, S = SK ? (uint32_t)SK : L.states;
  for (uint32_t m = 0; m < L.rate_matrices; ++m)
  {
    double * pm = L.par + par_matrix(R, S, m);
    if (SK == 4 || (SK == 0 && S == 4))
      update_eigen_regs<4>(pm + pm_freqs(4), pm + pm_subst(4), pm + pm_evals(4), pm + pm_evecs(4), pm + pm_ievecs(4));
    else
      update_eigen_dev<20>(pm + pm_freqs(20), pm + pm_subst(20), pm + pm_evals(20), pm + pm_evecs(20), pm + pm_ievecs(20));
  }
}

// Batched substitution-parameter proposal (bpa_plan_set_params): one lane per locus of the plan installs its
// new values (which: 1 base frequencies of rate matrix 0, 2 its exchangeabilities, 4 category rates) in the
// locus's parameter block and, when the rate matrix changed, refreshes its eigensystem in place
// (pll_update_eigen on the next locus_update_matrices, locus.c:2462-2476) — K6 on the device, no host trip.
template <int SK>                                 // as eigen_kernel: the state count of every locus of the plan, or 0
__global__ void __launch_bounds__(64) params_install_kernel(const LocusDev * loci, const uint32_t * task_locus, uint32_t ntasks,
                                                           uint32_t which, const double * __restrict__ values, uint32_t len)
{
  const uint32_t t = blockIdx.x*64 + threadIdx.x;
  if (t >= ntasks) return;
  const LocusDev & L = loci[task_locus[t]];
  const uint32_t R = L.rate_cats, S = SK ? (uint32_t)SK : L.states;
  const double * v = values + (size_t)t*len;
  double * pm = L.par + par_matrix(R, S, 0);
  if (which == 4u) { for (uint32_t k = 0; k < R; ++k) L.par[par_rates(R) + k] = v[k]; return; }
  if (which == 1u) for (uint32_t i = 0; i < S; ++i) pm[pm_freqs(S) + i] = v[i];
  if (which == 2u) for (uint32_t i = 0; i < S*(S-1)/2; ++i) pm[pm_subst(S) + i] = v[i];
  if (L.model == 0) return;                       // JC69: no eigensystem
  if (SK == 4 || (SK == 0 && S == 4)) update_eigen_regs<4>(pm + pm_freqs(4), pm + pm_subst(4), pm + pm_evals(4), pm + pm_evecs(4), pm + pm_ievecs(4));
  else                                update_eigen_dev<20>(pm + pm_freqs(20), pm + pm_subst(20), pm + pm_evals(20), pm + pm_evecs(20), pm + pm_ievecs(20));
}

// pll_update_eigen over staged arrays (bpa_update_eigen)
__global__ void eigen_lib_kernel(uint32_t S, const double * freqs, const double * subst,
                                 double * evals, double * evecs, double * ievecs)
{
  if (threadIdx.x || blockIdx.x) return;
  if (S == 4) update_eigen_dev<4>(freqs, subst, evals, evecs, ievecs);
  else        update_eigen_dev<20>(freqs, subst, evals, evecs, ievecs);
}

// thr_task[g] = task owning pattern-thread g (binary search in the prefix array)
__global__ void __launch_bounds__(BPA_BLOCK) build_thr_task_kernel(const uint32_t * __restrict__ task_pat_off,
                                                                   uint32_t ntasks, uint32_t npatterns,
                                                                   uint32_t * __restrict__ thr_task)
{
  const uint32_t g = blockIdx.x*BPA_BLOCK + threadIdx.x;
  if (g >= npatterns) return;
  uint32_t lo = 0, hi = ntasks;           // task_pat_off[lo] <= g < task_pat_off[hi]
  while (hi - lo > 1)
  {
    const uint32_t mid = (lo + hi) >> 1;
    if (task_pat_off[mid] <= g) lo = mid; else hi = mid;
  }
  thr_task[g] = lo;
}
