// What HBM rate does a kernel of the 20-state K1+K2 kernel's ACCESS SHAPE reach with no arithmetic at all?
// (is partials_lnl_tiledk_kernel at the ceiling of its access pattern?)
//   stream16 : the plain ceiling — contiguous 16-byte loads of two arrays, 16-byte stores of a third (read:write 2:1)
//   planes   : one workgroup = 64 patterns x 4 rate categories of one locus's node update; wave k reads the 20
//              state planes of the k-th category of two child CLVs (8-byte loads, 64 lanes wide, plane stride = Np
//              doubles) and writes the 20 planes of the parent — config 4's layout, Np = 105 (second workgroup of a
//              locus 41/64 full) and, for comparison, Np = 128 (every workgroup full)
// bytes counted = bytes the lanes actually touch (inactive lanes touch nothing)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) stream16(const double2 * __restrict__ a, const double2 * __restrict__ b, double2 * __restrict__ o, size_t n)
{
  size_t i = (size_t)blockIdx.x*256 + threadIdx.x, st = (size_t)gridDim.x*256;
  for (; i < n; i += st) { double2 x = a[i], y = b[i]; o[i] = make_double2(x.x + y.x, x.y + y.y); }
}

template<int NT> __global__ void __launch_bounds__(256) planes(const double * __restrict__ c1, const double * __restrict__ c2, double * __restrict__ par,
                                                              int np, int wg_per_locus, int nupd, int ld)
{
  // blockIdx -> (locus, pattern block); the update index walks nupd node buffers of the locus
  int locus = blockIdx.x / wg_per_locus, pb = blockIdx.x % wg_per_locus;
  int k = threadIdx.x >> 6, lane = threadIdx.x & 63, n = pb*64 + lane;
  if (n >= np) return;
  size_t node = (size_t)4*20*ld;                       // one CLV: [category][state][pattern], plane stride ld >= np
  for (int u = 0; u < nupd; ++u)
  {
    size_t base = ((size_t)locus*nupd + u)*node + (size_t)k*20*ld + n;
    double x[20], y[20];
    #pragma unroll
    for (int s = 0; s < 20; ++s) { x[s] = NT ? __builtin_nontemporal_load(c1 + base + (size_t)s*ld) : c1[base + (size_t)s*ld]; }
    #pragma unroll
    for (int s = 0; s < 20; ++s) { y[s] = NT ? __builtin_nontemporal_load(c2 + base + (size_t)s*ld) : c2[base + (size_t)s*ld]; }
    #pragma unroll
    for (int s = 0; s < 20; ++s) { if (NT) __builtin_nontemporal_store(x[s] + y[s], par + base + (size_t)s*ld); else par[base + (size_t)s*ld] = x[s] + y[s]; }
  }
}

static float median_us(std::vector<float> & v) { std::sort(v.begin(), v.end()); return v[v.size()/2]; }

int main()
{
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int loci = 2000, nupd = 3;
  const int shapes[][2] = {{105,105},{105,112},{105,128},{128,128},{64,64},{192,192}};
  for (auto & sh : shapes)
  {
    int np = sh[0], ld = sh[1];
    size_t node = (size_t)4*20*ld, tot = (size_t)loci*nupd*node;
    double *c1, *c2, *o;
    if (hipMalloc(&c1, tot*8) != hipSuccess || hipMalloc(&c2, tot*8) != hipSuccess || hipMalloc(&o, tot*8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemsetAsync(c1, 0, tot*8, s); (void)hipMemsetAsync(c2, 0, tot*8, s); (void)hipMemsetAsync(o, 0, tot*8, s);
    int wgl = (np + 63)/64;
    double bytes = 3.0*8*(double)loci*nupd*4*20*np;
    for (int nt = 0; nt < 2; ++nt)
    {
      std::vector<float> v;
      for (int r = 0; r < 40; ++r)
      {
        if (nt) hipExtLaunchKernelGGL(planes<1>, dim3(loci*wgl), dim3(256), 0, s, e0, e1, 0, c1, c2, o, np, wgl, nupd, ld);
        else    hipExtLaunchKernelGGL(planes<0>, dim3(loci*wgl), dim3(256), 0, s, e0, e1, 0, c1, c2, o, np, wgl, nupd, ld);
        (void)hipStreamSynchronize(s); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (r >= 5) v.push_back(ms*1e3f);
      }
      float us = median_us(v);
      printf("planes   Np %3d stride %3d %s: %7.1f MB touched, median %7.1f us = %6.0f GB/s (%.3f of 8 TB/s)\n", np, ld, nt ? "nontemporal" : "plain      ", bytes/1e6, us, bytes/us/1e3, bytes/us/1e3/8000.0);
    }
    {
      std::vector<float> v; size_t n16 = tot/2;
      for (int r = 0; r < 40; ++r)
      {
        hipExtLaunchKernelGGL(stream16, dim3(256*16), dim3(256), 0, s, e0, e1, 0, (const double2*)c1, (const double2*)c2, (double2*)o, n16);
        (void)hipStreamSynchronize(s); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (r >= 5) v.push_back(ms*1e3f);
      }
      float us = median_us(v);
      double sb = 3.0*16*n16;
      printf("stream16 (%7.1f MB, read:write 2:1)      : median %7.1f us = %6.0f GB/s (%.3f of 8 TB/s)\n", sb/1e6, us, sb/us/1e3, sb/us/1e3/8000.0);
    }
    (void)hipFree(c1); (void)hipFree(c2); (void)hipFree(o);
  }
  return 0;
}
