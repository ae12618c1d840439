// latency probes on MI355X: empty grid, dependent-load chains (cold/warm), to calibrate the
// fused step kernel's floor.  hipcc --offload-arch=gfx950 -O3 tools/probe.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
__global__ void empty_kernel(int * p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void chase_kernel(const unsigned * __restrict__ next, unsigned * out, int depth, unsigned stride)
{
  unsigned i = (blockIdx.x*blockDim.x + threadIdx.x)*stride;
  for (int d = 0; d < depth; ++d) i = next[i];
  out[blockIdx.x*blockDim.x + threadIdx.x] = i;
}
__global__ void math_kernel(double * out, int n)
{
  double x = 1e-3*(threadIdx.x + 1);
  for (int i = 0; i < n; ++i) x = log(exp(-x) + 1.0);
  out[blockIdx.x*blockDim.x + threadIdx.x] = x;
}
static float timeit(hipStream_t s, void (*launch)(hipStream_t, hipEvent_t, hipEvent_t), int reps)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  std::vector<float> v;
  for (int r = 0; r < reps; ++r) { launch(s, a, b); hipStreamSynchronize(s); float ms; hipEventElapsedTime(&ms, a, b); v.push_back(ms*1e3f); }
  std::sort(v.begin(), v.end());
  return v[v.size()/2];
}
static unsigned * d_next; static unsigned * d_out; static double * d_dout; static int g_depth; static unsigned g_stride; static int g_blocks = 839;
int main()
{
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const size_t N = 64u << 20;                      // 256 MB of indices: cold for L2, partly MALL
  std::vector<unsigned> h(N);
  for (size_t i = 0; i < N; ++i) h[i] = (unsigned)((i*2654435761ull + 12345) % N);
  hipMalloc(&d_next, N*4); hipMemcpy(d_next, h.data(), N*4, hipMemcpyHostToDevice);
  hipMalloc(&d_out, 1 << 22); hipMalloc(&d_dout, 1 << 23);
  printf("empty 839x64: %.2f us\n", timeit(s, [](hipStream_t st, hipEvent_t a, hipEvent_t b){ hipExtLaunchKernelGGL(empty_kernel, dim3(839), dim3(64), 0, st, a, b, 0, (int*)nullptr); }, 200));
  for (int depth : {1, 2, 4, 8})
  {
    g_depth = depth; g_stride = 977;
    printf("chase cold depth %d: %.2f us\n", depth, timeit(s, [](hipStream_t st, hipEvent_t a, hipEvent_t b){ hipExtLaunchKernelGGL(chase_kernel, dim3(g_blocks), dim3(64), 0, st, a, b, 0, d_next, d_out, g_depth, g_stride); }, 50));
  }
  // warm: small table that stays in L2
  for (size_t i = 0; i < (1u << 16); ++i) h[i] = (unsigned)((i*40503u + 7) % (1u << 16));
  hipMemcpy(d_next, h.data(), (1u << 16)*4, hipMemcpyHostToDevice);
  for (int depth : {1, 4, 16})
  {
    g_depth = depth; g_stride = 1;
    printf("chase warm depth %d: %.2f us\n", depth, timeit(s, [](hipStream_t st, hipEvent_t a, hipEvent_t b){ hipExtLaunchKernelGGL(chase_kernel, dim3(g_blocks), dim3(64), 0, st, a, b, 0, d_next, d_out, g_depth, g_stride); }, 200));
  }
  for (int n : {1, 4, 16})
  {
    g_depth = n;
    printf("exp+log x%d: %.2f us\n", n, timeit(s, [](hipStream_t st, hipEvent_t a, hipEvent_t b){ hipExtLaunchKernelGGL(math_kernel, dim3(g_blocks), dim3(64), 0, st, a, b, 0, d_dout, g_depth); }, 200));
  }
  // back-to-back launches without sync: steady-state per-launch time
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, s);
  for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(839), dim3(64), 0, s, (int*)nullptr);
  hipEventRecord(b, s); hipStreamSynchronize(s);
  float ms; hipEventElapsedTime(&ms, a, b); printf("1000 back-to-back empty launches: %.2f us each\n", ms);
  return 0;
}
