// FP64 issue rates of gfx950, measured: v_fma_f64 (VALU) against v_mfma_f64_4x4x4 and v_mfma_f64_16x16x4 (matrix cores).
// Independent accumulator chains, everything in registers; prints TFLOP/s of the whole chip.
//   hipcc --offload-arch=gfx950 -O3 tools/f64_rate.hip -o tools/f64_rate.bin && tools/f64_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ void __launch_bounds__(256) rate_kernel(double * out, int iters, double seed)
{
  double a = seed + threadIdx.x*1e-9, b = 1.0 + 1e-12*threadIdx.x;
  if (MODE == 0)
  {
    double c[16];
    for (int i = 0; i < 16; ++i) c[i] = i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) c[i] = __builtin_fma(a, b, c[i]);
    double s = 0; for (int i = 0; i < 16; ++i) s += c[i];
    out[blockIdx.x*256 + threadIdx.x] = s;
  }
  else if (MODE == 1)
  {
    double c[16];
    for (int i = 0; i < 16; ++i) c[i] = i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
    double s = 0; for (int i = 0; i < 16; ++i) s += c[i];
    out[blockIdx.x*256 + threadIdx.x] = s;
  }
  else
  {
    d4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = d4{(double)i, 1, 2, 3};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    double s = 0; for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x*256 + threadIdx.x] = s;
  }
}
template <int MODE> static double run(double * d, int iters, double flops_per_wave_inst, int insts_per_iter)
{
  const int blocks = 256*8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d, 16, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double waves = blocks*4.0;
  return waves*iters*insts_per_iter*flops_per_wave_inst/(ms*1e-3)/1e12;
}
int main()
{
  double * d; hipMalloc(&d, 256*8*256*sizeof(double));
  const int iters = 20000;
  printf("v_fma_f64            : %7.1f TFLOP/s (64 lanes x 2 flops per instruction)\n", run<0>(d, iters, 128.0, 16));
  printf("v_mfma_f64_4x4x4     : %7.1f TFLOP/s (4 blocks x 4x4x4 x 2 flops = 512 per instruction)\n", run<1>(d, iters, 512.0, 16));
  printf("v_mfma_f64_16x16x4   : %7.1f TFLOP/s (16x16x4 x 2 flops = 2048 per instruction)\n", run<2>(d, iters, 2048.0, 4));
  return 0;
}
