// f64 MFMA probes on gfx950: rate of v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 and whether the
// k-accumulation is a sequential fma chain (needed to keep a summation order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstring>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void rate16(double * out, int iters)
{
  double a = 1.0 + threadIdx.x*1e-3, b = 1.0 - threadIdx.x*1e-3;
  double4_t c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i)
  {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x*blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void rate4(double * out, int iters)
{
  double a = 1.0 + threadIdx.x*1e-3, b = 1.0 - threadIdx.x*1e-3;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  for (int i = 0; i < iters; ++i)
  {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
  }
  out[blockIdx.x*blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
__global__ void ratefma(double * out, int iters)
{
  double a = 1.0 + threadIdx.x*1e-9, b = 1e-9*threadIdx.x;
  double c[16];
  for (int j = 0; j < 16; ++j) c[j] = j;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = __builtin_fma(a, c[j], b);
  double s = 0; for (int j = 0; j < 16; ++j) s += c[j];
  out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}
// layout + order check for 16x16x4: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)+4*r][col=l&15]
__global__ void check16(const double * A, const double * B, double * D)   // A 16x4, B 4x16 row-major
{
  const int l = threadIdx.x;
  double4_t c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15)*4 + (l >> 4)], B[(l >> 4)*16 + (l & 15)], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4*r)*16 + (l & 15)] = c[r];
}
// 4x4x4 4 blocks: block = l>>4 ; within a block: A[i=l&3][k=(l>>2)&3], B[k=(l>>2)&3][j=l&3] ? D[i][j] ?
__global__ void check4(const double * A, const double * B, double * D)    // per block: A 4x4, B 4x4, D 4x4 (lane dump)
{
  const int l = threadIdx.x;
  double c = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
  D[l] = c;
}
int main()
{
  double * d; hipMalloc(&d, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int dev; hipGetDevice(&dev); hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
  const double clk = pr.clockRate*1e3;   // Hz (max)
  const int iters = 20000, blocks = 256*4*2, threads = 64;   // 2 waves per SIMD
  auto run = [&](const char * name, void (*k)(double *, int), double flop_per_wave_iter)
  {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = flop_per_wave_iter*iters*blocks/(ms*1e-3)/1e12;
    printf("%-28s %.3f ms  %.1f TFLOP/s\n", name, ms, tf);
  };
  run("mfma_f64_16x16x4 (x4 acc)", rate16, 4*2.0*16*16*4);
  run("mfma_f64_4x4x4_4b (x8 acc)", rate4, 8*2.0*4*4*4*4);
  run("v_fma_f64 (x16 acc)", ratefma, 16*2.0*64);
  printf("clock (max) %.0f MHz\n", clk/1e6);
  // numerics / layout
  std::vector<double> A(64), B(64), D(256), Dh(256);
  srand(3); for (auto & x : A) x = rand()/(double)RAND_MAX - 0.3; for (auto & x : B) x = rand()/(double)RAND_MAX - 0.6;
  double * dA, * dB, * dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check16, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
  int seq = 0, rev = 0, tot = 0; double maxerr = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j)
  {
    double s = 0; for (int k = 0; k < 4; ++k) s = fma(A[i*4+k], B[k*16+j], s);
    double r = 0; for (int k = 3; k >= 0; --k) r = fma(A[i*4+k], B[k*16+j], r);
    seq += (s == D[i*16+j]); rev += (r == D[i*16+j]); ++tot; maxerr = fmax(maxerr, fabs(s - D[i*16+j]));
  }
  printf("16x16x4: matches ascending-k fma chain %d/%d, descending %d/%d, max abs diff %.3g\n", seq, tot, rev, tot, maxerr);
  hipLaunchKernelGGL(check4, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
  // try to identify the 4x4x4 layout: for each lane l find (blk, i, j) conventions
  // hypothesis H: blk = l>>4, i = l&3 (row of A... ), k = (l>>2)&3 ; output lane l holds D[blk][i=?][j=?]
  int ok1 = 0, ok2 = 0;
  for (int l = 0; l < 64; ++l)
  {
    int blk = l >> 4, q = l & 15;
    // candidate 1: A lane (blk, i=q&3, k=q>>2), B lane (blk, k=q>>2, j=q&3), D lane (blk, i=q>>2?, j=q&3)
    for (int cand = 0; cand < 2; ++cand)
    {
      int i = cand == 0 ? (q >> 2) : (q & 3), j = cand == 0 ? (q & 3) : (q >> 2);
      double s = 0;
      for (int k = 0; k < 4; ++k) s = fma(A[blk*16 + (k << 2) + i], B[blk*16 + (k << 2) + j], s);
      if (s == D[l]) (cand == 0 ? ok1 : ok2)++;
    }
  }
  printf("4x4x4_4b layout: A[blk][i=l&3][k=(l>>2)&3], B[blk][k=(l>>2)&3][j=l&3]; D lane=(i=q>>2,j=q&3): %d/64, D lane=(i=q&3,j=q>>2): %d/64\n", ok1, ok2);
  return 0;
}
