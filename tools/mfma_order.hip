// v_mfma_f64_4x4x4_4b_f64 on gfx950: verify layout A[blk][i][k] @ lane k*16+blk*4+i, B[blk][k][j] @ lane k*16+blk*4+j,
// D[blk][i][j] @ lane i*16+blk*4+j, and that the k-accumulation is an ascending fma chain seeded with C.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double * A, const double * B, const double * C, double * D)
{
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], C[l], 0, 0, 0);
}
int main()
{
  double * dA, * dB, * dC, * dD; (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dC, 512); (void)hipMalloc(&dD, 512);
  std::vector<double> A(64), B(64), C(64), D(64);
  srand(7);
  int asc = 0, desc = 0, tot = 0;
  for (int trial = 0; trial < 50; ++trial)
  {
    for (int i = 0; i < 64; ++i) { A[i] = rand()/(double)RAND_MAX*1e-3; B[i] = rand()/(double)RAND_MAX; C[i] = rand()/(double)RAND_MAX*1e-4; }
    (void)hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(dC, C.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    (void)hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
    for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
    {
      const int o = i*16 + blk*4 + j;
      double s = C[o]; for (int kk = 0; kk < 4; ++kk) s = fma(A[kk*16 + blk*4 + i], B[kk*16 + blk*4 + j], s);
      double r = C[o]; for (int kk = 3; kk >= 0; --kk) r = fma(A[kk*16 + blk*4 + i], B[kk*16 + blk*4 + j], r);
      asc += (s == D[o]); desc += (r == D[o]); ++tot;
    }
  }
  printf("4x4x4_4b: ascending-k fma chain from C: %d/%d bit-exact; descending: %d/%d\n", asc, tot, desc, tot);
  return 0;
}
