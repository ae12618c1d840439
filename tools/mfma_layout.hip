// discover the lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 empirically
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double * A, const double * B, double * D)
{
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}
int main()
{
  double * dA, * dB, * dD; (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dD, 512);
  std::vector<double> A(64), B(64), D(64);
  // pair[la][lb] = output lane where A[la]*B[lb] lands (or -1)
  static int pairout[64][64];
  for (int la = 0; la < 64; ++la)
  {
    for (int i = 0; i < 64; ++i) { A[i] = (i == la) ? 1.0 : 0.0; B[i] = i + 1.0; }
    (void)hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
    for (int lb = 0; lb < 64; ++lb) pairout[la][lb] = -1;
    for (int o = 0; o < 64; ++o) if (D[o] != 0.0) pairout[la][(int)D[o] - 1] = o;
  }
  // print for A lane la: list of (B lane -> out lane)
  for (int la = 0; la < 64; ++la)
  {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb) if (pairout[la][lb] >= 0) printf(" B%d->D%d", lb, pairout[la][lb]);
    printf("\n");
  }
  return 0;
}
