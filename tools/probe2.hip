#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void empty_kernel(int * p) { if (p && threadIdx.x == 9999) *p = 1; }
int main()
{
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  int shapes[][2] = {{839,64},{420,128},{210,256},{105,512},{53,1024},{256,64},{64,64},{1,64},{2048,64},{8192,64}};
  for (auto & sh : shapes)
  {
    std::vector<float> v;
    for (int r = 0; r < 300; ++r)
    {
      hipExtLaunchKernelGGL(empty_kernel, dim3(sh[0]), dim3(sh[1]), 0, s, a, b, 0, (int*)nullptr);
      (void)hipStreamSynchronize(s); float ms; (void)hipEventElapsedTime(&ms, a, b); v.push_back(ms*1e3f);
    }
    std::sort(v.begin(), v.end());
    // back-to-back
    (void)hipEventRecord(a, s);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(sh[0]), dim3(sh[1]), 0, s, (int*)nullptr);
    (void)hipEventRecord(b, s); (void)hipStreamSynchronize(s); float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("grid %5d x %4d: single (ext events) median %.2f us ; back-to-back %.2f us each\n", sh[0], sh[1], v[v.size()/2], ms/2.0f);
  }
  return 0;
}
