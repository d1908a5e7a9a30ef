// Which XCD does workgroup b run on? (s_getreg XCC_ID) - checks the round-robin assumption of the join's partitioned probe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out)
{
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x & 0xf;
}
int main()
{
  for (int threads : {256, 1024}) {
    int const n = 4096;
    unsigned* d;
    hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k, dim3(n), dim3(threads), 0, 0, d);
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int i = 0; i < n; ++i) ok += h[i] == unsigned(i % 8);
    printf("threads %d: first 24 blocks:", threads);
    for (int i = 0; i < 24; ++i) printf(" %u", h[i]);
    printf("\n  blocks with xcc == b %% 8: %d of %d\n", ok, n);
    hipFree(d);
  }
}
