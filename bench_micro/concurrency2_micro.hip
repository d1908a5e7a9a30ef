// A-first order: does a 256 x 256-thread kernel (57 KB LDS, 80 VGPRs) START while a resident 256 x 1024-thread kernel (98 KB, 96 VGPRs) is
// still running? The long kernel spins until the short one has raised a flag (or 20 ms); counts how many of its workgroups saw it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
template <int V, int S>
__global__ void __launch_bounds__(1024) k_long(int* flag, int* out, int* cus, long long ticks)
{
  extern __shared__ char lds[];
  if constexpr (V == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if constexpr (S > 0) asm volatile("s_mov_b32 s95, 0" ::: "s95");
  long long const t0 = wall_clock64();
  if (threadIdx.x == 0) {
    int seen = 0;
    while (wall_clock64() - t0 < ticks) {
      if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { seen = 1; break; }
      __builtin_amdgcn_s_sleep(32);
    }
    if (seen) atomicAdd(out, 1);
    lds[0] = 1;
  }
}
template <int V, int S>
__global__ void __launch_bounds__(256) k_short(int* flag, int* started)
{
  extern __shared__ char lds[];
  if constexpr (V == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
  if constexpr (S > 0) asm volatile("s_mov_b32 s65, 0" ::: "s65");
  if (threadIdx.x == 0) { lds[0] = 1; atomicAdd(started, 1); if (atomicAdd(started + 1, 0) >= 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
int main()
{
  int *flag, *out; CK(hipMalloc(&flag, 4)); CK(hipMalloc(&out, 16));
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  auto run = [&](auto kl, auto ks, size_t ll, size_t ls, const char* name) {
    CK(hipFuncSetAttribute(reinterpret_cast<void const*>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<void const*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(flag, 0, 4)); CK(hipMemset(out, 0, 16)); CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(kl, dim3(256), dim3(1024), ll, sa, flag, out, out + 3, 2000000ll);
      hipLaunchKernelGGL(ks, dim3(256), dim3(256), ls, sb, flag, out + 1);
      CK(hipDeviceSynchronize());
      int h[4]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
      printf("%-70s long first: %d of 256 long workgroups saw the short kernel start\n", name, h[0]);
    }
  };
  run(k_long<96, 0>, k_short<80, 0>, 100352, 57640, "96 / 80 VGPRs, LDS 100352 + 57640");
  run(k_long<96, 1>, k_short<80, 1>, 100352, 57640, "96 / 80 VGPRs, 96 / 66 SGPRs, LDS 100352 + 57640");
  run(k_long<96, 1>, k_short<80, 1>, 100384, 57640, "96 / 80 VGPRs, 96 / 66 SGPRs, LDS 100384 + 57640");
  run(k_long<96, 1>, k_short<80, 1>, 98304, 57640, "96 / 80 VGPRs, 96 / 66 SGPRs, LDS 98304 + 57640");
  run(k_long<96, 1>, k_short<80, 1>, 100384, 53248, "96 / 80 VGPRs, 96 / 66 SGPRs, LDS 100384 + 53248");
  run(k_long<96, 1>, k_short<80, 1>, 100384, 49152, "96 / 80 VGPRs, 96 / 66 SGPRs, LDS 100384 + 49152");
  return 0;
}
