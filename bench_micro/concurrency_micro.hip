// Do two kernels launched on two streams run at the same time on this box, and with which shapes?
// Y (launched first) spins until X (launched second, other stream) has raised a flag, or gives up after 0.3 s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
template <int V>
__global__ void __launch_bounds__(256) k_wait(int* flag, int* out, long long ticks)
{
  if constexpr (V == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
  if constexpr (V == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  extern __shared__ char lds[];
  long long const t0 = wall_clock64();
  int seen = 0;
  if (threadIdx.x == 0) {
    while (wall_clock64() - t0 < ticks) {
      if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { seen = 1; break; }
      __builtin_amdgcn_s_sleep(32);
    }
    if (seen) atomicAdd(out, 1);
    lds[0] = 1;
  }
}
template <int V>
__global__ void __launch_bounds__(1024) k_raise(int* flag)
{
  if constexpr (V == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if constexpr (V == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if constexpr (V == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
  extern __shared__ char lds[];
  if (threadIdx.x == 0) { lds[0] = 1; __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
int main()
{
  int *flag, *out; CK(hipMalloc(&flag, 4)); CK(hipMalloc(&out, 4));
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));

  struct { int gy, ty, ly, gx, tx, lx; const char* name; } cases[] = {
    {1, 64, 0, 1, 64, 0, "1 x 64 waiting, 1 x 64 raising"},
    {256, 256, 0, 256, 1024, 0, "256 x 256 waiting, 256 x 1024 raising, no LDS"},
    {256, 256, 57 * 1024, 256, 1024, 98 * 1024, "256 x 256 (57 KB) waiting, 256 x 1024 (98 KB) raising"},
    {256, 256, 57 * 1024, 256, 768, 98 * 1024, "256 x 256 (57 KB) waiting, 256 x 768 (98 KB) raising"},
    {512, 256, 57 * 1024, 256, 1024, 98 * 1024, "512 x 256 (57 KB) waiting, 256 x 1024 (98 KB) raising"},
  };
  auto run = [&](auto kw, auto kr, const char* regs) {
    CK(hipFuncSetAttribute(reinterpret_cast<void const*>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<void const*>(kw), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (auto& c : cases) {
      for (int order = 0; order < 2; ++order) {
        CK(hipMemset(flag, 0, 4)); CK(hipMemset(out, 0, 4)); CK(hipDeviceSynchronize());
        if (order == 0) {
          hipLaunchKernelGGL(kw, dim3(c.gy), dim3(c.ty), c.ly, sb, flag, out, 30000000ll);
          hipLaunchKernelGGL(kr, dim3(c.gx), dim3(c.tx), c.lx, sa, flag);
        } else {
          hipLaunchKernelGGL(kr, dim3(c.gx), dim3(c.tx), c.lx, sa, flag);
          hipLaunchKernelGGL(kw, dim3(c.gy), dim3(c.ty), c.ly, sb, flag, out, 30000000ll);
        }
        CK(hipDeviceSynchronize());
        int h; CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
        printf("[%s] %-62s %s: %d of %d waiting workgroups saw the flag\n", regs, c.name, order == 0 ? "waiter launched first" : "raiser launched first", h, c.gy);
      }
    }
  };
  run(k_wait<0>, k_raise<0>, "few registers");
  run(k_wait<80>, k_raise<64>, "waiter 80, raiser 64 VGPRs");
  run(k_wait<80>, k_raise<96>, "waiter 80, raiser 96 VGPRs");
  run(k_wait<128>, k_raise<96>, "waiter 128, raiser 96 VGPRs");
  run(k_wait<80>, k_raise<128>, "waiter 80, raiser 128 VGPRs");
  return 0;
}
