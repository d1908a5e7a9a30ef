// Microbenchmarks that decide the groupby/join kernel design on MI355X (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 microbench.hip -o microbench
// Not part of the product; results are recorded in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

// ---- streaming read: sum of 16-B loads
__global__ void k_read(const uint4* __restrict__ in, size_t n16, uint64_t* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  uint64_t acc = 0;
  for (; i < n16; i += stride) { uint4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 0x1234567) out[0] = acc;
}
__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ outp, size_t n16) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) outp[i] = in[i];
}
__global__ void k_fill(uint4* outp, size_t n16) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) { uint64_t h = mix64(i); outp[i] = make_uint4((uint32_t)h, (uint32_t)(h >> 32), (uint32_t)i, 7u); }
}

// ---- scattered global atomics. MODE: 0 u64 add agent, 1 f64 add agent, 2 u64 add workgroup scope,
// 3 f64 add workgroup scope, 4 u64 CAS agent, 5 plain load (random 8B), 6 plain RMW non-atomic (load+store)
template <int MODE>
__global__ void k_atomic(uint64_t* table, size_t slots_mask, size_t nops) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  uint64_t acc = 0;
  for (; i < nops; i += stride) {
    size_t s = mix64(i) & slots_mask;
    if (MODE == 0) __hip_atomic_fetch_add(&table[s], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) __hip_atomic_fetch_add((double*)&table[s], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 2) __hip_atomic_fetch_add(&table[s], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 3) __hip_atomic_fetch_add((double*)&table[s], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 4) { unsigned long long exp = 0; __hip_atomic_compare_exchange_strong((unsigned long long*)&table[s], &exp, (unsigned long long)i, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc += exp; }
    if (MODE == 5) acc += table[s];
    if (MODE == 6) table[s] = table[s] + 1;
  }
  if (acc == 0x1234567) table[0] = acc;
}

// ---- LDS atomics: random f64 add + u32 add into a 64 KB LDS table, keys streamed from registers
template <int MODE>
__global__ void __launch_bounds__(256) k_lds_atomic(uint64_t* out, int iters) {
  __shared__ double tbl[8192];
  __shared__ unsigned int cnt[8192];
  for (int j = threadIdx.x; j < 8192; j += 256) { tbl[j] = 0; cnt[j] = 0; }
  __syncthreads();
  uint64_t x = blockIdx.x * 256ull + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    x = mix64(x + it);
    int s = x & 8191;
    if (MODE == 0) { atomicAdd(&tbl[s], 1.0); }
    if (MODE == 1) { atomicAdd(&tbl[s], 1.0); atomicAdd(&cnt[s], 1u); }
    if (MODE == 2) { unsigned long long* p = (unsigned long long*)&tbl[s]; atomicCAS(p, 0ull, x); }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (uint64_t)tbl[5] + cnt[7];
}

// ---- run-scatter write: each wave writes runs of RUN bytes to pseudo-random run-aligned locations
// emulates the partition kernel's output pattern. Each lane writes 16 B; RUN/16 lanes form one run.
template <int RUN>
__global__ void k_runscatter(uint4* outp, size_t nruns_mask, size_t n16) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  constexpr int LPR = RUN / 16;  // lanes per run
  for (; i < n16; i += stride) {
    size_t run = i / LPR;
    size_t dst_run = mix64(run) & nruns_mask;
    outp[dst_run * LPR + (i % LPR)] = make_uint4((uint32_t)i, 1u, 2u, 3u);
  }
}

static float timeit(hipStream_t st, int reps, void (*fn)(void*), void* ctx) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  fn(ctx); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int r = 0; r < reps; ++r) fn(ctx);
  CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d MHz L2 %d\n", p.name, p.multiProcessorCount, p.clockRate / 1000, p.l2CacheSize);
  hipStream_t st = 0;
  const size_t big = 4ull << 30;  // 4 GiB buffers
  uint4 *a, *b; uint64_t* sink;
  CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big)); CK(hipMalloc(&sink, 1 << 20));
  size_t n16 = big / 16;
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, a, n16);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, b, n16);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto T = [&](auto&& launch, int reps) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
  };
  for (int grid : {2048, 4096, 8192}) {
    float ms = T([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, st, a, n16, sink); }, 5);
    printf("read   4GiB grid %5d: %.3f ms  %.2f TB/s\n", grid, ms, big / ms / 1e9);
    ms = T([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st, a, b, n16); }, 5);
    printf("copy   4GiB grid %5d: %.3f ms  %.2f TB/s (r+w)\n", grid, ms, 2.0 * big / ms / 1e9);
  }
  // MALL-resident ping-pong: copy within 64 MiB / 128 MiB windows repeatedly
  for (size_t win : {32ull << 20, 64ull << 20, 128ull << 20, 512ull << 20}) {
    size_t w16 = win / 16;
    float ms = T([&] { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, st, a, b, w16); }, 5);
    printf("copy window %4zu MiB x8: %.3f ms  %.2f TB/s (r+w)\n", win >> 20, ms, 8 * 2.0 * win / ms / 1e9);
    ms = T([&] { for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, st, a, w16, sink); }, 5);
    printf("read window %4zu MiB x8: %.3f ms  %.2f TB/s\n", win >> 20, ms, 8 * 1.0 * win / ms / 1e9);
  }
  // scattered atomics
  uint64_t* tbl = (uint64_t*)b;
  const size_t nops = 1ull << 28;
  for (size_t tbytes : {1ull << 20, 2ull << 20, 16ull << 20, 32ull << 20, 1ull << 30}) {
    size_t mask = tbytes / 8 - 1;
    CK(hipMemsetAsync(tbl, 0, tbytes, st));
    float ms;
#define RUNA(MODE, name) ms = T([&] { hipLaunchKernelGGL(k_atomic<MODE>, dim3(8192), dim3(256), 0, st, tbl, mask, nops); }, 2); \
    printf("atomic %-22s table %5zu MiB: %8.3f ms  %7.2f Gops/s\n", name, tbytes >> 20, ms, nops / ms / 1e6);
    RUNA(0, "u64 add agent"); RUNA(1, "f64 add agent"); RUNA(2, "u64 add wg-scope"); RUNA(3, "f64 add wg-scope");
    RUNA(4, "u64 cas agent"); RUNA(5, "plain load 8B"); RUNA(6, "plain load+store 8B");
  }
  // LDS atomics
  {
    int iters = 4096; int grid = 256 * 4;
    float ms;
    ms = T([&] { hipLaunchKernelGGL(k_lds_atomic<0>, dim3(grid), dim3(256), 0, st, sink, iters); }, 3);
    printf("lds f64 add          : %.3f ms  %.1f Gops/s\n", ms, (double)grid * 256 * iters / ms / 1e6);
    ms = T([&] { hipLaunchKernelGGL(k_lds_atomic<1>, dim3(grid), dim3(256), 0, st, sink, iters); }, 3);
    printf("lds f64 add + u32 add: %.3f ms  %.1f Grows/s\n", ms, (double)grid * 256 * iters / ms / 1e6);
    ms = T([&] { hipLaunchKernelGGL(k_lds_atomic<2>, dim3(grid), dim3(256), 0, st, sink, iters); }, 3);
    printf("lds u64 cas          : %.3f ms  %.1f Gops/s\n", ms, (double)grid * 256 * iters / ms / 1e6);
  }
  // run-scatter writes over a 4 GiB destination
  {
    float ms;
#define RUNS(RUN) ms = T([&] { hipLaunchKernelGGL(k_runscatter<RUN>, dim3(8192), dim3(256), 0, st, b, (big / RUN) - 1, n16); }, 3); \
    printf("run-scatter write run %5d B: %.3f ms  %.2f TB/s\n", RUN, ms, big / ms / 1e9);
    RUNS(16); RUNS(32); RUNS(64); RUNS(128); RUNS(256); RUNS(512); RUNS(1024);
  }
  return 0;
}
