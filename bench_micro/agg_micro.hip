// Isolates the per-partition LDS aggregation loop: what limits it (streaming pattern, LDS atomics, probe)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; return x; }

__global__ void k_gen(u64x2* rec, int64_t n, int64_t per_part, int groups_per_part) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t part = i / per_part;
    uint64_t h = mix64(i * 0x9E3779B97F4A7C15ull + 12345);
    uint64_t key = part * 100000 + (h % groups_per_part);
    double v = (double)(h >> 11) * (1.0 / 9007199254740992.0);
    u64x2 r; r.x = key; r.y = __double_as_longlong(v);
    rec[i] = r;
  }
}

// MODE 0: direct slot (key % cap), atomics only. MODE 1: hash + open addressing with u64 key CAS. MODE 2: tag protocol (st + keys).
// MODE 3: streaming only (sum to register). MODE 4: hash + probe read-only (no atomics)
template <int MODE, int R>
__global__ void __launch_bounds__(1024) k_agg(const u64x2* __restrict__ rec, int64_t per_part, int cap, double* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint64_t* keys = (uint64_t*)lds;
  double* sums = (double*)(keys + cap);
  unsigned long long* cnts = (unsigned long long*)(sums + cap);
  uint32_t* st = (uint32_t*)(cnts + cap);
  for (int s = threadIdx.x; s < cap; s += blockDim.x) { keys[s] = ~0ull; sums[s] = 0; cnts[s] = 0; st[s] = 0; }
  __syncthreads();
  int64_t begin = blockIdx.x * per_part, end = begin + per_part;
  int64_t B = blockDim.x;
  double acc = 0;
  for (int64_t base = begin; base < end; base += R * B) {
    u64x2 v[R]; bool ok[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { int64_t r = base + k * B + threadIdx.x; ok[k] = r < end; if (ok[k]) v[k] = rec[r]; }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (!ok[k]) continue;
      uint64_t key = v[k].x; double val = __longlong_as_double(v[k].y);
      if (MODE == 3) { acc += val + (double)(key & 1); continue; }
      int slot;
      if (MODE == 0) { slot = (int)(key % (uint64_t)cap); }
      else {
        uint64_t h = mix64(key ^ 0x9e3779b97f4a7c15ull);
        slot = (int)(((uint64_t)(uint32_t)h * (uint32_t)cap) >> 32);
        if (MODE == 1 || MODE == 4) {
          for (;;) {
            uint64_t cur = keys[slot];
            if (cur == key) break;
            if (cur == ~0ull) { uint64_t old = atomicCAS((unsigned long long*)&keys[slot], ~0ull, (unsigned long long)key); if (old == ~0ull || old == key) break; }
            slot = slot + 1 == cap ? 0 : slot + 1;
          }
        } else {
          uint32_t tag = ((uint32_t)(h >> 20) & ~3u) | 2u;
          for (;;) {
            uint32_t s = __hip_atomic_load(&st[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (s == 0) { uint32_t old = atomicCAS(&st[slot], 0u, 1u); if (old == 0) { keys[slot] = key; __hip_atomic_store(&st[slot], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); break; } s = old; }
            if (s == 1) continue;
            if (s == tag && keys[slot] == key) break;
            slot = slot + 1 == cap ? 0 : slot + 1;
          }
        }
      }
      if (MODE == 4) { acc += (double)slot; continue; }
      atomicAdd(&sums[slot], val);
      atomicAdd(&cnts[slot], 1ull);
    }
  }
  __syncthreads();
  if (MODE == 3 || MODE == 4) { if (acc == 1.2345) out[0] = acc; }
  if (threadIdx.x == 0) out[blockIdx.x] = sums[1] + (double)cnts[2];
}

int main(int argc, char** argv) {
  const int64_t n = 1000000000; const int P = argc > 1 ? atoi(argv[1]) : 1024; const int64_t per = n / P; const int cap = argc > 2 ? atoi(argv[2]) : 2633; const int gpp = 1000000 / P;
  printf("P %d cap %d groups/partition %d\n", P, cap, gpp);
  u64x2* rec; double* out; CK(hipMalloc(&rec, n * 16)); CK(hipMalloc(&out, P * 8));
  hipLaunchKernelGGL(k_gen, dim3(4096), dim3(256), 0, 0, rec, n, per, gpp); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  size_t lds = (size_t)cap * 28;
  auto run = [&](auto kern, const char* name, int block) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipLaunchKernelGGL(kern, dim3(P), dim3(block), lds, 0, rec, per, cap, out); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(P), dim3(block), lds, 0, rec, per, cap, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    printf("%-34s block %4d: %7.3f ms  %6.1f Grows/s  %5.2f TB/s\n", name, block, ms, n / ms / 1e6, n * 16.0 / ms / 1e9);
  };
  for (int block : {1024}) {
    run(k_agg<3, 4>, "stream only R=4", block);
    run(k_agg<3, 8>, "stream only R=8", block);
    run(k_agg<0, 4>, "direct slot + 2 atomics R=4", block);
    run(k_agg<4, 4>, "hash + probe(u64 cas) no atomics", block);
    run(k_agg<1, 4>, "hash + u64-CAS probe + atomics R=4", block);
    run(k_agg<1, 8>, "hash + u64-CAS probe + atomics R=8", block);
    run(k_agg<2, 4>, "hash + tag probe + atomics R=4", block);
  }
  return 0;
}
