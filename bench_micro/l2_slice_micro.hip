// Can an XCD keep a table slice in its L2 while it streams other data through? Workgroup b (XCD b % 8) does random
// 16-byte loads inside window (b % 8) of `win_mb` MB, optionally interleaved with streaming 16-byte loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
template <bool STREAM>
__global__ void __launch_bounds__(256) k(u64x2 const* table, uint64_t slots_per_win, u64x2 const* stream, uint64_t per_block, int iters, uint64_t* out)
{
  int const x = blockIdx.x & 7;
  u64x2 const* win = table + x * slots_per_win;
  u64x2 const* st  = stream + blockIdx.x * per_block;
  uint64_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint64_t idx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint64_t key = (uint64_t(blockIdx.x) * iters + it) * 1024 + k * 256 + threadIdx.x;
      if (STREAM) key ^= st[(uint64_t(it) * 1024 + k * 256 + threadIdx.x) % per_block].x;
      idx[k] = (uint32_t(mix(key)) * slots_per_win) >> 32;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += win[idx[k]].y;
  }
  if (acc == 0x1234567) out[0] = acc;
}
int main()
{
  uint64_t const total_slots = (1600ull << 20) / 16;  // 1.6 GB table
  u64x2 *table, *stream; uint64_t* out;
  CK(hipMalloc(&table, total_slots * 16)); CK(hipMemset(table, 0, total_slots * 16));
  uint64_t const per_block = 1 << 18;  // 4 MB of stream per block
  int const blocks = 2048, iters = 256;  // 2048 * 256 * 1024 = 537M probes
  CK(hipMalloc(&stream, uint64_t(blocks) * per_block * 16)); CK(hipMemset(stream, 0, uint64_t(blocks) * per_block * 16));
  CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (double win_mb : {1.0, 2.0, 3.0, 4.0, 8.0, 64.0, 200.0}) {
    uint64_t const spw = uint64_t(win_mb * (1 << 20)) / 16;
    for (int s = 0; s < 2; ++s) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (s) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, table, spw, stream, per_block, iters, out);
        else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, table, spw, stream, per_block, iters, out);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("window %6.1f MB per XCD, %s: %.3f ms  %.1f G probes/s\n", win_mb, s ? "with 16-B streaming " : "random loads only   ", ms, double(blocks) * iters * 1024 / ms / 1e6);
    }
  }
  return 0;
}
