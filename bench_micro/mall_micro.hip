// Can the 256 MiB Infinity Cache carry the partition intermediates?
//  A  copy 8 GiB of streamed input into a REUSED ring of R bytes (dst wraps): does a cache-resident destination
//     lift the 4.7 TB/s (r+w) HBM copy ceiling?
//  B  re-read a buffer of R bytes: Infinity-Cache read rate by footprint
//  C  chunked pipeline, one stream: [copy chunk -> ring] [read ring] per chunk, over 8 GiB: per-chunk launches + gaps
//  D  the same on two streams with a double ring (copy of chunk c+1 beside the read of chunk c)
//  E  granule scatter (128-B granules to pseudo-random 768-B cells of the ring, the write-combining scatter's pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
using gv = __attribute__((address_space(1))) u64x2;
__device__ __forceinline__ u64x2 gl(u64x2 const* p) { return *reinterpret_cast<gv const*>(reinterpret_cast<uintptr_t>(p)); }
__device__ __forceinline__ void gs(u64x2* p, u64x2 v) { *reinterpret_cast<gv*>(reinterpret_cast<uintptr_t>(p)) = v; }
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// src[base .. base+n16) -> ring[(i) & mask]
__global__ __launch_bounds__(256) void k_copy_ring(u64x2 const* src, size_t n16, u64x2* ring, size_t mask16)
{
  size_t const stride = static_cast<size_t>(gridDim.x) * blockDim.x * 4;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) * 4 + threadIdx.x; i < n16; i += stride) {
    u64x2 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * 256 < n16) v[k] = gl(src + i + k * 256);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * 256 < n16) gs(ring + ((i + k * 256) & mask16), v[k]);
  }
}
__global__ __launch_bounds__(256) void k_read(u64x2 const* buf, size_t n16, size_t mask16, unsigned long long* sink)
{
  size_t const stride = static_cast<size_t>(gridDim.x) * blockDim.x * 4;
  unsigned long long acc = 0;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) * 4 + threadIdx.x; i < n16; i += stride) {
    u64x2 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * 256 < n16) { v[k] = gl(buf + ((i + k * 256) & mask16)); }
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * 256 < n16) acc += v[k].x ^ v[k].y;
  }
  if (acc == 0x123456789abcdefULL) *sink = acc;
}
// granule g of the stream (8 lanes x 16 B) goes to cell mix(g / per_cell) % cells, slot (g % per_cell): a cell is
// per_cell granules long
__global__ __launch_bounds__(256) void k_granule_scatter(u64x2 const* src, size_t n16, u64x2* ring, size_t cells, int per_cell, size_t salt)
{
  size_t const stride = static_cast<size_t>(gridDim.x) * blockDim.x * 4;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) * 4 + threadIdx.x; i < n16; i += stride) {
    u64x2 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * 256 < n16) v[k] = gl(src + i + k * 256);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * 256 < n16) {
      size_t const idx = i + k * 256, g = idx >> 3;
      size_t const cell = mix64(g / per_cell + salt) % cells;
      gs(ring + (cell * per_cell + g % per_cell) * 8 + (idx & 7), v[k]);
    }
  }
}

int main(int argc, char** argv)
{
  size_t const SRC = 8ull << 30;
  u64x2 *src, *ring; unsigned long long* sink;
  CK(hipMalloc(&src, SRC)); CK(hipMalloc(&ring, 8ull << 30)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(src, 1, SRC)); CK(hipMemset(ring, 2, 8ull << 30));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  int const GRID = 2048;
  auto timeit = [&](auto&& body, int reps) {
    body(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s0));
    for (int r = 0; r < reps; ++r) body();
    CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
  };
  std::vector<size_t> sizes = {16ull << 20, 32ull << 20, 64ull << 20, 96ull << 20, 128ull << 20, 192ull << 20, 256ull << 20, 512ull << 20, 2ull << 30, 8ull << 30};
  printf("A: copy 8 GiB stream -> ring of R bytes (one launch; GB/s counts the 8 GiB once = input rate)\n");
  for (size_t R : sizes) {
    if (R & (R - 1)) { // not a power of two: round the mask down, use chunks instead
      continue;
    }
    float ms = timeit([&] { hipLaunchKernelGGL(k_copy_ring, dim3(GRID), dim3(256), 0, s0, src, SRC / 16, ring, R / 16 - 1); }, 3);
    printf("  R=%6zu MiB  %7.3f ms  %6.2f TB/s input (x2 = r+w)\n", R >> 20, ms, SRC / ms / 1e9);
  }
  printf("B: read 8 GiB worth from a buffer of R bytes (wrapping)\n");
  for (size_t R : sizes) {
    if (R & (R - 1)) continue;
    float ms = timeit([&] { hipLaunchKernelGGL(k_read, dim3(GRID), dim3(256), 0, s0, ring, SRC / 16, R / 16 - 1, sink); }, 3);
    printf("  R=%6zu MiB  %7.3f ms  %6.2f TB/s\n", R >> 20, ms, SRC / ms / 1e9);
  }
  printf("C: chunked pipeline on one stream: per chunk [copy chunk -> ring][read ring]; 8 GiB total\n");
  for (size_t R : {16ull << 20, 32ull << 20, 48ull << 20, 64ull << 20, 96ull << 20, 128ull << 20, 192ull << 20, 256ull << 20, 1024ull << 20}) {
    size_t const nchunks = SRC / R;
    float ms = timeit([&] {
      for (size_t c = 0; c < nchunks; ++c) {
        hipLaunchKernelGGL(k_copy_ring, dim3(GRID), dim3(256), 0, s0, src + c * (R / 16), R / 16, ring, ~size_t{0});
        hipLaunchKernelGGL(k_read, dim3(GRID), dim3(256), 0, s0, ring, R / 16, ~size_t{0}, sink);
      }
    }, 2);
    printf("  chunk=%5zu MiB x %4zu  %7.3f ms  %6.2f TB/s input   (%.1f us per chunk pair)\n", R >> 20, nchunks, ms, nchunks * R / ms / 1e9, ms * 1e3 / nchunks);
  }
  printf("C2: the same with a 1024-workgroup grid\n");
  for (size_t R : {32ull << 20, 64ull << 20, 96ull << 20, 128ull << 20}) {
    size_t const nchunks = SRC / R;
    float ms = timeit([&] {
      for (size_t c = 0; c < nchunks; ++c) {
        hipLaunchKernelGGL(k_copy_ring, dim3(1024), dim3(256), 0, s0, src + c * (R / 16), R / 16, ring, ~size_t{0});
        hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s0, ring, R / 16, ~size_t{0}, sink);
      }
    }, 2);
    printf("  chunk=%5zu MiB x %4zu  %7.3f ms  %6.2f TB/s input   (%.1f us per chunk pair)\n", R >> 20, nchunks, ms, nchunks * R / ms / 1e9, ms * 1e3 / nchunks);
  }
  printf("D: two streams, double ring: copy(c+1) beside read(c)\n");
  {
    std::vector<hipEvent_t> copied(2), readdone(2);
    for (auto& e : copied) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : readdone) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (size_t R : {16ull << 20, 32ull << 20, 48ull << 20, 64ull << 20, 96ull << 20}) {
      size_t const nchunks = SRC / R;
      float ms = timeit([&] {
        // e0/e1 are recorded on s0: s0 waits for s1's tail at the end
        for (size_t c = 0; c < nchunks; ++c) {
          int const b = c & 1;
          if (c >= 2) CK(hipStreamWaitEvent(s0, readdone[b], 0));
          hipLaunchKernelGGL(k_copy_ring, dim3(GRID / 2), dim3(256), 0, s0, src + c * (R / 16), R / 16, ring + b * (R / 16), ~size_t{0});
          CK(hipEventRecord(copied[b], s0));
          CK(hipStreamWaitEvent(s1, copied[b], 0));
          hipLaunchKernelGGL(k_read, dim3(GRID / 2), dim3(256), 0, s1, ring + b * (R / 16), R / 16, ~size_t{0}, sink);
          CK(hipEventRecord(readdone[b], s1));
        }
        CK(hipStreamWaitEvent(s0, readdone[0], 0)); CK(hipStreamWaitEvent(s0, readdone[1], 0));
      }, 2);
      printf("  chunk=%5zu MiB x %4zu  %7.3f ms  %6.2f TB/s input   (%.1f us per chunk)\n", R >> 20, nchunks, ms, nchunks * R / ms / 1e9, ms * 1e3 / nchunks);
    }
  }
  printf("E: granule scatter of 8 GiB into cells of 6 granules (768 B) spread over R bytes\n");
  for (size_t R : {64ull << 20, 128ull << 20, 256ull << 20, 1024ull << 20, 8192ull << 20}) {
    size_t salt = 0;
    float ms = timeit([&] { hipLaunchKernelGGL(k_granule_scatter, dim3(GRID), dim3(256), 0, s0, src, SRC / 16, ring, R / 768, 6, salt); salt += 0; }, 3);
    printf("  R=%6zu MiB  %7.3f ms  %6.2f TB/s input\n", R >> 20, ms, SRC / ms / 1e9);
  }
  printf("F: launch gap: 1000 empty-ish kernels back to back\n");
  {
    float ms = timeit([&] { for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, s0, ring, size_t{0}, ~size_t{0}, sink); }, 2);
    printf("  %.2f us per launch\n", ms);
  }
  return 0;
}
