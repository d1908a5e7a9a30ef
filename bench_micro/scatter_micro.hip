// How much do unaligned partition runs cost? Each "run" of RUN bytes goes to a pseudo-random destination whose
// start is shifted by SHIFT bytes off 128-byte alignment (16-byte records, one record per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int RUN>
__global__ void k_runscatter(uint4* outp, size_t nslots_mask, size_t n16, int shift16, int mode) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  constexpr int LPR = RUN / 16;
  for (; i < n16; i += stride) {
    size_t run = i / LPR;
    size_t dst_slot = mix64(run) & nslots_mask;   // slot = 512-byte aligned region
    size_t sh = mode == 0 ? 0 : (mode == 1 ? shift16 : (mix64(run * 7 + 1) & 7));  // 0 aligned, 1 fixed shift, 2 random shift (16B units)
    outp[dst_slot * 32 + sh + (i % LPR)] = make_uint4((uint32_t)i, 1u, 2u, 3u);
  }
}
int main() {
  const size_t big = 8ull << 30; uint4* b; CK(hipMalloc(&b, big + 4096));
  size_t n16 = (4ull << 30) / 16;  // write 4 GiB of records into an 8 GiB region of 512-B slots
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto T = [&](auto kern, const char* name, int mode, int shift) {
    hipLaunchKernelGGL(kern, dim3(8192), dim3(256), 0, 0, b, (big / 512) - 1, n16, shift, mode); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(8192), dim3(256), 0, 0, b, (big / 512) - 1, n16, shift, mode);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    printf("%-40s %7.3f ms  %5.2f TB/s\n", name, ms, (4ull << 30) / ms / 1e9);
  };
  T(k_runscatter<128>, "run 128 B aligned", 0, 0);
  T(k_runscatter<128>, "run 128 B shifted +16 B", 1, 1);
  T(k_runscatter<128>, "run 128 B shifted +64 B", 1, 4);
  T(k_runscatter<128>, "run 128 B random 16B-shift", 2, 0);
  T(k_runscatter<256>, "run 256 B aligned", 0, 0);
  T(k_runscatter<256>, "run 256 B shifted +16 B", 1, 1);
  T(k_runscatter<256>, "run 256 B random 16B-shift", 2, 0);
  T(k_runscatter<64>, "run 64 B aligned", 0, 0);
  T(k_runscatter<64>, "run 64 B shifted +16", 1, 1);
  return 0;
}
