// SPDX-License-Identifier: Apache-2.0
// gfx950 kernels of the partitioned dense join (engine.hpp dense_part_args): the direct-address table of a dense build side
// (join_args::dense_head) is hundreds of megabytes, and a random 4-byte access to it costs one trip over the fabric (45-60 G
// requests/s on the whole chip; 3.2 ms for C3's 142M in-range probe rows, 2.1 ms for its 50M atomic exchanges). Partitioned by
// key RANGE first (k_radix_scatter's dense mode: 8-byte records {offset, row}), the accesses of one partition fall into ONE
// contiguous slice of the table (<= 1 MB) that stays in the L2 of the XCD whose workgroups walk that partition (blocks b and
// b + 8 share an XCD: observed placement, used for speed only - bench_micro/l2_slice_micro.hip: 266 G random loads/s inside an
// L2-sized window against 57 G/s over 200 MB).
//   build : plain stores head[offset] = row, then one streaming count of the filled entries: as many as records <=> no key
//           repeats (the caller runs the atomic-exchange build when they differ);
//   probe : head[offset] per record; the pairs of a region are staged like the radix join's (k_radix_emit_staged copies them).
// Replaces, for big inner joins on a dense unique build key, k_dense_build / k_dense_count / k_dense_retrieve (kernels.hip); the
// reference structure is cuco::static_multiset insert + count + retrieve (cpp/src/join/hash_join/hash_join.cu:62-149).
#include "engine.hpp"
#include "../common/profiler.hpp"

#include <cudf/join/join.hpp>
#include <cudf/utilities/error.hpp>

#include <algorithm>
#include <cstdlib>

namespace cudf::detail::join {
namespace {

__global__ void k_store_dense_part_args(dense_part_args v, dense_part_args* dst) { *dst = v; }

// One launch serves EIGHT partitions, p0 ... p0 + 7: workgroup b the partition p0 + b % 8 (one XCD's workgroups share one
// partition, i.e. one <= 1 MB slice of the table in that XCD's 4 MB L2), within it the regions s = b / 8, b / 8 + gridDim.x / 8, ...
// The launch boundary is what keeps an XCD on one slice: with one launch for all partitions the workgroups of an XCD drifted over
// many partitions (a workgroup's share of a partition is ~6 batches) and every lookup fetched its sector from beyond L2
// (FETCH_SIZE: 5.6 GB for 1.1 GB of records + 142M lookups).
// (G: partitions per XCD and launch, one after the other - the store pass takes two: its launches are short and it only writes)
template <typename F>
__device__ __forceinline__ void for_each_region(dense_part_args const& a, int p0, int G, F&& f)
{
  int const w = blockIdx.x >> 3, W = gridDim.x >> 3;
  for (int g = 0; g < G; ++g) {
    int const p = p0 + 8 * g + (blockIdx.x & 7);
    if (p >= a.P) return;
    for (int s = w; s < a.S; s += W) f(static_cast<int64_t>(p) * a.S + s);
  }
}

__global__ void __launch_bounds__(256) k_dense_part_store(dense_part_args const* __restrict__ ap, int p0, int G)
{
  dense_part_args const& a = *ap;
  if (*a.overflow != 0) return;
  for_each_region(a, p0, G, [&](int64_t reg) {
    int32_t const cnt   = min(max(a.region_count[reg], 0), static_cast<int32_t>(a.region_cap));
    uint64_t const* rec = a.recs + reg * a.region_cap;
    for (int32_t i0 = 0; i0 < cnt; i0 += 256 * 4) {
      uint64_t v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int32_t const i = i0 + j * 256 + static_cast<int32_t>(threadIdx.x);
        v[j]            = i < cnt ? gload_stream(rec + i) : ~uint64_t{0};  // (read once: must not displace the table slice in L2)
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i0 + j * 256 + static_cast<int32_t>(threadIdx.x) < cnt) gstore(a.head + static_cast<uint32_t>(v[j]), static_cast<int32_t>(v[j] >> 32));
    }
  });
}

// sum of the (non-negative) region counts -> *out (preset to 0): the number of records a scatter produced
__global__ void __launch_bounds__(256) k_sum_region_counts(int32_t const* __restrict__ counts, int64_t n, unsigned long long* out)
{
  unsigned long long c = 0;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    c += static_cast<unsigned long long>(max(counts[i], 0));
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// entries of head[0, n) that hold a row (>= 0) -> *out (preset to 0)
__global__ void __launch_bounds__(256) k_dense_count_filled(int32_t const* __restrict__ head, uint64_t n, unsigned long long* out)
{
  unsigned long long c = 0;
  uint64_t const n4 = n / 4, stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i0 = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i0 < n4; i0 += 4 * stride) {
    i32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = i0 + j * stride < n4 ? gload(reinterpret_cast<i32x4 const*>(head) + i0 + j * stride) : i32x4{-1, -1, -1, -1};
#pragma unroll
    for (int j = 0; j < 4; ++j) c += (v[j].x >= 0) + (v[j].y >= 0) + (v[j].z >= 0) + (v[j].w >= 0);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) c += head[n4 * 4 + threadIdx.x] >= 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// the pairs of ALL regions a workgroup serves (over all launches) go to its own stage, one after the other: one cursor, one
// count per workgroup (pair_counts[b], zero before the first launch)
__global__ void __launch_bounds__(256) k_dense_part_lookup(dense_part_args const* __restrict__ ap, int p0)
{
  dense_part_args const& a = *ap;
  __shared__ unsigned long long s_cursor;
  bool const ok        = *a.overflow == 0;
  int const lane       = threadIdx.x & 63;
  uint64_t const below = (1ull << lane) - 1ull;
  uint64_t* stage      = a.stage + static_cast<int64_t>(blockIdx.x) * a.stage_cap;
  if (threadIdx.x == 0) s_cursor = a.pair_counts[blockIdx.x];
  __syncthreads();
  for_each_region(a, p0, 1, [&](int64_t reg) {
    int32_t const cnt   = ok ? min(max(a.region_count[reg], 0), static_cast<int32_t>(a.region_cap)) : 0;
    uint64_t const* rec = a.recs + reg * a.region_cap;
    constexpr int R = 8;  // records in flight per thread: a region is ~12 records per thread, latency is what a launch pays for
    for (int32_t i0 = 0; i0 < cnt; i0 += 256 * R) {  // (uniform trip count: the ballots below are wave-wide)
      uint64_t v[R];
      int32_t h[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        int32_t const i = i0 + j * 256 + static_cast<int32_t>(threadIdx.x);
        v[j]            = i < cnt ? gload_stream(rec + i) : ~uint64_t{0};  // (read once: must not displace the table slice in L2)
      }
#pragma unroll
      for (int j = 0; j < R; ++j) {
        h[j] = -1;
        if (i0 + j * 256 + static_cast<int32_t>(threadIdx.x) < cnt) h[j] = gload(a.head + static_cast<uint32_t>(v[j]));
      }
      unsigned long long m[R];
      int tot = 0;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        m[j] = __ballot(h[j] >= 0);
        tot += __popcll(m[j]);
      }
      if (tot != 0) {
        unsigned long long pos = 0;
        if (lane == 0) pos = atomicAdd(&s_cursor, static_cast<unsigned long long>(tot));
        pos = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(pos >> 32))) << 32) |
              __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(pos));
#pragma unroll
        for (int j = 0; j < R; ++j) {
          if (h[j] >= 0) {
            uint32_t const prow = static_cast<uint32_t>(static_cast<int64_t>(v[j] >> 32) + a.probe_row_base);
            gstore_stream(stage + pos + __popcll(m[j] & below), static_cast<uint64_t>(prow) | (static_cast<uint64_t>(static_cast<uint32_t>(h[j])) << 32));
          }
          pos += __popcll(m[j]);
        }
      }
    }
  });
  __syncthreads();
  if (threadIdx.x == 0) a.pair_counts[blockIdx.x] = s_cursor;
}

__global__ void k_store_dense_stage_args(dense_stage_args v, dense_stage_args* dst) { *dst = v; }

// The rows' loads of one round, all issued before any of them is used: NARROW (a 4-byte key column) and MASKED (a validity mask) are
// compile-time so that no load sits under a run-time branch, rows past the end read the last row instead of being skipped, and a row
// that fails the range test still reads head[0] (one hot line) - the compiler then counts its waits instead of draining every load:
// 34 `s_waitcnt vmcnt(0)` for 32 loads before, eight dependent round trips per round; now the R key loads, the R mask loads and the R
// table lookups each travel together.
template <bool NARROW, bool MASKED, int R, typename RowOf>
__device__ __forceinline__ void dense_lookup_round(dense_stage_args const& a, int64_t last_row, RowOf row_of, int64_t limit, int32_t (&h)[R])
{
  uint64_t raw[R];
  uint32_t mword[R];
  uint32_t const* keys32 = reinterpret_cast<uint32_t const*>(a.keys);
#pragma unroll
  for (int j = 0; j < R; ++j) {
    int64_t const r = min(row_of(j), last_row);
    if constexpr (NARROW) raw[j] = gload_stream(keys32 + r);
    else raw[j] = gload_stream(a.keys + r);
    if constexpr (MASKED) mword[j] = gload(a.mask + ((a.mask_offset + r) >> 5));
  }
  uint64_t idx[R];
  bool in[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    int64_t const r = row_of(j);
    uint64_t key    = raw[j];
    if constexpr (NARROW) key = a.key_signed ? static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(static_cast<uint32_t>(raw[j])))) : raw[j];
    idx[j] = key - a.dense_lo;
    in[j]  = r < limit && idx[j] < a.dense_range;  // a key outside the build side's range matches nothing
    if constexpr (MASKED) in[j] = in[j] && ((mword[j] >> ((a.mask_offset + r) & 31)) & 1u);
  }
#pragma unroll
  for (int j = 0; j < R; ++j) h[j] = gload(a.head + (in[j] ? idx[j] : 0));
#pragma unroll
  for (int j = 0; j < R; ++j) h[j] = in[j] ? h[j] : -1;
}

// (engine.hpp dense_stage_args) one wave = one contiguous range of probe rows, walked in order, 8 row sets of 64 in flight
template <bool NARROW, bool MASKED>
__global__ void __launch_bounds__(256) k_dense_probe_staged(dense_stage_args const* __restrict__ ap)
{
  dense_stage_args const& a = *ap;
  constexpr int R = 8;
  int const lane = threadIdx.x & 63;
  int64_t const w = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= a.nwaves) return;
  uint64_t const below = (1ull << lane) - 1ull;
  int64_t const begin = w * a.wave_rows, end = min(a.nrows, begin + a.wave_rows);
  uint64_t* stage      = a.stage + w * a.wave_rows;
  unsigned long long npairs = 0;
  for (int64_t base = begin; base < end; base += 64 * R) {
    int32_t h[R];
    dense_lookup_round<NARROW, MASKED, R>(a, end - 1, [&](int j) { return base + j * 64 + lane; }, end, h);
#pragma unroll
    for (int j = 0; j < R; ++j) {
      unsigned long long const m = __ballot(h[j] >= 0);
      if (h[j] >= 0) {
        uint32_t const prow = static_cast<uint32_t>(base + j * 64 + lane + a.probe_row_base);
        gstore_stream(stage + npairs + __popcll(m & below), static_cast<uint64_t>(prow) | (static_cast<uint64_t>(static_cast<uint32_t>(h[j])) << 32));
      }
      npairs += __popcll(m);
    }
  }
  if (lane == 0) a.pair_counts[w] = npairs;
}

// (engine.hpp launch_dense_inrange_sample) out[0] += sampled rows, out[1] += sampled rows whose key is valid and inside the table's range
__global__ void __launch_bounds__(256) k_dense_inrange_sample(dense_stage_args const* __restrict__ ap, int64_t stride, unsigned long long* __restrict__ out)
{
  dense_stage_args const& a = *ap;
  int64_t const r = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * stride;
  bool in = false, sampled = r < a.nrows;
  if (sampled) {
    uint64_t idx;
    if (a.key_width == 4) {
      uint32_t const k32 = gload(reinterpret_cast<uint32_t const*>(a.keys) + r);
      idx = (a.key_signed ? static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(k32))) : static_cast<uint64_t>(k32)) - a.dense_lo;
    } else {
      idx = gload(a.keys + r) - a.dense_lo;
    }
    in = idx < a.dense_range;
    if (in && a.mask != nullptr) in = (gload(a.mask + ((a.mask_offset + r) >> 5)) >> ((a.mask_offset + r) & 31)) & 1u;
  }
  unsigned long long const ns = __popcll(__ballot(sampled)), ni = __popcll(__ballot(in));
  if ((threadIdx.x & 63) == 0 && ns > 0) {
    atomicAdd(out, ns);
    if (ni > 0) atomicAdd(out + 1, ni);
  }
}

// (engine.hpp launch_dense_left_direct)
template <bool NARROW, bool MASKED>
__global__ void __launch_bounds__(256) k_dense_left_direct(dense_stage_args const* __restrict__ ap, size_type* __restrict__ out_probe,
                                                           size_type* __restrict__ out_build)
{
  dense_stage_args const& a = *ap;
  constexpr int R = 8;
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x * R;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * blockDim.x * R; base < a.nrows; base += stride) {
    int32_t h[R];
    dense_lookup_round<NARROW, MASKED, R>(a, a.nrows - 1, [&](int j) { return base + j * 256 + threadIdx.x; }, a.nrows, h);
#pragma unroll
    for (int j = 0; j < R; ++j) {
      int64_t const r = base + j * 256 + threadIdx.x;
      if (r < a.nrows) {
        gstore_stream(out_probe + r, static_cast<size_type>(r + a.probe_row_base));
        gstore_stream(out_build + r, h[j] >= 0 ? static_cast<size_type>(h[j]) : JoinNoMatch);
      }
    }
  }
}

// one launch among the four (NARROW, MASKED) variants of a kernel template
#define CUDF_AMD_DENSE_VARIANT(KERNEL, a, grid, stream, ...)                                                                         \
  do {                                                                                                                               \
    bool const narrow_ = (a).key_width == 4, masked_ = (a).mask != nullptr;                                                          \
    if (narrow_ && masked_) hipLaunchKernelGGL((KERNEL<true, true>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__);                 \
    else if (narrow_) hipLaunchKernelGGL((KERNEL<true, false>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__);                      \
    else if (masked_) hipLaunchKernelGGL((KERNEL<false, true>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__);                      \
    else hipLaunchKernelGGL((KERNEL<false, false>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__);                                  \
  } while (0)

}  // namespace

void launch_dense_inrange_sample(dense_stage_args const& a, dense_stage_args* d_args, int64_t samples, unsigned long long* out, hipStream_t stream)
{
  CUDF_EXPECTS(a.keys != nullptr && out != nullptr && a.nrows >= 1 && samples >= 1 && (a.key_width == 4 || a.key_width == 8), "dense join, probe sample: arguments");
  hipLaunchKernelGGL(k_store_dense_stage_args, dim3(1), dim3(1), 0, stream, a, d_args);
  CUDF_HIP_TRY(hipMemsetAsync(out, 0, 2 * sizeof(unsigned long long), stream));
  int64_t const stride = std::max<int64_t>(1, a.nrows / samples);
  int64_t const n      = (a.nrows + stride - 1) / stride;
  cudf::detail::prof::scope prof_{"join_sample", stream};
  hipLaunchKernelGGL(k_dense_inrange_sample, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, d_args, stride, out);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_dense_left_direct(dense_stage_args const& a, dense_stage_args* d_args, size_type* out_probe, size_type* out_build, hipStream_t stream)
{
  CUDF_EXPECTS(a.keys != nullptr && a.head != nullptr && out_probe != nullptr && out_build != nullptr && a.nrows >= 1 && (a.key_width == 4 || a.key_width == 8),
               "dense join, direct left join: arguments");
  hipLaunchKernelGGL(k_store_dense_stage_args, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_retrieve", stream};
  unsigned const grid = static_cast<unsigned>(std::clamp<int64_t>((a.nrows + 2047) / 2048, 1, 8192));
  CUDF_AMD_DENSE_VARIANT(k_dense_left_direct, a, grid, stream, d_args, out_probe, out_build);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_dense_probe_staged(dense_stage_args const& a, dense_stage_args* d_args, hipStream_t stream)
{
  CUDF_EXPECTS(a.keys != nullptr && a.head != nullptr && a.stage != nullptr && a.pair_counts != nullptr && a.wave_rows >= 64 && a.wave_rows % 64 == 0 &&
                 a.nwaves >= 1 && static_cast<int64_t>(a.nwaves) * a.wave_rows >= a.nrows && (a.key_width == 4 || a.key_width == 8),
               "dense join, ordered probe: arguments");
  hipLaunchKernelGGL(k_store_dense_stage_args, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_count", stream};
  CUDF_AMD_DENSE_VARIANT(k_dense_probe_staged, a, (a.nwaves + 3) / 4, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

int32_t dense_part_grid() { return 2048; }  // 8 workgroups of 256 threads per CU: 256 per XCD, one region of a partition each
int64_t dense_part_regions_per_workgroup(int32_t P, int32_t S)
{
  int64_t const W = dense_part_grid() / 8;
  return static_cast<int64_t>((P + 7) / 8) * ((S + W - 1) / W);  // (P: the partitions that can hold rows, dense_part_args::P_used)
}

void launch_dense_part_store(dense_part_args const& a, dense_part_args* d_args, hipStream_t stream)
{
  CUDF_EXPECTS(a.P >= 1 && a.S >= 1 && a.P_used >= 1 && a.P_used <= a.P && a.recs != nullptr && a.head != nullptr && a.overflow != nullptr,
               "partitioned dense join: arguments");
  hipLaunchKernelGGL(k_store_dense_part_args, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_build", stream};
  static int const G = std::getenv("CUDF_AMD_JOIN_STORE_G") ? std::max(1, std::atoi(std::getenv("CUDF_AMD_JOIN_STORE_G"))) : 2;
  for (int p0 = 0; p0 < a.P_used; p0 += 8 * G) hipLaunchKernelGGL(k_dense_part_store, dim3(dense_part_grid()), dim3(256), 0, stream, d_args, p0, G);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_sum_region_counts(int32_t const* counts, int64_t n, unsigned long long* out, hipStream_t stream)
{
  hipLaunchKernelGGL(k_sum_region_counts, dim3(static_cast<unsigned>(std::clamp<int64_t>((n + 255) / 256, 1, 256))), dim3(256), 0, stream, counts, n, out);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_dense_count_filled(int32_t const* head, uint64_t n, unsigned long long* out, hipStream_t stream)
{
  cudf::detail::prof::scope prof_{"join_build", stream};
  hipLaunchKernelGGL(k_dense_count_filled, dim3(2048), dim3(256), 0, stream, head, n, out);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_dense_part_lookup(dense_part_args const& a, dense_part_args* d_args, hipStream_t stream)
{
  CUDF_EXPECTS(a.P >= 1 && a.S >= 1 && a.recs != nullptr && a.head != nullptr && a.overflow != nullptr && a.stage != nullptr && a.pair_counts != nullptr &&
                 a.stage_cap >= dense_part_regions_per_workgroup(a.P_used, a.S) * a.region_cap && a.P_used >= 1 && a.P_used <= a.P,
               "partitioned dense join: arguments");
  hipLaunchKernelGGL(k_store_dense_part_args, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_count", stream};
  CUDF_HIP_TRY(hipMemsetAsync(a.pair_counts, 0, (static_cast<std::size_t>(dense_part_grid()) + 1) * sizeof(unsigned long long), stream));
  for (int p0 = 0; p0 < a.P_used; p0 += 8) hipLaunchKernelGGL(k_dense_part_lookup, dim3(dense_part_grid()), dim3(256), 0, stream, d_args, p0);
  CUDF_HIP_TRY(hipGetLastError());
}

}  // namespace cudf::detail::join
