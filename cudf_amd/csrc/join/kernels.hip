// SPDX-License-Identifier: Apache-2.0
// gfx950 kernels of the hash-join engine (see engine.hpp).
#include "engine.hpp"
#include "../common/profiler.hpp"

#include <cudf/join/join.hpp>
#include <cudf/utilities/error.hpp>

namespace cudf::detail::join {
namespace {

template <typename T>
__global__ void k_store_args(T v, T* dst)
{
  *dst = v;
}

// 64-bit row hash of the join keys (the engine's own; results do not depend on it). A NULL element hashes
// to a constant so that NULL == NULL rows meet under null_equality::EQUAL.
__device__ __forceinline__ uint64_t join_row_hash(device_table const& t, int64_t i, bool check_nulls)
{
  uint64_t h = 0x9e3779b97f4a7c15ull;
  for (int c = 0; c < t.ncols; ++c) {
    uint64_t bits = 0x6a09e667f3bcc909ull;
    if (!check_nulls || col_is_valid(t.col[c], i)) bits = normalize_key_bits(col_load_bits(t.col[c], i), t.col[c].cls);
    h = mix64(h ^ bits);
  }
  return h;
}
__device__ __forceinline__ uint64_t hash64_single(uint64_t key) { return mix64(0x9e3779b97f4a7c15ull ^ key); }

__device__ __forceinline__ uint64_t home_slot(uint64_t h, uint64_t capacity)
{
  return (static_cast<uint64_t>(static_cast<uint32_t>(h)) * capacity) >> 32;  // capacity < 2^32
}
__device__ __forceinline__ uint64_t make_entry(uint64_t h, int64_t row)
{
  return ((h >> 32) << 32) | static_cast<uint32_t>(row);
}

// ------------------------------------------------------------------ build
template <bool SINGLE64>
__global__ void __launch_bounds__(256) k_build(join_args const* __restrict__ ap)
{
  join_args const& a = *ap;
  int64_t const n      = a.build.nrows;
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  uint64_t const cap   = a.capacity;
  uint64_t const* keys = SINGLE64 ? static_cast<uint64_t const*>(a.build.col[0].head) + a.build.col[0].offset : nullptr;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride) {
    uint64_t h;
    if constexpr (SINGLE64) {
      // with null_equality::UNEQUAL a NULL key row is never inserted
      if (a.check_nulls && a.build.col[0].mask != nullptr && !col_is_valid(a.build.col[0], i)) continue;
      h = hash64_single(gload(keys + i));
    } else {
      // null_equality::UNEQUAL: rows containing a NULL can never match and are not inserted
      // (reference hash_join.cu:77-84, join_common_utils.cuh:36-47)
      if (a.check_nulls && !a.nulls_equal && row_has_null(a.build, i)) continue;
      h = join_row_hash(a.build, i, a.check_nulls);
    }
    uint64_t const entry = make_entry(h, i);
    uint64_t slot        = home_slot(h, cap);
    for (;;) {
      unsigned long long const old =
        atomicCAS(reinterpret_cast<unsigned long long*>(a.table + slot), EMPTY_SLOT, static_cast<unsigned long long>(entry));
      if (old == EMPTY_SLOT) break;
      slot = slot + 1 == cap ? 0 : slot + 1;
    }
  }
}

// ------------------------------------------------------------------ probe: count pass
// One workgroup per contiguous chunk of probe rows; R rows per lane with their first slot loads issued together
// (the probe is a chain of dependent random 8-byte reads: bytes in flight are what buys throughput). Writes the
// per-row match cache and the per-workgroup pair count.
template <bool SINGLE64>
__global__ void __launch_bounds__(256) k_probe_count(join_args const* __restrict__ ap)
{
  join_args const& a = *ap;
  __shared__ unsigned long long s_total;
  int64_t const n    = a.probe.nrows;
  uint64_t const cap = a.capacity;
  int const kind     = a.kind;
  uint64_t const* pkeys = SINGLE64 ? static_cast<uint64_t const*>(a.probe.col[0].head) + a.probe.col[0].offset : nullptr;
  uint64_t const* bkeys = SINGLE64 ? static_cast<uint64_t const*>(a.build.col[0].head) + a.build.col[0].offset : nullptr;
  int64_t const begin = static_cast<int64_t>(blockIdx.x) * a.chunk;
  int64_t const end   = min(n, begin + a.chunk);
  bitmask_type const* probe_mask = (SINGLE64 && a.check_nulls) ? a.probe.col[0].mask : nullptr;
  int64_t const probe_off        = SINGLE64 ? a.probe.col[0].offset : 0;
  if (threadIdx.x == 0) s_total = 0;
  __syncthreads();
  constexpr int R = 4;
  unsigned long long local_count = 0;
  for (int64_t base = begin; base < end; base += static_cast<int64_t>(blockDim.x) * R) {
    int64_t j[R];
    bool live[R], active[R];
    uint64_t h[R], pkey[R], slot[R], e[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      j[k]      = base + static_cast<int64_t>(k) * blockDim.x + threadIdx.x;
      live[k]   = j[k] < end;
      active[k] = live[k];
      h[k] = pkey[k] = 0;
      if (live[k]) {
        if constexpr (SINGLE64) {
          if (probe_mask != nullptr && !((gload(probe_mask + ((probe_off + j[k]) >> 5)) >> ((probe_off + j[k]) & 31)) & 1u)) {
            active[k] = false;  // NULL probe key under null_equality::UNEQUAL matches nothing
          } else {
            pkey[k] = gload(pkeys + j[k]);
            h[k]    = hash64_single(pkey[k]);
          }
        } else if (a.check_nulls && !a.nulls_equal && row_has_null(a.probe, j[k])) {
          active[k] = false;  // matches nothing (null_equality::UNEQUAL)
        } else {
          h[k] = join_row_hash(a.probe, j[k], a.check_nulls);
        }
      }
      slot[k] = active[k] ? home_slot(h[k], cap) : 0;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) e[k] = active[k] ? gload(a.table + slot[k]) : EMPTY_SLOT;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (!live[k]) continue;
      uint32_t const tag = static_cast<uint32_t>(h[k] >> 32);
      uint32_t first     = MATCH_NONE;
      unsigned int cnt   = 0;
      uint64_t ee        = e[k];
      uint64_t sl        = slot[k];
      while (ee != EMPTY_SLOT) {
        if (static_cast<uint32_t>(ee >> 32) == tag) {
          size_type const brow = static_cast<size_type>(static_cast<uint32_t>(ee));
          bool match;
          if constexpr (SINGLE64) match = gload(bkeys + brow) == pkey[k];
          else match = rows_equal(a.probe, j[k], a.build, brow, a.nulls_equal != 0);
          if (match) {
            if (cnt == 0) first = static_cast<uint32_t>(brow);
            ++cnt;
          }
        }
        sl = sl + 1 == cap ? 0 : sl + 1;
        ee = gload(a.table + sl);
      }
      gstore(a.match_cache + j[k], cnt > 1 ? (first | MATCH_MULTI) : first);
      local_count += (cnt == 0 && kind != 0) ? 1u : cnt;  // left/full joins emit lonely probe rows once
    }
  }
  for (int o = 32; o > 0; o >>= 1) local_count += __shfl_down(local_count, o);
  if ((threadIdx.x & 63) == 0 && local_count) atomicAdd(&s_total, local_count);
  __syncthreads();
  if (threadIdx.x == 0) a.block_counts[blockIdx.x] = s_total;
}

// ------------------------------------------------------------------ probe: retrieve pass
// Streams the match cache. Rows with zero or one match are emitted straight from it; rows with several matches walk
// the table again. Output slots come from a workgroup-local LDS cursor (64-lane ballot + popcount prefix, one LDS
// atomic per wave and round) on top of the chunk's exclusive offset from the count pass: no global atomics (a
// single global counter was measured 16x slower than the whole count pass: same-address atomics serialise).
template <bool SINGLE64>
__global__ void __launch_bounds__(256) k_probe_retrieve(join_args const* __restrict__ ap)
{
  join_args const& a = *ap;
  __shared__ unsigned long long s_cursor;
  int64_t const n    = a.probe.nrows;
  uint64_t const cap = a.capacity;
  int const kind     = a.kind;
  uint64_t const* pkeys = SINGLE64 ? static_cast<uint64_t const*>(a.probe.col[0].head) + a.probe.col[0].offset : nullptr;
  uint64_t const* bkeys = SINGLE64 ? static_cast<uint64_t const*>(a.build.col[0].head) + a.build.col[0].offset : nullptr;
  int64_t const begin = static_cast<int64_t>(blockIdx.x) * a.chunk;
  int64_t const end   = min(n, begin + a.chunk);
  if (threadIdx.x == 0) s_cursor = a.block_counts[blockIdx.x];
  __syncthreads();
  int const lane = threadIdx.x & 63;
  // wave-level slot allocation for the lanes with `want`
  auto emit = [&](bool want, size_type prow, size_type brow) {
    unsigned long long const ballot = __ballot(want);
    if (ballot == 0) return;
    int const lead = __ffsll(static_cast<long long>(ballot)) - 1;
    int const rank = __popcll(ballot & ((1ull << lane) - 1));
    unsigned long long base = 0;
    if (lane == lead) base = atomicAdd(&s_cursor, static_cast<unsigned long long>(__popcll(ballot)));
    base = __shfl(base, lead);
    if (want) {
      uint64_t const o = base + rank;
      if (o < a.out_capacity) {
        gstore(a.out_probe + o, prow);
        gstore(a.out_build + o, brow);
      }
      if (kind == 2 && brow != JoinNoMatch) gstore(a.build_matched + brow, uint8_t{1});
    }
  };
  for (int64_t j0 = begin; j0 < end; j0 += blockDim.x) {  // every lane of a wave iterates together (ballots)
    int64_t const j   = j0 + threadIdx.x;
    bool const live   = j < end;
    uint32_t const c  = live ? gload(a.match_cache + j) : MATCH_NONE;
    bool const none   = c == MATCH_NONE;
    bool const multi  = !none && (c & MATCH_MULTI);
    // zero or one match: straight from the cache
    emit(live && !none && !multi, static_cast<size_type>(j), static_cast<size_type>(c & ~MATCH_MULTI));
    if (kind != 0) emit(live && none, static_cast<size_type>(j), JoinNoMatch);
    // several matches: walk the table again (rare with unique build keys)
    if (__any(multi)) {
      uint64_t h = 0, pkey = 0;
      if (multi) {
        if constexpr (SINGLE64) {
          pkey = gload(pkeys + j);
          h    = hash64_single(pkey);
        } else {
          h = join_row_hash(a.probe, j, a.check_nulls);
        }
      }
      uint64_t slot      = multi ? home_slot(h, cap) : 0;
      uint32_t const tag = static_cast<uint32_t>(h >> 32);
      bool walking       = multi;
      while (__any(walking)) {
        bool match     = false;
        size_type brow = 0;
        if (walking) {
          uint64_t const e = gload(a.table + slot);
          if (e == EMPTY_SLOT) {
            walking = false;
          } else {
            if (static_cast<uint32_t>(e >> 32) == tag) {
              brow = static_cast<size_type>(static_cast<uint32_t>(e));
              if constexpr (SINGLE64) match = gload(bkeys + brow) == pkey;
              else match = rows_equal(a.probe, j, a.build, brow, a.nulls_equal != 0);
            }
            slot = slot + 1 == cap ? 0 : slot + 1;
          }
        }
        emit(match, static_cast<size_type>(j), brow);
      }
    }
  }
}

// exclusive scan of the per-block pair counts (nblocks <= 65536): one workgroup
__global__ void __launch_bounds__(1024) k_scan_counts(unsigned long long* counts, int32_t nblocks)
{
  __shared__ unsigned long long wave_tot[16];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < nblocks; base += 1024) {
    int const i                = base + threadIdx.x;
    unsigned long long const v = i < nblocks ? counts[i] : 0;
    unsigned long long inc     = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      unsigned long long const t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    unsigned long long off = carry;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (i < nblocks) counts[i] = off + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[nblocks] = carry;
}

// full join: rows of the build side no probe row matched; one global atomic per workgroup
__global__ void __launch_bounds__(256) k_complement(join_args const* __restrict__ ap)
{
  join_args const& a = *ap;
  __shared__ unsigned long long s_cursor;
  __shared__ unsigned int s_count;
  int64_t const n      = a.build.nrows;
  int64_t const chunk  = (n + gridDim.x - 1) / gridDim.x;
  int64_t const begin  = static_cast<int64_t>(blockIdx.x) * chunk;
  int64_t const end    = min(n, begin + chunk);
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  unsigned int mine = 0;
  for (int64_t r = begin + threadIdx.x; r < end; r += blockDim.x) mine += a.build_matched[r] == 0;
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_count, mine);
  __syncthreads();
  if (threadIdx.x == 0) s_cursor = s_count ? atomicAdd(a.total, static_cast<unsigned long long>(s_count)) : 0ull;
  __syncthreads();
  if (s_count == 0) return;
  int const lane = threadIdx.x & 63;
  for (int64_t r0 = begin; r0 < end; r0 += blockDim.x) {
    int64_t const r   = r0 + threadIdx.x;
    bool const lonely = r < end && a.build_matched[r] == 0;
    unsigned long long const ballot = __ballot(lonely);
    if (ballot == 0) continue;
    int const lead = __ffsll(static_cast<long long>(ballot)) - 1;
    int const rank = __popcll(ballot & ((1ull << lane) - 1));
    unsigned long long base = 0;
    if (lane == lead) base = atomicAdd(&s_cursor, static_cast<unsigned long long>(__popcll(ballot)));
    base = __shfl(base, lead);
    if (lonely) {
      uint64_t const o = base + rank;
      if (o < a.out_capacity) {
        gstore(a.out_probe + o, JoinNoMatch);
        gstore(a.out_build + o, static_cast<size_type>(r));
      }
    }
  }
}

unsigned grid_for(int64_t n)
{
  int64_t const blocks = (n + 255) / 256;
  return static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(blocks, 256 * 16)));
}
}  // namespace

void launch_build(join_args const& a, join_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<join_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_build", stream};
  if (a.single64) hipLaunchKernelGGL(k_build<true>, dim3(grid_for(a.build.nrows)), dim3(256), 0, stream, d_args);
  else hipLaunchKernelGGL(k_build<false>, dim3(grid_for(a.build.nrows)), dim3(256), 0, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}
void launch_count(join_args const& a, join_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<join_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_count", stream};
  if (a.single64) hipLaunchKernelGGL(k_probe_count<true>, dim3(a.nblocks), dim3(256), 0, stream, d_args);
  else hipLaunchKernelGGL(k_probe_count<false>, dim3(a.nblocks), dim3(256), 0, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}
void launch_scan(join_args const& a, hipStream_t stream)
{
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, stream, a.block_counts, a.nblocks);
  CUDF_HIP_TRY(hipGetLastError());
}
void launch_retrieve(join_args const& a, join_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<join_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_retrieve", stream};
  if (a.single64) hipLaunchKernelGGL(k_probe_retrieve<true>, dim3(a.nblocks), dim3(256), 0, stream, d_args);
  else hipLaunchKernelGGL(k_probe_retrieve<false>, dim3(a.nblocks), dim3(256), 0, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}
void launch_complement(join_args const& a, join_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<join_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_complement", stream};
  hipLaunchKernelGGL(k_complement, dim3(grid_for(a.build.nrows)), dim3(256), 0, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}
}  // namespace cudf::detail::join
