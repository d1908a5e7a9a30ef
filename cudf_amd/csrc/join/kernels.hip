// SPDX-License-Identifier: Apache-2.0
// gfx950 kernels of the hash-join engine (see engine.hpp).
#include <mutex>
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

__device__ __forceinline__ uint64_t home_slot(uint64_t h, join_args const& a)
{
  return (static_cast<uint64_t>(static_cast<uint32_t>(h)) * a.capacity) >> 32;  // capacity < 2^32
}
__device__ __forceinline__ uint64_t make_entry(uint64_t h, int64_t row)
{
  return ((h >> 32) << 32) | static_cast<uint32_t>(row);
}

// Probe sequence of a row hash, in STEPS of 16 bytes (two 8-byte slots, or one inline-key slot): SEQ_LINEAR consecutive
// steps from the home step, then hops of a key-dependent stride coprime to the number of steps (the hops reach every
// step). Plain linear probing lets ONE heavily duplicated build key poison its neighbourhood: 100,000 equal keys form a
// 100,000-slot run that every probe with a home slot inside it walks to the end (0.25 % of 100M probes: 1 s). With the
// hops, the duplicates beyond the window sit on their key's own trail. Build, count and retrieve passes walk the same
// sequence; an entry always takes the first empty slot of its sequence, so a walk may stop at the first empty slot.
constexpr uint32_t SEQ_LINEAR = 16;
__device__ __noinline__ uint64_t seq_stride(uint64_t h, uint64_t nsteps)
{
  if (nsteps <= 2) return 1;
  uint64_t s = 1 + mix64(h ^ 0x2545f4914f6cdd1dull) % (nsteps - 1);
  for (;;) {
    uint64_t x = s, y = nsteps;
    while (y != 0) {
      uint64_t const t = x % y;
      x                = y;
      y                = t;
    }
    if (x == 1) return s;
    s = s + 1 >= nsteps ? 1 : s + 1;
  }
}
// Advances from step number n (the home step st0 is number 0) and updates n; `stride` caches seq_stride (0 = not computed
// yet). Hop positions that fall back into the linear window were visited already: they keep their number but are
// passed over (a probe that walked them twice counted their matches twice). Tiny tables stay linear.
__device__ __forceinline__ uint64_t seq_next(uint64_t st, uint32_t& n, uint64_t h, uint64_t nsteps, uint64_t& stride, uint64_t st0)
{
  if (n + 1 < SEQ_LINEAR || nsteps <= 4 * SEQ_LINEAR) {
    ++n;
    return st + 1 == nsteps ? 0 : st + 1;
  }
  if (stride == 0) stride = seq_stride(h, nsteps);
  for (;;) {
    st += stride;
    if (st >= nsteps) st -= nsteps;
    ++n;
    uint64_t const d = st >= st0 ? st - st0 : st + nsteps - st0;
    if (d >= SEQ_LINEAR) return st;
  }
}
// position number n of the sequence that starts at step st0 (n: a number seq_next stopped at)
__device__ __forceinline__ uint64_t seq_at(uint64_t st0, uint32_t n, uint64_t h, uint64_t nsteps, uint64_t& stride)
{
  if (n < SEQ_LINEAR || nsteps <= 4 * SEQ_LINEAR) return (st0 + n) % nsteps;
  if (stride == 0) stride = seq_stride(h, nsteps);
  // n < 2^20, stride < 2^32: no overflow
  return (static_cast<uint64_t>(n - (SEQ_LINEAR - 1)) * stride + (st0 + SEQ_LINEAR - 1) % nsteps) % nsteps;
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
    uint64_t const home  = home_slot(h, a);
    constexpr int sw     = 1;
    constexpr bool pairs = true;                         // 8-byte slots: a step is an aligned pair of slots
    uint64_t const nsteps = pairs ? cap >> 1 : cap;      // (capacity is even)
    uint64_t const st0   = pairs ? home >> 1 : home;
    uint64_t st = st0, stride = 0, slot = home;
    uint32_t n  = 0;                                     // step number; the slots of steps [0, n) are known to be occupied
    // The home slot is tried with the compare-and-swap itself (empty three times out of four); further slots are read
    // first and only an empty one is claimed: a walk over occupied slots costs loads, not atomics.
    bool blind = true;
    // Hints {44 hash bits | n}: "the first n steps of this hash's sequence are full". An insert that has walked
    // BUILD_SKIP_AFTER steps looks its hint up and jumps ahead; one that ended beyond that leaves a hint. Hints are
    // advisory: any value ever stored is true (slots are never emptied during the build), a racing store can only lose
    // some of the shortcut. (Two different heavy keys that agree in 44 hash bits would share hints: 2^-44 per pair.)
    uint64_t* const hint    = a.build_skip + ((h >> 12) & (BUILD_SKIP_ENTRIES - 1));
    uint64_t const hint_tag = (h >> 20) << 20;
    for (;;) {
      bool done = false;
      for (int q = (pairs && n == 0) ? static_cast<int>(home & 1) : 0; q < (pairs ? 2 : 1); ++q) {
        slot                  = pairs ? st * 2 + q : st;
        uint64_t* const where = a.table + slot * sw;
        if (blind || __hip_atomic_load(where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == EMPTY_SLOT) {
          blind = false;
          if (atomicCAS(reinterpret_cast<unsigned long long*>(where), EMPTY_SLOT, static_cast<unsigned long long>(entry)) == EMPTY_SLOT) {
            done = true;
            break;
          }
        }
      }
      if (done) break;
      uint32_t const before = n;
      st = seq_next(st, n, h, nsteps, stride, st0);
      if ((before / BUILD_SKIP_AFTER) != (n / BUILD_SKIP_AFTER)) {  // a long walk (many equal keys): is a longer full prefix known?
        uint64_t const e = __hip_atomic_load(hint, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t const d = static_cast<uint32_t>(e & 0xfffffull);
        if ((e >> 20 << 20) == hint_tag && d > n) {
          n  = d;
          st = seq_at(st0, n, h, nsteps, stride);
        }
      }
    }
    if (n >= BUILD_SKIP_AFTER && n < 0xfffffu) {
      uint64_t const e = __hip_atomic_load(hint, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((e >> 20 << 20) != hint_tag || (e & 0xfffffull) < n) __hip_atomic_store(hint, hint_tag | n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------ probe: count pass
// One workgroup per contiguous chunk of probe rows; R rows per lane with their first loads issued together. Writes the per-row match cache and the
// per-workgroup pair count.
// MODE 0: any key columns (row hash, rows_equal).
// MODE 1: one 8-byte integer key whose NULLs never match, 8-byte slots {tag | row}: candidates are verified against
//         the build key column.
// The pass is bound by the NUMBER of random memory requests (1.2 G requests in 18.6 ms = 66 G/s, the rate the
// random-load microbenchmark reaches), not by their dependency chains: reading a 64-byte window of slots per row with
// four 16-byte loads measured SLOWER (21.9 ms). So 8-byte slots are walked in aligned PAIRS - one 16-byte request
// covers two slots - and the planner keeps the table at load <= 0.25, where a walk to the first empty slot is 1.4
// slots on average instead of 2.5.
constexpr int MODE_GENERIC = 0, MODE_KEY64 = 1;

template <int MODE>
__global__ void __launch_bounds__(256) k_probe_count(join_args const* __restrict__ ap)
{
  join_args const& a = *ap;
  __shared__ unsigned long long s_total;
  constexpr bool KEY64 = MODE != MODE_GENERIC;
  int64_t const n    = a.probe.nrows;
  uint64_t const cap = a.capacity;
  int const kind     = a.kind;
  constexpr bool pairs = true;  // 8-byte slots, walked in aligned pairs
  uint64_t const nsteps = pairs ? cap >> 1 : cap;  // 16-byte steps in the table (capacity is even)
  uint64_t const* pkeys = KEY64 ? static_cast<uint64_t const*>(a.probe.col[0].head) + a.probe.col[0].offset : nullptr;
  uint64_t const* bkeys = KEY64 ? static_cast<uint64_t const*>(a.build.col[0].head) + a.build.col[0].offset : nullptr;
  u64x2 const* table16  = reinterpret_cast<u64x2 const*>(a.table);
  bitmask_type const* probe_mask = (KEY64 && a.check_nulls) ? a.probe.col[0].mask : nullptr;
  int64_t const probe_off        = KEY64 ? a.probe.col[0].offset : 0;
  if (threadIdx.x == 0) s_total = 0;
  __syncthreads();
  constexpr int R = 4;
  unsigned long long local_count = 0;
  {
    int64_t const begin = static_cast<int64_t>(blockIdx.x) * a.chunk;
    int64_t const end   = min(n, begin + a.chunk);
    for (int64_t base = begin; base < end; base += static_cast<int64_t>(blockDim.x) * R) {
      int64_t j[R];
      bool live[R], active[R];
      uint64_t h[R], pkey[R], slot[R];
      u64x2 v0[R];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        j[k]      = base + static_cast<int64_t>(k) * blockDim.x + threadIdx.x;
        live[k]   = j[k] < end;
        active[k] = live[k];
        h[k] = pkey[k] = 0;
        if (live[k]) {
          if constexpr (KEY64) {
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
        slot[k] = active[k] ? home_slot(h[k], a) : 0;
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        v0[k] = u64x2{EMPTY_SLOT, EMPTY_SLOT};
        if (active[k]) v0[k] = gload(table16 + (pairs ? slot[k] >> 1 : slot[k]));
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (!live[k]) continue;
        uint32_t const tag = static_cast<uint32_t>(h[k] >> 32);
        uint32_t first     = MATCH_NONE;
        unsigned int cnt   = 0;
        // one table entry: false at the first empty slot
        auto visit = [&](uint64_t ee) -> bool {
          if (ee == EMPTY_SLOT) return false;
          bool match = false;
          if (static_cast<uint32_t>(ee >> 32) == tag) {
            size_type const brow = static_cast<size_type>(static_cast<uint32_t>(ee));
            if constexpr (MODE == MODE_KEY64) match = gload(bkeys + brow) == pkey[k];
            else match = rows_equal(a.probe, j[k], a.build, brow, a.nulls_equal != 0);
          }
          if (match) {
            if (cnt == 0) first = static_cast<uint32_t>(ee);
            ++cnt;
          }
          return true;
        };
        if (active[k]) {
          u64x2 v     = v0[k];
          uint64_t st = pairs ? slot[k] >> 1 : slot[k], stride = 0;
          uint32_t ns = 0;                       // step number along the key's probe sequence (seq_next)
          bool skip   = pairs && (slot[k] & 1);  // the home slot is the second of its pair
          for (;;) {
            if (!skip && !visit(v.x)) break;
            skip = false;
            if (!visit(v.y)) break;
            st = seq_next(st, ns, h[k], nsteps, stride, pairs ? slot[k] >> 1 : slot[k]);
            v  = gload(table16 + st);
          }
        }
        gstore(a.match_cache + j[k], cnt > 1 ? (first | MATCH_MULTI) : first);
        unsigned int const emitted = (cnt == 0 && kind != 0) ? 1u : cnt;  // left/full joins emit lonely probe rows once
        if (a.row_counts != nullptr) gstore(a.row_counts + j[k], static_cast<size_type>(emitted));
        local_count += emitted;
      }
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
template <int MODE>
__global__ void __launch_bounds__(256) k_probe_retrieve(join_args const* __restrict__ ap)
{
  join_args const& a = *ap;
  __shared__ unsigned long long s_cursor;
  int64_t const n    = a.probe.nrows;
  uint64_t const cap = a.capacity;
  int const kind     = a.kind;
  constexpr bool KEY64 = MODE != MODE_GENERIC;
  constexpr int sw   = 1;
  uint64_t const* pkeys = KEY64 ? static_cast<uint64_t const*>(a.probe.col[0].head) + a.probe.col[0].offset : nullptr;
  uint64_t const* bkeys = KEY64 ? static_cast<uint64_t const*>(a.build.col[0].head) + a.build.col[0].offset : nullptr;
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
  {
  int64_t const begin = static_cast<int64_t>(blockIdx.x) * a.chunk;
  int64_t const end   = min(n, begin + a.chunk);
  for (int64_t j0 = begin; j0 < end; j0 += blockDim.x) {  // every lane of a wave iterates together (ballots)
    int64_t const j   = j0 + threadIdx.x;
    bool const live   = j < end;
    uint32_t const c  = live ? gload(a.match_cache + j) : MATCH_NONE;
    bool const none   = c == MATCH_NONE;
    bool const multi  = !none && (c & MATCH_MULTI);
    size_type prow    = static_cast<size_type>(j + a.probe_row_base);
    uint64_t pkey     = 0;
    // zero or one match: straight from the cache
    emit(live && !none && !multi, prow, static_cast<size_type>(c & ~MATCH_MULTI));
    if (kind != 0) emit(live && none, prow, JoinNoMatch);
    // several matches: walk the table again (rare with unique build keys)
    if (__any(multi)) {
      uint64_t h = 0;
      if (multi) {
        if constexpr (KEY64) {
          pkey = gload(pkeys + j);
          h = hash64_single(pkey);
        } else {
          h = join_row_hash(a.probe, j, a.check_nulls);
        }
      }
      uint64_t slot      = multi ? home_slot(h, a) : 0;
      uint32_t const tag = static_cast<uint32_t>(h >> 32);
      bool walking       = multi;
      constexpr bool pairs = true;  // the sequence advances in 16-byte steps: aligned pairs of 8-byte slots
      uint64_t const nsteps = pairs ? cap >> 1 : cap;
      uint64_t const st0 = pairs ? slot >> 1 : slot;
      uint64_t stride    = 0;
      uint32_t ns        = 0;
      while (__any(walking)) {
        bool match     = false;
        size_type brow = 0;
        if (walking) {
          uint64_t const e = gload(a.table + slot * sw);
          if (e == EMPTY_SLOT) {
            walking = false;
          } else {
            brow = static_cast<size_type>(static_cast<uint32_t>(e));
            if (static_cast<uint32_t>(e >> 32) == tag) {
              if constexpr (MODE == MODE_KEY64) match = gload(bkeys + brow) == pkey;
              else match = rows_equal(a.probe, j, a.build, brow, a.nulls_equal != 0);
            }
            if (pairs && (slot & 1) == 0) {
              slot += 1;
            } else {
              uint64_t const st = seq_next(pairs ? slot >> 1 : slot, ns, h, nsteps, stride, st0);
              slot = pairs ? st * 2 : st;
            }
          }
        }
        emit(match, prow, brow);
      }
    }
  }
  }
}


// ------------------------------------------------------------------ dense build keys: direct-address table (engine.hpp)
template <bool SIGNED>
__global__ void __launch_bounds__(256) k_key_minmax(join_args const* __restrict__ ap, uint64_t* out)
{
  join_args const& a = *ap;
  using T = std::conditional_t<SIGNED, long long, unsigned long long>;
  __shared__ T s_lo[4], s_hi[4];
  int64_t const n      = a.build.nrows;
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  bool const narrow    = a.build.col[0].width == 4;  // a 4-byte integer key column: widened like the join's scatter does
  uint64_t const* keys = static_cast<uint64_t const*>(a.build.col[0].head) + (narrow ? 0 : a.build.col[0].offset);
  uint32_t const* keys32 = static_cast<uint32_t const*>(a.build.col[0].head) + a.build.col[0].offset;
  bool const masked    = a.check_nulls && a.build.col[0].mask != nullptr;
  T lo = SIGNED ? static_cast<T>(INT64_MAX) : static_cast<T>(UINT64_MAX), hi = SIGNED ? static_cast<T>(INT64_MIN) : T{0};
  constexpr int R = 4;  // loads in flight per thread
  for (int64_t i0 = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i0 < n; i0 += R * stride) {
    T k[R];
    bool ok[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      int64_t const i = i0 + j * stride;
      ok[j]           = i < n;
      if (narrow) {
        uint32_t const k32 = ok[j] ? gload(keys32 + i) : 0u;
        k[j]               = SIGNED ? static_cast<T>(static_cast<int32_t>(k32)) : static_cast<T>(k32);
      } else {
        k[j] = ok[j] ? static_cast<T>(gload(keys + i)) : T{0};
      }
      ok[j]           = ok[j] && !(masked && !col_is_valid(a.build.col[0], i));
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      lo = ok[j] && k[j] < lo ? k[j] : lo;
      hi = ok[j] && k[j] > hi ? k[j] : hi;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    T const l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) {
    s_lo[threadIdx.x >> 6] = lo;
    s_hi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      lo = s_lo[w] < lo ? s_lo[w] : lo;
      hi = s_hi[w] > hi ? s_hi[w] : hi;
    }
    atomicMin(reinterpret_cast<T*>(out), lo);
    atomicMax(reinterpret_cast<T*>(out) + 1, hi);
  }
}

__global__ void __launch_bounds__(256) k_dense_build(join_args const* __restrict__ ap)
{
  join_args const& a   = *ap;
  int64_t const n      = a.build.nrows;
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  uint64_t const* keys = static_cast<uint64_t const*>(a.build.col[0].head) + a.build.col[0].offset;
  bool const masked    = a.check_nulls && a.build.col[0].mask != nullptr;
  bool dup             = false;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride) {
    if (masked && !col_is_valid(a.build.col[0], i)) continue;  // a NULL key is never inserted (null_equality::UNEQUAL)
    uint64_t const idx = gload(keys + i) - a.dense_lo;         // < dense_range: lo / range come from these very keys
    int32_t const old  = atomicExch(a.dense_head + idx, static_cast<int32_t>(i));
    gstore(a.dense_next + i, old);
    dup = dup || old >= 0;
  }
  if (__any(dup) && (threadIdx.x & 63) == 0) gstore(a.dense_dups, 1);
}

// count pass over a dense table: same outputs as k_probe_count (match cache, per-workgroup pair counts, optional row counts)
__global__ void __launch_bounds__(256) k_dense_count(join_args const* __restrict__ ap)
{
  join_args const& a = *ap;
  __shared__ unsigned long long s_total;
  int64_t const n       = a.probe.nrows;
  int const kind        = a.kind;
  uint64_t const* pkeys = static_cast<uint64_t const*>(a.probe.col[0].head) + a.probe.col[0].offset;
  bitmask_type const* probe_mask = a.check_nulls ? a.probe.col[0].mask : nullptr;
  int64_t const probe_off        = a.probe.col[0].offset;
  uint64_t const lo = a.dense_lo, range = a.dense_range;
  bool const dups   = a.dense_has_dups != 0;
  if (threadIdx.x == 0) s_total = 0;
  __syncthreads();
  constexpr int R = 4;
  unsigned long long local_count = 0;
  int64_t const begin = static_cast<int64_t>(blockIdx.x) * a.chunk, end = min(n, begin + a.chunk);
  for (int64_t base = begin; base < end; base += static_cast<int64_t>(blockDim.x) * R) {
    int64_t j[R];
    bool live[R], in[R];
    uint64_t idx[R];
    int32_t head[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      j[k]    = base + static_cast<int64_t>(k) * blockDim.x + threadIdx.x;
      live[k] = j[k] < end;
      in[k]   = false;
      idx[k]  = 0;
      if (live[k]) {
        bool valid = true;
        if (probe_mask != nullptr) valid = (gload(probe_mask + ((probe_off + j[k]) >> 5)) >> ((probe_off + j[k]) & 31)) & 1u;
        idx[k] = gload(pkeys + j[k]) - lo;
        in[k]  = valid && idx[k] < range;  // a key outside the build side's range matches nothing and reads nothing
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) head[k] = in[k] ? gload(a.dense_head + idx[k]) : -1;
    // (some build key repeats: dense_head holds the OFFSETS of the keys' row lists - engine.hpp - and a key's row count is the
    // difference to its neighbour; a chain walk per probe row took 149 ms for 1000 probes of a key with 100,000 build rows)
    int32_t next_off[R];
#pragma unroll
    for (int k = 0; k < R; ++k) next_off[k] = (dups && in[k]) ? gload(a.dense_head + idx[k] + 1) : 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (!live[k]) continue;
      unsigned int const cnt = dups ? (in[k] ? static_cast<unsigned int>(next_off[k] - head[k]) : 0u) : (head[k] >= 0 ? 1u : 0u);
      uint32_t const what    = dups ? static_cast<uint32_t>(idx[k]) : static_cast<uint32_t>(head[k]);  // the key's list / its one build row
      gstore(a.match_cache + j[k], cnt == 0 ? MATCH_NONE : (what | (cnt > 1 ? MATCH_MULTI : 0u)));
      unsigned int const emitted = (cnt == 0 && kind != 0) ? 1u : cnt;
      if (a.row_counts != nullptr) gstore(a.row_counts + j[k], static_cast<size_type>(emitted));
      local_count += emitted;
    }
  }
  for (int o = 32; o > 0; o >>= 1) local_count += __shfl_down(local_count, o);
  if ((threadIdx.x & 63) == 0 && local_count) atomicAdd(&s_total, local_count);
  __syncthreads();
  if (threadIdx.x == 0) a.block_counts[blockIdx.x] = s_total;
}

// retrieve pass over a dense table: streams the match cache; rows with several matches walk their key's chain
__global__ void __launch_bounds__(256) k_dense_retrieve(join_args const* __restrict__ ap)
{
  join_args const& a = *ap;
  __shared__ unsigned long long s_cursor;
  int64_t const n = a.probe.nrows;
  int const kind  = a.kind;
  if (threadIdx.x == 0) s_cursor = a.block_counts[blockIdx.x];
  __syncthreads();
  int const lane = threadIdx.x & 63;
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
  int64_t const begin = static_cast<int64_t>(blockIdx.x) * a.chunk, end = min(n, begin + a.chunk);
  bool const dups = a.dense_has_dups != 0;
  for (int64_t j0 = begin; j0 < end; j0 += blockDim.x) {
    int64_t const j  = j0 + threadIdx.x;
    bool const live  = j < end;
    uint32_t const c = live ? gload(a.match_cache + j) : MATCH_NONE;
    bool const none  = c == MATCH_NONE;
    bool const multi = !none && (c & MATCH_MULTI);
    size_type const prow = static_cast<size_type>(j + a.probe_row_base);
    int32_t r = static_cast<int32_t>(c & ~MATCH_MULTI);
    if (!dups) {  // unique build keys: the cache holds the one build row
      emit(live && !none, prow, r);
      if (kind != 0) emit(live && none, prow, JoinNoMatch);
      continue;
    }
    // Row lists (some build key repeats): the cache holds the key's index; its rows are dense_next[off, off + cnt).
    uint32_t const off = none ? 0u : static_cast<uint32_t>(gload(a.dense_head + r));
    uint32_t const cnt = none ? 0u : (multi ? static_cast<uint32_t>(gload(a.dense_head + r + 1)) - off : 1u);
    emit(live && !none, prow, none ? JoinNoMatch : gload(a.dense_next + off));
    if (kind != 0) emit(live && none, prow, JoinNoMatch);
    // very long lists (a hot build key): only the output range is reserved here, the pairs are written by the whole grid
    // (k_dense_big_emit) - the rows that hit a hot key may all sit in ONE workgroup's chunk of probe rows
    bool huge = a.big_list != nullptr && cnt > BIG_LIST;
    if (huge) {
      uint32_t const w = atomicAdd(a.big_count, 1u);
      if (w < BIG_LIST_CAP) {
        unsigned long long const base = atomicAdd(&s_cursor, static_cast<unsigned long long>(cnt - 1));
        gstore(reinterpret_cast<u32x4*>(a.big_list) + w, u32x4{static_cast<uint32_t>(prow), off + 1u, cnt - 1u, static_cast<uint32_t>(base)});
        gstore(a.big_list + 4u * BIG_LIST_CAP + w, static_cast<uint32_t>(base >> 32));
      } else {
        huge = false;  // (the list is full: this wave emits the row's pairs itself)
      }
    }
    // short lists: every lane walks its own; long lists: the WAVE emits 64 pairs per step, coalesced
    bool const big = cnt > 64 && !huge;
    uint32_t t     = 1;
    bool walking   = cnt > 1 && !big && !huge;
    while (__any(walking)) {
      int32_t const br = walking ? gload(a.dense_next + off + t) : JoinNoMatch;
      emit(walking, prow, br);
      ++t;
      walking = walking && t < cnt;
    }
    unsigned long long bigm = __ballot(big);
    while (bigm != 0) {
      int const src = __ffsll(static_cast<long long>(bigm)) - 1;
      bigm &= bigm - 1;
      uint32_t const boff = __shfl(off, src), bcnt = __shfl(cnt, src);
      size_type const bp  = __shfl(prow, src);
      for (uint32_t base = 1; base < bcnt; base += 64) {
        uint32_t const tt = base + static_cast<uint32_t>(lane);
        bool const w      = tt < bcnt;
        emit(w, bp, w ? gload(a.dense_next + boff + tt) : JoinNoMatch);
      }
    }
  }
}

// The pairs of the work list of hot-key probe rows (join_args::big_list): every workgroup takes its share of every entry's chunks.
__global__ void __launch_bounds__(256) k_dense_big_emit(join_args const* __restrict__ ap)
{
  join_args const& a   = *ap;
  uint32_t const items = min(gload(a.big_count), BIG_LIST_CAP);
  constexpr uint32_t CH = 2048;  // pairs per chunk
  for (uint32_t it = 0; it < items; ++it) {
    u32x4 const e   = gload(reinterpret_cast<u32x4 const*>(a.big_list) + it);
    uint64_t const base = (static_cast<uint64_t>(gload(a.big_list + 4u * BIG_LIST_CAP + it)) << 32) | e.w;
    uint32_t const chunks = (e.z + CH - 1) / CH;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
      for (uint32_t t = c * CH + threadIdx.x; t < min(e.z, (c + 1) * CH); t += blockDim.x) {
        int32_t const br = gload(a.dense_next + e.y + t);
        uint64_t const o = base + t;
        if (o < a.out_capacity) {
          gstore(a.out_probe + o, static_cast<size_type>(e.x));
          gstore(a.out_build + o, br);
        }
        if (a.kind == 2) gstore(a.build_matched + br, uint8_t{1});
      }
    }
  }
}

// ---- row lists for duplicated dense build keys: counts (k_dense_csr_count) -> exclusive scan in place (k_scan_u32_*) -> lists
// (k_dense_csr_fill). dense_head: [padded range + 1] counts, then offsets; dense_next: the build rows grouped by key.
__global__ void __launch_bounds__(256) k_dense_csr_count(join_args const* __restrict__ ap)
{
  join_args const& a   = *ap;
  int64_t const n      = a.build.nrows;
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  uint64_t const* keys = static_cast<uint64_t const*>(a.build.col[0].head) + a.build.col[0].offset;
  bool const masked    = a.check_nulls && a.build.col[0].mask != nullptr;
  int const lane = threadIdx.x & 63;
  int64_t const rounds = (n + stride - 1) / stride;
  int64_t i            = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  for (int64_t it = 0; it < rounds; ++it, i += stride) {  // (every lane runs every round: the ballots below are wave-wide)
    bool pending       = i < n && !(masked && !col_is_valid(a.build.col[0], i));
    uint64_t const idx = pending ? gload(keys + i) - a.dense_lo : 0;
    // same-address global atomics serialise (a build key with 100,000 rows): the first keys of the wave are counted once per
    // key, by one lane
    for (int round = 0; round < 2; ++round) {
      unsigned long long const todo = __ballot(pending);
      if (todo == 0) break;
      int const lead                = __ffsll(static_cast<long long>(todo)) - 1;
      uint64_t const lidx           = __shfl(static_cast<unsigned long long>(idx), lead);
      unsigned long long const same = __ballot(pending && idx == lidx);
      if (lane == lead) atomicAdd(a.dense_head + lidx, __popcll(same));
      if (idx == lidx) pending = false;
    }
    if (pending) atomicAdd(a.dense_head + idx, 1);
  }
}
__global__ void __launch_bounds__(256) k_dense_csr_fill(join_args const* __restrict__ ap, int32_t* cursor)
{
  join_args const& a   = *ap;
  int64_t const n      = a.build.nrows;
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  uint64_t const* keys = static_cast<uint64_t const*>(a.build.col[0].head) + a.build.col[0].offset;
  bool const masked    = a.check_nulls && a.build.col[0].mask != nullptr;
  int const lane = threadIdx.x & 63;
  int64_t const rounds = (n + stride - 1) / stride;
  int64_t i            = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  for (int64_t it = 0; it < rounds; ++it, i += stride) {
    bool const valid   = i < n && !(masked && !col_is_valid(a.build.col[0], i));
    bool pending       = valid;
    uint64_t const idx = valid ? gload(keys + i) - a.dense_lo : 0;
    int32_t pos        = 0;
    for (int round = 0; round < 2; ++round) {  // (wave-aggregated as in k_dense_csr_count: one atomic per key, ranks by popcount)
      unsigned long long const todo = __ballot(pending);
      if (todo == 0) break;
      int const lead                = __ffsll(static_cast<long long>(todo)) - 1;
      uint64_t const lidx           = __shfl(static_cast<unsigned long long>(idx), lead);
      unsigned long long const same = __ballot(pending && idx == lidx);
      int32_t base                  = 0;
      if (lane == lead) base = atomicAdd(cursor + lidx, __popcll(same));
      base = __shfl(base, lead);
      if (pending && idx == lidx) {
        pos     = base + __popcll(same & ((1ull << lane) - 1));
        pending = false;
      }
    }
    if (pending) pos = atomicAdd(cursor + idx, 1);
    if (valid) gstore(a.dense_next + gload(a.dense_head + idx) + pos, static_cast<int32_t>(i));
  }
}
// Exclusive scan of n uint32 (n a multiple of SCAN_TILE, at most 2^29 + SCAN_TILE) in place, three launches: tile sums, scan of the
// tile sums by one workgroup, tile-local scan + tile offset.
constexpr int SCAN_TILE = 1024 * 8;
__global__ void __launch_bounds__(1024) k_scan_u32_sums(uint32_t const* __restrict__ v, uint32_t* __restrict__ sums)
{
  __shared__ uint32_t s_w[16];
  u32x4 const* p = reinterpret_cast<u32x4 const*>(v + static_cast<int64_t>(blockIdx.x) * SCAN_TILE) + threadIdx.x * 2;
  u32x4 const a = gload(p), b = gload(p + 1);
  uint32_t t    = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int w = 0; w < 16; ++w) tot += s_w[w];
    sums[blockIdx.x] = tot;
  }
}
__global__ void __launch_bounds__(1024) k_scan_u32_tiles(uint32_t* sums, int32_t ntiles)
{
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < ntiles; base += 1024) {
    int const i      = base + threadIdx.x;
    uint32_t const x = i < ntiles ? sums[i] : 0;
    uint32_t inc     = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t const t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_w[w];
    uint32_t const carry = s_carry;
    if (i < ntiles) sums[i] = carry + woff + inc - x;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + woff + inc;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(1024) k_scan_u32_apply(uint32_t* __restrict__ v, uint32_t const* __restrict__ sums)
{
  __shared__ uint32_t s_w[16];
  u32x4* p = reinterpret_cast<u32x4*>(v + static_cast<int64_t>(blockIdx.x) * SCAN_TILE) + threadIdx.x * 2;
  u32x4 a = gload(p), b = gload(p + 1);
  uint32_t const mine = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t const t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  uint32_t run = gload(sums + blockIdx.x) + inc - mine;
  for (int w = 0; w < wave; ++w) run += s_w[w];
  u32x4 oa, ob;
  oa.x = run; run += a.x;
  oa.y = run; run += a.y;
  oa.z = run; run += a.z;
  oa.w = run; run += a.w;
  ob.x = run; run += b.x;
  ob.y = run; run += b.y;
  ob.z = run; run += b.z;
  ob.w = run;
  gstore(p, oa);
  gstore(p + 1, ob);
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

__global__ void __launch_bounds__(256) k_mark_matched(size_type const* right_indices, std::size_t n, uint8_t* build_matched)
{
  for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<std::size_t>(gridDim.x) * blockDim.x) {
    size_type const r = gload(right_indices + i);
    if (r != JoinNoMatch) gstore(build_matched + r, uint8_t{1});
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
void launch_key_minmax(join_args const& a, join_args* d_args, int is_signed, uint64_t* out, hipStream_t stream)
{
  struct range_init { uint64_t lo, hi; };
  range_init const init{is_signed ? static_cast<uint64_t>(INT64_MAX) : UINT64_MAX, is_signed ? static_cast<uint64_t>(INT64_MIN) : uint64_t{0}};
  hipLaunchKernelGGL(k_store_args<range_init>, dim3(1), dim3(1), 0, stream, init, reinterpret_cast<range_init*>(out));
  hipLaunchKernelGGL(k_store_args<join_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_build", stream};
  unsigned const grid = static_cast<unsigned>(std::clamp<int64_t>((a.build.nrows + 4095) / 4096, 1, 4096));
  if (is_signed) hipLaunchKernelGGL(k_key_minmax<true>, dim3(grid), dim3(256), 0, stream, d_args, out);
  else hipLaunchKernelGGL(k_key_minmax<false>, dim3(grid), dim3(256), 0, stream, d_args, out);
  CUDF_HIP_TRY(hipGetLastError());
}
void launch_dense_build(join_args const& a, join_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<join_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_build", stream};
  hipLaunchKernelGGL(k_dense_build, dim3(grid_for(a.build.nrows)), dim3(256), 0, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}
std::size_t dense_csr_entries(uint64_t range) { return static_cast<std::size_t>((range + 1 + SCAN_TILE - 1) / SCAN_TILE) * SCAN_TILE; }
void launch_dense_csr(join_args const& a, join_args* d_args, int32_t* cursor, uint32_t* tile_sums, hipStream_t stream)
{
  // a.dense_head: dense_csr_entries(range) zeroed words; cursor: `range` zeroed words; tile_sums: entries / SCAN_TILE words
  hipLaunchKernelGGL(k_store_args<join_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_build", stream};
  int32_t const ntiles = static_cast<int32_t>(dense_csr_entries(a.dense_range) / SCAN_TILE);
  hipLaunchKernelGGL(k_dense_csr_count, dim3(grid_for(a.build.nrows)), dim3(256), 0, stream, d_args);
  hipLaunchKernelGGL(k_scan_u32_sums, dim3(ntiles), dim3(1024), 0, stream, reinterpret_cast<uint32_t const*>(a.dense_head), tile_sums);
  hipLaunchKernelGGL(k_scan_u32_tiles, dim3(1), dim3(1024), 0, stream, tile_sums, ntiles);
  hipLaunchKernelGGL(k_scan_u32_apply, dim3(ntiles), dim3(1024), 0, stream, reinterpret_cast<uint32_t*>(a.dense_head), tile_sums);
  hipLaunchKernelGGL(k_dense_csr_fill, dim3(grid_for(a.build.nrows)), dim3(256), 0, stream, d_args, cursor);
  CUDF_HIP_TRY(hipGetLastError());
}
void launch_count(join_args const& a, join_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<join_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_count", stream};
  if (a.dense_head != nullptr) hipLaunchKernelGGL(k_dense_count, dim3(a.nblocks), dim3(256), 0, stream, d_args);
  else if (a.single64) hipLaunchKernelGGL((k_probe_count<MODE_KEY64>), dim3(a.nblocks), dim3(256), 0, stream, d_args);
  else hipLaunchKernelGGL((k_probe_count<MODE_GENERIC>), dim3(a.nblocks), dim3(256), 0, stream, d_args);
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
  if (a.dense_head != nullptr) hipLaunchKernelGGL(k_dense_retrieve, dim3(a.nblocks), dim3(256), 0, stream, d_args);
  else if (a.single64) hipLaunchKernelGGL((k_probe_retrieve<MODE_KEY64>), dim3(a.nblocks), dim3(256), 0, stream, d_args);
  else hipLaunchKernelGGL((k_probe_retrieve<MODE_GENERIC>), dim3(a.nblocks), dim3(256), 0, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}
void launch_dense_big_emit(join_args const& a, join_args* d_args, hipStream_t stream)
{
  (void)a;
  cudf::detail::prof::scope prof_{"join_retrieve", stream};
  hipLaunchKernelGGL(k_dense_big_emit, dim3(2048), dim3(256), 0, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}
void launch_mark_matched(size_type const* right_indices, std::size_t n, uint8_t* build_matched, hipStream_t stream)
{
  if (n == 0) return;
  hipLaunchKernelGGL(k_mark_matched, dim3(grid_for(static_cast<int64_t>(n))), dim3(256), 0, stream, right_indices, n, build_matched);
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
