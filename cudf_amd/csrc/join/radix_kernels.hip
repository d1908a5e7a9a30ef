// SPDX-License-Identifier: Apache-2.0
// gfx950 kernels of the LDS radix join (engine.hpp radix_scatter_args / radix_join_args).
//
// k_radix_scatter: the ring scatter of the groupby's dense path (groupby/dense_ring_kernels.hip) for {key, row id} rows: every
// partition owns a ring of key slots and a ring of row-id slots in LDS; a row reserves its position with ONE returning LDS atomic;
// after a barrier the owner lanes flush every complete 128-byte granule (16 keys / 32 row ids); a second barrier ends the
// 4096-row tile. 12 bytes per row leave the workgroup, in two streams.
// k_radix_join: one workgroup per partition builds an open-addressing multiset of the partition's build rows in LDS (keys as the
// slots' state, probed in aligned pairs = one ds_read_b128) and streams the partition's probe rows past it - count pass and
// retrieve pass (the reference's size_impl.cuh:26-61 / retrieve_impl.cuh:29-133 structure), pairs allocated from a
// workgroup-local cursor by 64-lane ballots.
#include "engine.hpp"
#include "../common/profiler.hpp"
#include "../common/ring_scatter.hpp"

#include <cudf/join/join.hpp>
#include <cudf/utilities/error.hpp>

#include <mutex>

namespace cudf::detail::join {
namespace {

template <typename T>
__global__ void k_store_radix_args(T v, T* dst)
{
  *dst = v;
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint64_t radix_hash(uint64_t key) { return mix64(0x9e3779b97f4a7c15ull ^ key); }
// two-word keys: ONE 64-bit word that stands for the pair in the partition digits and as the LDS table's slot state (every candidate it
// yields is verified against both words)
__device__ __forceinline__ uint64_t fold_key(uint64_t k0, uint64_t k1)
{
  uint64_t const h1 = mix64(0xc2b2ae3d27d4eb4full ^ k1);
  return k0 ^ ((h1 << 31) | (h1 >> 33)) ^ 0x165667b19e3779f9ull;
}

constexpr int RADIX_MAX_REGION_LIST = 256;

// DENSE (level 1 only; the dense direct-address join's partition pass, dense_part_kernels.hip): a row is the ONE 8-byte record
// {key - dense_lo (32 bits) | row id << 32}, its partition the top bits of that offset (a contiguous slice of the direct-address
// table); rows whose key lies outside [dense_lo, dense_lo + dense_range) are dropped like NULL rows. No row-id stream.
// SHAPE (level 1): what the key columns are, so that the row loop holds NO run-time branch and NO arithmetic on a loaded word before
// the tile is decoded (either one makes the compiler wait for all outstanding loads: the general loader's first level ran C3's shape at
// 4.3 instead of 2.4 ms - 243 against 28 `s_waitcnt vmcnt(0)` in the kernel):
//   0  anything, tested at run time (4-byte keys, two packed 4-byte columns, float32, two-word keys)
//   1  one 8-byte integer column, read as is                                  (C3)
//   2  one float64 column: its normalised bits                                (normalize_key_bits, arithmetic only)
//   3 / 4 / 5  two integer columns of 8+8 / 8+4 / 4+8 bytes packed into one exact word (engine.hpp radix_scatter_args::pack)
//   6 / 7 / 8  two integer columns of 8+8 / 8+4 / 4+8 bytes as a two-word key (KW == 2)
constexpr int SHAPE_ANY = 0, SHAPE_INT64 = 1, SHAPE_F64 = 2, SHAPE_PACK88 = 3, SHAPE_PACK84 = 4, SHAPE_PACK48 = 5, SHAPE_TWO88 = 6, SHAPE_TWO84 = 7,
              SHAPE_TWO48 = 8;
template <int LEVEL, int RPT, int D, bool DENSE, int B = 1024, int KW = 1, int SHAPE = SHAPE_ANY>
__global__ void __launch_bounds__(B) k_radix_scatter(radix_scatter_args const* __restrict__ ap)
{
  static_assert(SHAPE == SHAPE_ANY || (LEVEL == 1 && (KW == 2) == (SHAPE >= SHAPE_TWO88)), "SHAPE: the first level; 6-8 are the two-word keys");
  static_assert(SHAPE < SHAPE_PACK88 || !DENSE, "two key columns: the hash-partitioned (radix) join only");
  constexpr bool PLAIN = SHAPE == SHAPE_INT64 || SHAPE == SHAPE_F64;
  constexpr bool PACK  = SHAPE >= SHAPE_PACK88 && SHAPE <= SHAPE_PACK48;
  constexpr bool TWO   = SHAPE >= SHAPE_TWO88;
  constexpr int W0 = (SHAPE == SHAPE_PACK48 || SHAPE == SHAPE_TWO48) ? 4 : 8, W1 = (SHAPE == SHAPE_PACK84 || SHAPE == SHAPE_TWO84) ? 4 : 8;  // two columns: widths
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int s_pending, s_abort, s_rounds;
  __shared__ int32_t s_pre[RADIX_MAX_REGION_LIST + 1];
  radix_scatter_args const& a = *ap;
  constexpr uint32_t G = 16, GT = 32;  // keys / row ids per 128-byte granule
  int const P = a.P, capl = a.capl;
  uint32_t const CAP = 1u << capl, cmask = CAP - 1u, tcmask = (CAP << 1) - 1u;
  uint32_t const nslots = static_cast<uint32_t>(P) << capl;
  uint64_t* rkey  = reinterpret_cast<uint64_t*>(lds_raw);                        // [P << capl]
  [[maybe_unused]] uint64_t* rkey1 = rkey + nslots;                               // [P << capl] (KW == 2)
  uint32_t* rrow  = reinterpret_cast<uint32_t*>(rkey + static_cast<uint32_t>(KW) * nslots);  // [P << (capl + 1)] (not DENSE)
  uint32_t* tail  = rrow + (DENSE ? 0u : 2u * nslots);                            // [P] next virtual position
  uint32_t* limit = tail + P;                                                     // [P] head + CAP as of the last flush
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = B >> 6;
  int const PW = P / nwaves;  // partitions owned by a wave (1 ... 16)
  uint32_t head = 0, headt = 0;
  for (int d = threadIdx.x; d < P; d += B) {
    tail[d]  = 0;
    limit[d] = CAP;
  }
  if (threadIdx.x == 0) {
    s_pending = 0;
    s_abort   = 0;
    s_rounds  = 0;
  }
  constexpr int64_t T = static_cast<int64_t>(B) * RPT;
  int item = blockIdx.x, seg = 0;
  int64_t begin, end, step;
  int nreg = 0;
  int64_t rfirst = 0, rstride = 0;
  if constexpr (LEVEL == 2) {
    seg     = blockIdx.x / a.slices;
    item    = blockIdx.x % a.slices;
    nreg    = (a.in_slices - item + a.slices - 1) / a.slices;
    rfirst  = (static_cast<int64_t>(seg) * a.in_slices + item) * a.in_region_cap;
    rstride = static_cast<int64_t>(a.slices) * a.in_region_cap;
    if (threadIdx.x == 0) {
      int32_t run = 0;
      for (int j = 0; j < nreg; ++j) {
        s_pre[j] = run;
        int64_t const c = a.in_region_count[static_cast<int64_t>(seg) * a.in_slices + item + static_cast<int64_t>(j) * a.slices];
        run += static_cast<int32_t>(min(max(c, int64_t{0}), a.in_region_cap));
      }
      s_pre[nreg] = run;
    }
    __syncthreads();
    begin = 0;
    end   = s_pre[nreg];
    step  = T;
  } else {
    begin = static_cast<int64_t>(item) * T;  // workgroup w takes the row tiles w, w + slices, ...
    end   = a.nrows;
    step  = static_cast<int64_t>(a.slices) * T;
    __syncthreads();
  }
  int64_t const region0     = static_cast<int64_t>(seg) * P * a.slices;
  uint32_t const region_cap = static_cast<uint32_t>(a.region_cap);
  int const shift           = a.shift;
  uint32_t const pmask      = static_cast<uint32_t>(P - 1);
  int reg_hint[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) reg_hint[k] = 0;
  auto record_of = [&](int64_t v, int& reg) -> int64_t {  // (a thread's rows ascend from tile to tile: the region only moves forward)
    while (reg + 1 < nreg && s_pre[reg + 1] <= v) ++reg;
    return rfirst + static_cast<int64_t>(reg) * rstride + (v - s_pre[reg]);
  };
  // The loads of a tile are issued D tiles ahead and only LOADED there: every widening, normalisation or packing of a loaded word
  // happens where the tile is decoded. (Arithmetic on a loaded value inside issue() makes the compiler wait for the load right there:
  // the level-1 kernel then ran at 4.3 instead of 2.4 ms - 243 against 28 `s_waitcnt vmcnt(0)`.)
  constexpr bool SECOND = KW == 2 || PACK || (LEVEL == 1 && SHAPE == SHAPE_ANY && !DENSE);  // a second loaded word per row (second key column)
  struct tile_regs {
    uint64_t k[RPT];                  // level 1: the first key column's raw bits (zero-extended); level 2: the key
    uint64_t k1[SECOND ? RPT : 1];    // level 1: the second key column's raw bits; level 2, KW == 2: the second key word
    uint32_t r[RPT];                  // level 1: the validity word of the row; level 2: the row id
    uint32_t r2[(LEVEL == 1 && SECOND) ? RPT : 1];  // level 1: the second column's validity word
  };
  auto issue = [&](int64_t tile, tile_regs& t) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      int64_t const row = tile + static_cast<int64_t>(k) * B + threadIdx.x;
      if (row < end) {
        if constexpr (LEVEL == 1) {
          if constexpr (PLAIN) {
            t.k[k] = gload(a.keys + row);
          } else if constexpr (PACK || TWO) {
            if constexpr (W0 == 4) t.k[k] = gload(reinterpret_cast<uint32_t const*>(a.keys) + row);
            else t.k[k] = gload(a.keys + row);
            if constexpr (W1 == 4) t.k1[k] = gload(static_cast<uint32_t const*>(a.key2) + row);
            else t.k1[k] = gload(static_cast<uint64_t const*>(a.key2) + row);
            t.r2[k] = a.mask2 != nullptr ? gload(a.mask2 + ((a.mask2_offset + row) >> 5)) : 0xffffffffu;
          } else {
            if (a.key_width == 4) t.k[k] = gload(reinterpret_cast<uint32_t const*>(a.keys) + row);
            else t.k[k] = gload(a.keys + row);
            if constexpr (SECOND) {
              void const* const second = a.keys2 != nullptr ? static_cast<void const*>(a.keys2) : a.key2;
              if (second != nullptr) {
                if (a.keys2 != nullptr || a.key2_width == 4) t.k1[k] = gload(static_cast<uint32_t const*>(second) + row);
                else t.k1[k] = gload(static_cast<uint64_t const*>(second) + row);
              }
              t.r2[k] = a.mask2 != nullptr ? gload(a.mask2 + ((a.mask2_offset + row) >> 5)) : 0xffffffffu;
            }
          }
          t.r[k] = a.mask != nullptr ? gload(a.mask + ((a.mask_offset + row) >> 5)) : 0xffffffffu;
        } else {
          int64_t const ri = record_of(row, reg_hint[k]);
          t.k[k]           = gload(a.in_key + ri);
          if constexpr (KW == 2) t.k1[k] = gload(a.in_key1 + ri);
          t.r[k]           = gload(a.in_row + ri);
        }
      }
    }
  };
  auto flush = [&](bool final) {  // (common/ring_scatter.hpp) keys: granules of 16; row ids: granules of 32, in rings twice as long
    int const dmine = wave * PW + lane;
    auto const f    = cudf::detail::ring::plan_flush<G, GT>(tail, lane < PW, dmine, head, headt, CAP, region_cap, final, s_abort);
    auto const rbase_of = [&](int d) { return (region0 + static_cast<int64_t>(d) * a.slices + item) * a.region_cap; };
    auto const key_slot = [&](uint32_t d, uint32_t pos) { return (d << capl) + (pos & cmask); };
    cudf::detail::ring::flush_stream(rkey, a.out_key, wave, PW, lane, f.nrec, head, f.ab, key_slot, rbase_of);
    if constexpr (KW == 2) cudf::detail::ring::flush_stream(rkey1, a.out_key1, wave, PW, lane, f.nrec, head, f.ab, key_slot, rbase_of);
    if constexpr (!DENSE)
      cudf::detail::ring::flush_stream(rrow, a.out_row, wave, PW, lane, f.nrect, headt, f.ab,
                                       [&](uint32_t d, uint32_t pos) { return (d << (capl + 1)) + (pos & tcmask); }, rbase_of);
    cudf::detail::ring::commit_flush(limit, lane < PW, dmine, head, headt, f, CAP);
  };
  auto put = [&](uint32_t d, uint32_t pos, uint64_t key, [[maybe_unused]] uint64_t key1, uint32_t rowid) {
    rkey[(d << capl) + (pos & cmask)] = key;
    if constexpr (KW == 2) rkey1[(d << capl) + (pos & cmask)] = key1;
    if constexpr (!DENSE) rrow[(d << (capl + 1)) + (pos & tcmask)] = rowid;
  };

  tile_regs pre[D];
#pragma unroll
  for (int j = 0; j < D; ++j) issue(begin + j * step, pre[j]);
  for (int64_t tile = begin; tile < end; tile += D * step) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      int64_t const t0 = tile + j * step;
      bool keep[RPT];
      uint32_t d[RPT], rowid[RPT];
      uint64_t key[RPT];
      [[maybe_unused]] uint64_t key1[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        int64_t const row = t0 + static_cast<int64_t>(k) * B + threadIdx.x;
        keep[k]           = row < end;
        key[k]            = pre[j].k[k];
        key1[k]           = 0;
        if constexpr (LEVEL == 1) {
          rowid[k] = static_cast<uint32_t>(row);
          keep[k]  = keep[k] && ((pre[j].r[k] >> ((a.mask_offset + row) & 31)) & 1u);  // a NULL key joins nothing (UNEQUAL)
          auto const widen32 = [](uint64_t raw, int is_signed) {  // (a select, not a branch)
            return is_signed ? static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(static_cast<uint32_t>(raw)))) : (raw & 0xffffffffull);
          };
          if constexpr (SHAPE == SHAPE_F64) {
            key[k] = normalize_key_bits(key[k], cudf::detail::CLS_F64);  // equal values, equal bits
          } else if constexpr (PACK) {  // two integer columns inside the build side's value box: one exact word (engine.hpp)
            uint64_t const c0 = W0 == 4 ? widen32(key[k], a.key_signed) : key[k];
            uint64_t const c1 = W1 == 4 ? widen32(pre[j].k1[k], a.key2_signed) : pre[j].k1[k];
            uint64_t const d0 = c0 - a.pack_lo0, d1 = c1 - a.pack_lo1;
            bool const inside = d0 <= a.pack_range0 && d1 <= a.pack_range1;
            key[k]  = inside ? ((d0 << a.pack_bits1) | d1) : (0x8000000000000000ull | (cudf::detail::mix64(static_cast<uint64_t>(row)) >> 1));
            keep[k] = keep[k] && ((pre[j].r2[k] >> ((a.mask2_offset + row) & 31)) & 1u) && (inside || a.pack_keep_outside != 0);
          } else if constexpr (TWO) {  // the first column widened like a single key column, the second's raw bits zero-extended
            if constexpr (W0 == 4) key[k] = widen32(key[k], a.key_signed);
            key1[k] = pre[j].k1[k];
            keep[k] = keep[k] && ((pre[j].r2[k] >> ((a.mask2_offset + row) & 31)) & 1u);
          } else if constexpr (SHAPE == SHAPE_ANY) {
            if (a.keys2 != nullptr) {  // TWO 4-byte integer key columns, packed into the 8-byte key (a bijection: rows are equal iff both are)
              key[k] = (key[k] & 0xffffffffull) | (pre[j].k1[k] << 32);
            } else if (a.key_width == 4) {  // a 4-byte integer key column: widened here (sign-extended when the type is signed)
              key[k] = widen32(key[k], a.key_signed);
            }
            if (a.key_class != 0) key[k] = normalize_key_bits(key[k], a.key_class);  // (a float key: equal values, equal bits)
            if constexpr (SECOND) {
              // (a NULL in the second column drops the row as well)
              keep[k] = keep[k] && ((pre[j].r2[k] >> ((a.mask2_offset + row) & 31)) & 1u);
              if constexpr (KW == 2) {  // the second column: raw bits zero-extended (equal values have equal bits once floats are normalised)
                key1[k] = pre[j].k1[k];
                if (a.key2_class != 0) key1[k] = normalize_key_bits(key1[k], a.key2_class);
              } else if (a.pack != 0) {
                uint64_t const c1 = a.key2_width == 4 ? widen32(pre[j].k1[k], a.key2_signed) : pre[j].k1[k];
                uint64_t const d0 = key[k] - a.pack_lo0, d1 = c1 - a.pack_lo1;
                bool const inside = d0 <= a.pack_range0 && d1 <= a.pack_range1;
                key[k]  = inside ? ((d0 << a.pack_bits1) | d1) : (0x8000000000000000ull | (cudf::detail::mix64(static_cast<uint64_t>(row)) >> 1));
                keep[k] = keep[k] && (inside || a.pack_keep_outside != 0);
              }
            }
          }
        } else {
          if constexpr (KW == 2) key1[k] = pre[j].k1[k];
          rowid[k] = pre[j].r[k];
        }
        if constexpr (DENSE) {
          uint64_t const off = key[k] - a.dense_lo;
          keep[k]            = keep[k] && off < a.dense_range;
          d[k]               = keep[k] ? static_cast<uint32_t>(off >> shift) : 0u;
          key[k]             = off | (static_cast<uint64_t>(rowid[k]) << 32);
        } else {
          d[k] = keep[k] ? static_cast<uint32_t>(radix_hash(KW == 2 ? fold_key(key[k], key1[k]) : key[k]) >> shift) & pmask : 0u;
        }
      }
      issue(t0 + D * step, pre[j]);
      if (t0 >= end) break;  // (uniform)
      // (DENSE: rows that arrive sorted or clustered by key fill one ring tile after tile; the caller's direct path is the better one
      // for them - a tile without a wait pays one round back, and the workgroup gives up once the waiting rounds outnumber the tiles)
      cudf::detail::ring::place_tile<RPT>(
        tail, limit, keep, d, s_pending, s_abort, [&](int k, uint32_t pos) { put(d[k], pos, key[k], key1[k], rowid[k]); }, [&] { flush(false); },
        [&] {
          if (DENSE && threadIdx.x == 0 && s_rounds > 0) --s_rounds;
        },
        [&] {
          if (DENSE && ++s_rounds > a.pending_budget) s_abort = 1;
        });
      if (s_abort) {
        if (threadIdx.x == 0) atomicOr(a.overflow, 1);
        return;
      }
    }
  }
  flush(true);
  lds_barrier();
  if (s_abort) {
    if (threadIdx.x == 0) atomicOr(a.overflow, 1);
    return;
  }
  if (lane < PW) a.region_count[region0 + static_cast<int64_t>(wave * PW + lane) * a.slices + item] = static_cast<int32_t>(head);
}

__global__ void __launch_bounds__(256) k_radix_partition_max(int32_t const* region_count, int32_t nparts, int32_t slices, int32_t* out_max)
{
  int const q = blockIdx.x * blockDim.x + threadIdx.x;
  int32_t tot = 0;
  if (q < nparts)
    for (int s = 0; s < slices; ++s) tot += max(region_count[static_cast<int64_t>(q) * slices + s], 0);
  for (int o = 32; o > 0; o >>= 1) tot = max(tot, __shfl_down(tot, o));
  if ((threadIdx.x & 63) == 0 && tot > 0) atomicMax(out_max, tot);
}

// ------------------------------------------------------------------ the join of one partition in LDS
constexpr uint64_t RJ_EMPTY = 0x7f4a7c159e3779b9ull;  // (the key that equals it takes the side list below)
constexpr int RJ_SPECIAL   = 256;

// bucket of a key in a partition's table: the partition is the top bits of mix64, the table takes a cheap multiplicative hash of
// the two key words (two 32-bit multiplies instead of mix64's eight: the join kernel is bound by instruction issue, not by memory)
__device__ __forceinline__ uint32_t table_hash(uint64_t key)
{
  return static_cast<uint32_t>(key) * 0x9e3779b1u + static_cast<uint32_t>(key >> 32) * 0x85ebca77u;
}
constexpr int RJ_RPT = 4;  // probe records a thread loads before it walks the table for them (HBM latency: ~48 KB in flight per CU)

// LEFT: a probe record without a partner yields the pair {probe row, JoinNoMatch} (left join; the probe rows the scatter dropped
// - NULL keys - get theirs from k_radix_null_rows)
// KW == 2: the slot state is fold_key(k0, k1), rows[] holds the build record's offset inside the partition, and a candidate counts only
// if both words of the build record (read back from the partition's regions: they were loaded a moment ago, L2) equal the probe's.
template <bool RETRIEVE, bool LEFT, int KW = 1>
__global__ void __launch_bounds__(1024) k_radix_join(radix_join_args const* __restrict__ ap)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ unsigned long long s_cursor;
  __shared__ uint32_t s_spec_n, s_nbuild;
  __shared__ int32_t s_spec_rows[RJ_SPECIAL];
  radix_join_args const& a = *ap;
  constexpr int B = 1024;
  int const q = blockIdx.x, cap = a.cap;
  // retrieve pass: only the partitions whose pairs did not fit their stage are joined again (k_radix_emit_staged copies the rest)
  if constexpr (RETRIEVE) {
    if (a.pair_counts[q + 1] - a.pair_counts[q] <= static_cast<unsigned long long>(a.stage_cap)) return;
  }
  uint64_t* keys = reinterpret_cast<uint64_t*>(lds_raw);          // [cap + 2]: the table and a spare bucket that stays empty
  uint32_t* rows = reinterpret_cast<uint32_t*>(keys + cap + 2);   // [cap]
  uint32_t const bmask = static_cast<uint32_t>(cap / 2 - 1);      // buckets of two slots: one ds_read_b128
  int const bshift     = __builtin_clz(bmask);                    // bucket = the top bits of table_hash
  for (int s = threadIdx.x; s < cap / 2 + 1; s += B) reinterpret_cast<u64x2*>(keys)[s] = u64x2{RJ_EMPTY, RJ_EMPTY};
  if (threadIdx.x == 0) {
    s_cursor = RETRIEVE ? a.pair_counts[q] : 0ull;
    s_spec_n = 0;
    s_nbuild = 0;
  }
  bool const upstream_ok = (*a.overflow & 3) == 0;
  // ---- build: the partition's build rows into the table (a multiset: equal keys take separate slots along the probe sequence)
  auto insert = [&](uint64_t key, uint32_t row) {
    if (key == RJ_EMPTY) {
      uint32_t const at = atomicAdd(&s_spec_n, 1u);
      if (at < RJ_SPECIAL) s_spec_rows[at] = static_cast<int32_t>(row);
      return;
    }
    uint32_t bkt = table_hash(key) >> bshift;
    for (int guard = 0; guard < 2 * cap; ++guard) {
      u64x2 const pr = *reinterpret_cast<u64x2 const volatile*>(keys + 2u * bkt);
      int const e    = pr.x == RJ_EMPTY ? 0 : (pr.y == RJ_EMPTY ? 1 : -1);
      if (e < 0) {
        bkt = (bkt + 1) & bmask;
        continue;
      }
      if (atomicCAS(reinterpret_cast<unsigned long long*>(keys + 2u * bkt + e), static_cast<unsigned long long>(RJ_EMPTY),
                    static_cast<unsigned long long>(key)) == RJ_EMPTY) {
        rows[2u * bkt + e] = row;
        return;
      }
      // (lost the slot: read the bucket again)
    }
  };
  // (the first 1024 rows of up to four slices are loaded before the table is ready: one round trip to HBM instead of four)
  uint64_t bk[4];
  uint32_t br[4];
  bool bv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bv[s] = false;
    if (upstream_ok && s < a.b_slices) {
      int64_t const reg = static_cast<int64_t>(q) * a.b_slices + s;
      int32_t const cnt = min(max(a.b_count[reg], 0), static_cast<int32_t>(a.b_cap));
      bv[s]             = static_cast<int32_t>(threadIdx.x) < cnt;
      if (bv[s]) {
        bk[s] = gload(a.b_key + reg * a.b_cap + threadIdx.x);
        if constexpr (KW == 2) {
          bk[s] = fold_key(bk[s], gload(a.b_key1 + reg * a.b_cap + threadIdx.x));
          br[s] = static_cast<uint32_t>(s * a.b_cap + threadIdx.x);  // (the record's offset inside the partition)
        } else {
          br[s] = gload(a.b_row + reg * a.b_cap + threadIdx.x);
        }
      }
    }
  }
  __syncthreads();  // (the table is empty everywhere)
  uint32_t nb = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (bv[s]) {
      insert(bk[s], br[s]);
      ++nb;
    }
  for (int s = 0; upstream_ok && s < a.b_slices; ++s) {
    int64_t const reg  = static_cast<int64_t>(q) * a.b_slices + s;
    int32_t const cnt  = min(max(a.b_count[reg], 0), static_cast<int32_t>(a.b_cap));
    int64_t const base = reg * a.b_cap;
    for (int32_t i = threadIdx.x + (s < 4 ? B : 0); i < cnt; i += B) {
      if constexpr (KW == 2) insert(fold_key(gload(a.b_key + base + i), gload(a.b_key1 + base + i)), static_cast<uint32_t>(s * a.b_cap + i));
      else insert(gload(a.b_key + base + i), gload(a.b_row + base + i));
      ++nb;
    }
  }
  for (int o = 32; o > 0; o >>= 1) nb += __shfl_down(nb, o);
  if ((threadIdx.x & 63) == 0 && nb) atomicAdd(&s_nbuild, nb);
  __syncthreads();
  // a partition whose build rows do not fit (a heavily duplicated key, a wrong size estimate): the caller falls back
  if (static_cast<int>(s_nbuild) > a.fill_limit || s_spec_n > RJ_SPECIAL) {
    if (threadIdx.x == 0) {
      atomicOr(a.overflow, 2);
      if (!RETRIEVE) a.pair_counts[q] = 0;
    }
    return;
  }
  uint32_t const nspec = s_spec_n;
  // ---- probe: the partition's probe rows past the table; a wave reserves the positions of its pairs with one LDS atomic
  int const lane             = threadIdx.x & 63;
  uint64_t const below       = (1ull << lane) - 1ull;
  uint64_t* const stage      = a.stage + static_cast<int64_t>(q) * a.stage_cap;
  auto put = [&](unsigned long long o, size_type prow, size_type brow) {
    if constexpr (RETRIEVE) {
      if (o < a.out_capacity) {
        gstore(a.out_probe + o, prow);
        gstore(a.out_build + o, brow);
      }
    } else {
      if (o < static_cast<unsigned long long>(a.stage_cap))
        gstore(stage + o, static_cast<uint64_t>(static_cast<uint32_t>(prow)) | (static_cast<uint64_t>(static_cast<uint32_t>(brow)) << 32));
    }
  };
  auto emit2 = [&](size_type prow, bool w0, size_type b0, bool w1, size_type b1) {  // (every lane of the wave calls it)
    unsigned long long const m0 = __ballot(w0), m1 = __ballot(w1);
    if ((m0 | m1) == 0) return;
    int const n0            = __popcll(m0);
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&s_cursor, static_cast<unsigned long long>(n0 + __popcll(m1)));
    base = __shfl(base, 0);
    if (w0) put(base + __popcll(m0 & below), prow, b0);
    if (w1) put(base + n0 + __popcll(m1 & below), prow, b1);
  };
  // KW == 2: the build record a slot points at - does it carry these two words? / its row id
  int64_t const pbase = static_cast<int64_t>(q) * a.b_slices * a.b_cap;
  [[maybe_unused]] auto verified = [&](uint32_t off, uint64_t k0, uint64_t k1) { return gload(a.b_key + pbase + off) == k0 && gload(a.b_key1 + pbase + off) == k1; };
  [[maybe_unused]] auto build_row = [&](uint32_t off) { return gload(a.b_row + pbase + off); };
  // one record of every lane the wave-uniform way: pairs are emitted while the chains are walked (any number of matches per row)
  auto walk_and_emit = [&](bool live, uint64_t key, [[maybe_unused]] uint64_t k0, [[maybe_unused]] uint64_t k1, size_type prow) {
    bool const special = live && key == RJ_EMPTY;
    bool walking       = live && !special;
    bool found         = KW == 1 && special && nspec != 0;
    uint32_t bkt       = table_hash(key) >> bshift;
    while (__any(walking)) {
      u64x2 pr{RJ_EMPTY, RJ_EMPTY};
      if (walking) pr = *reinterpret_cast<u64x2 const*>(keys + 2u * bkt);
      bool w0 = walking && pr.x == key, w1 = walking && pr.x != RJ_EMPTY && pr.y == key;
      size_type b0 = w0 ? static_cast<size_type>(rows[2u * bkt]) : 0, b1 = w1 ? static_cast<size_type>(rows[2u * bkt + 1]) : 0;
      if constexpr (KW == 2) {
        w0 = w0 && verified(static_cast<uint32_t>(b0), k0, k1);
        w1 = w1 && verified(static_cast<uint32_t>(b1), k0, k1);
        b0 = w0 ? static_cast<size_type>(build_row(static_cast<uint32_t>(b0))) : 0;
        b1 = w1 ? static_cast<size_type>(build_row(static_cast<uint32_t>(b1))) : 0;
      }
      found = found || w0 || w1;
      emit2(prow, w0, b0, w1, b1);
      walking = walking && pr.x != RJ_EMPTY && pr.y != RJ_EMPTY;  // an entry takes the first empty slot of its sequence
      bkt     = (bkt + 1) & bmask;
    }
    if (nspec != 0 && __any(special))
      for (uint32_t t = 0; t < nspec; ++t) {
        if constexpr (KW == 2) {
          bool const ok = special && verified(static_cast<uint32_t>(s_spec_rows[t]), k0, k1);
          found         = found || ok;
          emit2(prow, ok, ok ? static_cast<size_type>(build_row(static_cast<uint32_t>(s_spec_rows[t]))) : 0, false, 0);
        } else {
          emit2(prow, special, s_spec_rows[t], false, 0);
        }
      }
    if constexpr (KW == 2) found = found && live;
    if constexpr (LEFT) emit2(prow, live && !found, JoinNoMatch, false, 0);
  };
  struct batch {
    uint64_t k[RJ_RPT];
    uint64_t k1[KW == 2 ? RJ_RPT : 1];
    uint32_t w[RJ_RPT];
  };
  for (int s = 0; upstream_ok && s < a.p_slices; ++s) {
    int64_t const reg  = static_cast<int64_t>(q) * a.p_slices + s;
    int32_t const cnt  = min(max(a.p_count[reg], 0), static_cast<int32_t>(a.p_cap));
    int64_t const base = reg * a.p_cap;
    auto issue = [&](int32_t i0, batch& t) {
#pragma unroll
      for (int j = 0; j < RJ_RPT; ++j) {
        int32_t const i = i0 + j * B + static_cast<int32_t>(threadIdx.x);
        t.k[j]          = 0;
        t.w[j]          = 0;
        if constexpr (KW == 2) t.k1[j] = 0;
        if (i < cnt) {
          t.k[j] = gload(a.p_key + base + i);
          if constexpr (KW == 2) t.k1[j] = gload(a.p_key1 + base + i);
          t.w[j] = gload(a.p_row + base + i);
        }
      }
    };
    batch nxt;
    issue(0, nxt);
    for (int32_t i0 = 0; i0 < cnt; i0 += B * RJ_RPT) {  // (every lane runs every round: the ballots below are wave-wide)
      batch const cur = nxt;
      issue(i0 + B * RJ_RPT, nxt);  // (the next records are on their way while these walk the table)
      // the four records of a lane walk their chains together - four independent LDS reads per step instead of one - and only
      // note the first match and the number of matches
      // Straight-line steps: a lane that is not walking reads the spare bucket behind the table (always empty) with a key no
      // slot can equal, so that a step needs no branch: a bucket's second slot is taken only after its first, hence "the bucket
      // is full" = "its second slot is not empty", and a key in the second slot implies the first is taken.
      bool live[RJ_RPT];
      uint64_t hk[RJ_RPT];  // the word the table knows the record by
#pragma unroll
      for (int j = 0; j < RJ_RPT; ++j) hk[j] = KW == 2 ? fold_key(cur.k[j], cur.k1[j]) : cur.k[j];
      uint64_t wk[RJ_RPT];
      uint32_t bkt[RJ_RPT], at[RJ_RPT], nmatch[RJ_RPT], mslot[RJ_RPT], brow[RJ_RPT];
      bool slow = false, any_walking = false;
      uint32_t const spare = static_cast<uint32_t>(cap / 2);
#pragma unroll
      for (int j = 0; j < RJ_RPT; ++j) {
        live[j]            = i0 + j * B + static_cast<int32_t>(threadIdx.x) < cnt;
        bool const walking = live[j] && hk[j] != RJ_EMPTY;
        slow               = slow || (live[j] && hk[j] == RJ_EMPTY && nspec != 0);
        wk[j]              = walking ? hk[j] : ~RJ_EMPTY;
        bkt[j]             = table_hash(hk[j]) >> bshift;
        at[j]              = walking ? bkt[j] : spare;
        nmatch[j]          = 0;
        mslot[j]           = 0;
        any_walking        = any_walking || walking;
      }
      while (any_walking) {
        u64x2 pr[RJ_RPT];
#pragma unroll
        for (int j = 0; j < RJ_RPT; ++j) pr[j] = *reinterpret_cast<u64x2 const*>(keys + 2u * at[j]);
        any_walking = false;
#pragma unroll
        for (int j = 0; j < RJ_RPT; ++j) {
          bool const m0 = pr[j].x == wk[j], m1 = pr[j].y == wk[j], full = pr[j].y != RJ_EMPTY;
          mslot[j]      = (m0 || m1) ? 2u * at[j] + (m0 ? 0u : 1u) : mslot[j];
          nmatch[j] += (m0 ? 1u : 0u) + (m1 ? 1u : 0u);
          bkt[j]      = (bkt[j] + 1) & bmask;
          at[j]       = full ? bkt[j] : spare;
          wk[j]       = full ? wk[j] : ~RJ_EMPTY;
          any_walking = any_walking || full;
        }
      }
#pragma unroll
      for (int j = 0; j < RJ_RPT; ++j) brow[j] = rows[mslot[j]];  // (meaningful where nmatch == 1)
#pragma unroll
      for (int j = 0; j < RJ_RPT; ++j) slow = slow || nmatch[j] > 1;
      if constexpr (KW == 2) {  // the single candidates: both words against the build record, then its row id (loads of all four go out together)
        uint64_t c0[RJ_RPT], c1[RJ_RPT];
        uint32_t cr[RJ_RPT];
#pragma unroll
        for (int j = 0; j < RJ_RPT; ++j) {
          uint32_t const off = nmatch[j] == 1 ? brow[j] : 0u;
          c0[j] = gload(a.b_key + pbase + off);
          c1[j] = gload(a.b_key1 + pbase + off);
          cr[j] = gload(a.b_row + pbase + off);
        }
#pragma unroll
        for (int j = 0; j < RJ_RPT; ++j) {
          if (nmatch[j] == 1 && !(c0[j] == cur.k[j] && c1[j] == cur.k1[j])) nmatch[j] = 0;  // (two keys with one fold: not a match)
          brow[j] = cr[j];
        }
      }
      if (!__any(slow)) {
        // at most one pair per record: one reservation for the wave's pairs of all four records
        unsigned long long m[RJ_RPT];
        int tot = 0;
#pragma unroll
        for (int j = 0; j < RJ_RPT; ++j) {
          m[j] = __ballot(nmatch[j] == 1 || (LEFT && live[j] && nmatch[j] == 0));
          tot += __popcll(m[j]);
        }
        if (tot != 0) {
          unsigned long long pos = 0;
          if (lane == 0) pos = atomicAdd(&s_cursor, static_cast<unsigned long long>(tot));
          pos = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(pos >> 32))) << 32) |
                __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(pos));
#pragma unroll
          for (int j = 0; j < RJ_RPT; ++j) {
            if (nmatch[j] == 1) put(pos + __popcll(m[j] & below), static_cast<size_type>(cur.w[j] + a.probe_row_base), static_cast<size_type>(brow[j]));
            else if (LEFT && live[j] && nmatch[j] == 0) put(pos + __popcll(m[j] & below), static_cast<size_type>(cur.w[j] + a.probe_row_base), JoinNoMatch);
            pos += __popcll(m[j]);
          }
        }
      } else {
        // a row with several matches, or the marker key: these records again, wave-uniform
#pragma unroll
        for (int j = 0; j < RJ_RPT; ++j) {
          if (i0 + j * B >= cnt) break;  // (uniform)
          walk_and_emit(live[j], hk[j], cur.k[j], KW == 2 ? cur.k1[j] : 0, static_cast<size_type>(cur.w[j] + a.probe_row_base));
        }
      }
    }
  }
  if constexpr (!RETRIEVE) {
    __syncthreads();
    if (threadIdx.x == 0) {
      a.pair_counts[q] = s_cursor;
      if (s_cursor > static_cast<unsigned long long>(a.stage_cap)) atomicOr(a.overflow, 4);
    }
  }
}

// left join: the probe rows whose key is NULL (the scatter dropped them) -> {row, JoinNoMatch} pairs appended at *cursor
__global__ void __launch_bounds__(256) k_radix_null_rows(bitmask_type const* __restrict__ mask, int64_t mask_offset, int64_t nrows, int64_t chunk,
                                                         int64_t row_base, size_type* out_probe, size_type* out_build, unsigned long long out_capacity,
                                                         unsigned long long* cursor)
{
  __shared__ unsigned long long s_base;
  __shared__ uint32_t s_n;
  int64_t const begin = static_cast<int64_t>(blockIdx.x) * chunk, end = min(nrows, begin + chunk);
  int const lane = threadIdx.x & 63;
  // pass 1: how many; pass 2: where (one global atomic per workgroup)
  uint32_t mine = 0;
  for (int64_t r = begin + threadIdx.x; r < end; r += blockDim.x) mine += !((gload(mask + ((mask_offset + r) >> 5)) >> ((mask_offset + r) & 31)) & 1u);
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
  if (lane == 0 && mine) atomicAdd(&s_n, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    s_base = s_n ? atomicAdd(cursor, static_cast<unsigned long long>(s_n)) : 0ull;
    s_n    = 0;
  }
  __syncthreads();
  for (int64_t r0 = begin; r0 < end; r0 += blockDim.x) {
    int64_t const r  = r0 + threadIdx.x;
    bool const null_ = r < end && !((gload(mask + ((mask_offset + r) >> 5)) >> ((mask_offset + r) & 31)) & 1u);
    unsigned long long const m = __ballot(null_);
    uint32_t at = 0;
    if (lane == 0 && m) at = atomicAdd(&s_n, static_cast<uint32_t>(__popcll(m)));
    at = __builtin_amdgcn_readfirstlane(at);
    if (null_) {
      unsigned long long const o = s_base + at + __popcll(m & ((1ull << lane) - 1ull));
      if (o < out_capacity) {
        gstore(out_probe + o, static_cast<size_type>(r + row_base));
        gstore(out_build + o, JoinNoMatch);
      }
    }
  }
}

// the staged pairs of every partition that fit its stage -> the output columns
__global__ void __launch_bounds__(256) k_radix_emit_staged(radix_join_args const* __restrict__ ap)
{
  radix_join_args const& a = *ap;
  int const q                  = blockIdx.x;
  unsigned long long const off = a.pair_counts[q], n = a.pair_counts[q + 1] - off;
  if (n > static_cast<unsigned long long>(a.stage_cap)) return;
  uint64_t const* src = a.stage + static_cast<int64_t>(q) * a.stage_cap;
  for (unsigned long long i0 = 0; i0 < n; i0 += 1024) {
    uint64_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned long long const i = i0 + j * 256 + threadIdx.x;
      v[j]                       = i < n ? gload(src + i) : 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned long long const i = i0 + j * 256 + threadIdx.x, o = off + i;
      if (i < n && o < a.out_capacity) {
        gstore(a.out_probe + o, static_cast<size_type>(static_cast<uint32_t>(v[j])));
        gstore(a.out_build + o, static_cast<size_type>(static_cast<uint32_t>(v[j] >> 32)));
      }
    }
  }
}

}  // namespace

void launch_radix_scatter(radix_scatter_args const& a, radix_scatter_args* d_args, hipStream_t stream)
{
  bool const dense = a.dense != 0;
  CUDF_EXPECTS(a.level != 1 || a.key_width == 4 || a.key_width == 8, "radix join scatter: key width");
  CUDF_EXPECTS(a.keys2 == nullptr || (a.level == 1 && a.key_width == 4 && a.dense == 0), "radix join scatter: two packed key columns are 4 bytes each");
  CUDF_EXPECTS(a.P >= 16 && a.P <= 256 && (a.P & (a.P - 1)) == 0 && a.capl >= 5 && a.shift < 64 && a.region_cap % 32 == 0 && a.slices >= 1,
               "radix join scatter: geometry");
  CUDF_EXPECTS(dense ? (a.level == 1 && (a.P << a.capl) <= 2 * RADIX_RING_SLOTS && a.dense_range <= (uint64_t{1} << 32) &&
                        ((a.dense_range - 1) >> a.shift) < static_cast<uint64_t>(a.P))
                     : ((a.P << a.capl) == RADIX_RING_SLOTS / std::max(a.kw, 1) && a.shift >= 32),
               "radix join scatter: geometry");
  CUDF_EXPECTS(a.kw <= 1 || (a.kw == 2 && !dense && a.out_key1 != nullptr && (a.level == 2 ? a.in_key1 != nullptr : (a.key2 != nullptr && (a.key2_width == 4 || a.key2_width == 8) && a.keys2 == nullptr))),
               "radix join scatter: two-word keys");
  CUDF_EXPECTS(a.level == 1 || (a.in_slices + a.slices - 1) / a.slices <= RADIX_MAX_REGION_LIST, "radix join scatter: region list too long");
  std::size_t const lds = static_cast<std::size_t>(a.P << a.capl) * (dense ? 8 : (a.kw == 2 ? 24 : 16)) + 2048;
  static std::once_flag attr_once;
  std::call_once(attr_once, [] {
    for (void const* fn : {reinterpret_cast<void const*>(&k_radix_scatter<1, 4, 2, false>), reinterpret_cast<void const*>(&k_radix_scatter<2, 4, 2, false>),
                           reinterpret_cast<void const*>(&k_radix_scatter<1, 4, 2, true>), reinterpret_cast<void const*>(&k_radix_scatter<1, 4, 2, true, 512>),
                           reinterpret_cast<void const*>(&k_radix_scatter<1, 8, 2, true>), reinterpret_cast<void const*>(&k_radix_scatter<1, 2, 2, false, 1024, 2>),
                           reinterpret_cast<void const*>(&k_radix_scatter<2, 2, 2, false, 1024, 2>),
                           reinterpret_cast<void const*>(&k_radix_scatter<1, 4, 2, false, 1024, 1, SHAPE_INT64>),
                           reinterpret_cast<void const*>(&k_radix_scatter<1, 4, 2, true, 1024, 1, SHAPE_INT64>),
                           reinterpret_cast<void const*>(&k_radix_scatter<1, 4, 2, false, 1024, 1, SHAPE_F64>),
                           reinterpret_cast<void const*>(&k_radix_scatter<1, 4, 2, false, 1024, 1, SHAPE_PACK88>),
                           reinterpret_cast<void const*>(&k_radix_scatter<1, 4, 2, false, 1024, 1, SHAPE_PACK84>),
                           reinterpret_cast<void const*>(&k_radix_scatter<1, 4, 2, false, 1024, 1, SHAPE_PACK48>),
                           reinterpret_cast<void const*>(&k_radix_scatter<1, 2, 2, false, 1024, 2, SHAPE_TWO88>),
                           reinterpret_cast<void const*>(&k_radix_scatter<1, 2, 2, false, 1024, 2, SHAPE_TWO84>),
                           reinterpret_cast<void const*>(&k_radix_scatter<1, 2, 2, false, 1024, 2, SHAPE_TWO48>)}) {
      hipFuncAttributes attr{};
      CUDF_HIP_TRY(hipFuncGetAttributes(&attr, fn));
      CUDF_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - static_cast<int>(attr.sharedSizeBytes)));
    }
  });
  hipLaunchKernelGGL(k_store_radix_args<radix_scatter_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  int shape = SHAPE_ANY;
  if (a.level == 1 && a.kw == 2 && !dense && a.key_class == 0 && a.key2_class == 0) {
    int const w = a.key_width * 10 + a.key2_width;
    shape       = w == 88 ? SHAPE_TWO88 : w == 84 ? SHAPE_TWO84 : w == 48 ? SHAPE_TWO48 : SHAPE_ANY;
  }
  if (a.level == 1 && a.kw <= 1 && a.keys2 == nullptr) {
    if (a.pack != 0 && !dense) {
      int const w = a.key_width * 10 + a.key2_width;
      shape       = w == 88 ? SHAPE_PACK88 : w == 84 ? SHAPE_PACK84 : w == 48 ? SHAPE_PACK48 : SHAPE_ANY;
    } else if (a.pack == 0 && a.key_width == 8) {
      shape = a.key_class == 0 ? SHAPE_INT64 : (a.key_class == static_cast<int32_t>(cudf::detail::CLS_F64) && !dense) ? SHAPE_F64 : SHAPE_ANY;
    }
  }
  cudf::detail::prof::scope prof_{a.level == 1 ? "join_partition" : "join_partition_level2", stream};
  if (dense && a.block == 512) hipLaunchKernelGGL((k_radix_scatter<1, 4, 2, true, 512>), dim3(a.slices), dim3(512), lds, stream, d_args);
  else if (dense && a.rpt == 8) hipLaunchKernelGGL((k_radix_scatter<1, 8, 2, true>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else if (dense && shape == SHAPE_INT64) hipLaunchKernelGGL((k_radix_scatter<1, 4, 2, true, 1024, 1, SHAPE_INT64>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else if (dense) hipLaunchKernelGGL((k_radix_scatter<1, 4, 2, true>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else if (shape == SHAPE_TWO88) hipLaunchKernelGGL((k_radix_scatter<1, 2, 2, false, 1024, 2, SHAPE_TWO88>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else if (shape == SHAPE_TWO84) hipLaunchKernelGGL((k_radix_scatter<1, 2, 2, false, 1024, 2, SHAPE_TWO84>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else if (shape == SHAPE_TWO48) hipLaunchKernelGGL((k_radix_scatter<1, 2, 2, false, 1024, 2, SHAPE_TWO48>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else if (a.kw == 2 && a.level == 1) hipLaunchKernelGGL((k_radix_scatter<1, 2, 2, false, 1024, 2>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else if (a.kw == 2) hipLaunchKernelGGL((k_radix_scatter<2, 2, 2, false, 1024, 2>), dim3(a.nseg * a.slices), dim3(1024), lds, stream, d_args);
  else if (shape == SHAPE_INT64) hipLaunchKernelGGL((k_radix_scatter<1, 4, 2, false, 1024, 1, SHAPE_INT64>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else if (shape == SHAPE_F64) hipLaunchKernelGGL((k_radix_scatter<1, 4, 2, false, 1024, 1, SHAPE_F64>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else if (shape == SHAPE_PACK88) hipLaunchKernelGGL((k_radix_scatter<1, 4, 2, false, 1024, 1, SHAPE_PACK88>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else if (shape == SHAPE_PACK84) hipLaunchKernelGGL((k_radix_scatter<1, 4, 2, false, 1024, 1, SHAPE_PACK84>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else if (shape == SHAPE_PACK48) hipLaunchKernelGGL((k_radix_scatter<1, 4, 2, false, 1024, 1, SHAPE_PACK48>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else if (a.level == 1) hipLaunchKernelGGL((k_radix_scatter<1, 4, 2, false>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  else hipLaunchKernelGGL((k_radix_scatter<2, 4, 2, false>), dim3(a.nseg * a.slices), dim3(1024), lds, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_radix_partition_max(int32_t const* region_count, int32_t nparts, int32_t slices, int32_t* out_max, hipStream_t stream)
{
  hipLaunchKernelGGL(k_radix_partition_max, dim3((nparts + 255) / 256), dim3(256), 0, stream, region_count, nparts, slices, out_max);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_radix_join(radix_join_args const& a, radix_join_args* d_args, bool retrieve, hipStream_t stream)
{
  CUDF_EXPECTS(a.cap >= 64 && (a.cap & (a.cap - 1)) == 0 && a.cap * 12 + 2048 <= 160 * 1024 && a.nparts >= 1 && a.fill_limit >= 1 && a.fill_limit < a.cap &&
                 a.stage != nullptr && a.stage_cap >= 1,
               "radix join: table geometry");
  std::size_t const lds = static_cast<std::size_t>(a.cap) * 12 + 16;
  static std::once_flag attr_once;
  std::call_once(attr_once, [] {
    for (void const* fn : {reinterpret_cast<void const*>(&k_radix_join<false, false>), reinterpret_cast<void const*>(&k_radix_join<true, false>),
                           reinterpret_cast<void const*>(&k_radix_join<false, true>), reinterpret_cast<void const*>(&k_radix_join<true, true>),
                           reinterpret_cast<void const*>(&k_radix_join<false, false, 2>), reinterpret_cast<void const*>(&k_radix_join<true, false, 2>),
                           reinterpret_cast<void const*>(&k_radix_join<false, true, 2>), reinterpret_cast<void const*>(&k_radix_join<true, true, 2>)}) {
      hipFuncAttributes attr{};
      CUDF_HIP_TRY(hipFuncGetAttributes(&attr, fn));
      CUDF_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - static_cast<int>(attr.sharedSizeBytes)));
    }
  });
  hipLaunchKernelGGL(k_store_radix_args<radix_join_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{retrieve ? "join_retrieve" : "join_count", stream};
  if (a.kw == 2) {
    CUDF_EXPECTS(a.b_key1 != nullptr && a.p_key1 != nullptr && static_cast<int64_t>(a.b_slices) * a.b_cap < (int64_t{1} << 31), "radix join: two-word keys");
    if (a.left != 0) {
      if (retrieve) hipLaunchKernelGGL((k_radix_join<true, true, 2>), dim3(a.nparts), dim3(1024), lds, stream, d_args);
      else hipLaunchKernelGGL((k_radix_join<false, true, 2>), dim3(a.nparts), dim3(1024), lds, stream, d_args);
    } else {
      if (retrieve) hipLaunchKernelGGL((k_radix_join<true, false, 2>), dim3(a.nparts), dim3(1024), lds, stream, d_args);
      else hipLaunchKernelGGL((k_radix_join<false, false, 2>), dim3(a.nparts), dim3(1024), lds, stream, d_args);
    }
  } else if (a.left != 0) {
    if (retrieve) hipLaunchKernelGGL((k_radix_join<true, true>), dim3(a.nparts), dim3(1024), lds, stream, d_args);
    else hipLaunchKernelGGL((k_radix_join<false, true>), dim3(a.nparts), dim3(1024), lds, stream, d_args);
  } else {
    if (retrieve) hipLaunchKernelGGL((k_radix_join<true, false>), dim3(a.nparts), dim3(1024), lds, stream, d_args);
    else hipLaunchKernelGGL((k_radix_join<false, false>), dim3(a.nparts), dim3(1024), lds, stream, d_args);
  }
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_radix_null_rows(bitmask_type const* mask, int64_t mask_offset, int64_t nrows, int64_t row_base, size_type* out_probe, size_type* out_build,
                            unsigned long long out_capacity, unsigned long long* cursor, hipStream_t stream)
{
  cudf::detail::prof::scope prof_{"join_retrieve", stream};
  int64_t const chunk = 65536;
  hipLaunchKernelGGL(k_radix_null_rows, dim3(static_cast<unsigned>((nrows + chunk - 1) / chunk)), dim3(256), 0, stream, mask, mask_offset, nrows, chunk, row_base,
                     out_probe, out_build, out_capacity, cursor);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_radix_emit_staged(radix_join_args const& a, radix_join_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_radix_args<radix_join_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"join_retrieve", stream};
  hipLaunchKernelGGL(k_radix_emit_staged, dim3(a.nparts), dim3(256), 0, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

}  // namespace cudf::detail::join
