// SPDX-License-Identifier: Apache-2.0
// gfx950 kernels of the hash-groupby engine (see engine.hpp for the design). Every kernel is HBM- or
// LDS-bound integer/byte work: 64-lane waves, LDS hash tables with native ds_* atomics, LDS-staged
// multi-split for coalesced partition writes; no MFMA, no global atomics on the per-row path.
// Kernel arguments are read through a pointer to device memory (scalar loads), never by value.
#include "engine.hpp"
#include "../common/profiler.hpp"

#include <cudf/utilities/error.hpp>

namespace cudf::groupby::detail {

using cudf::detail::col_is_valid;
using cudf::detail::col_load_acc_bits;
using cudf::detail::col_load_bits;
using cudf::detail::gload;
using cudf::detail::gstore;
using cudf::detail::mix64;
using cudf::detail::normalize_key_bits;
using cudf::detail::u64x2;

namespace {

constexpr uint32_t ST_EMPTY  = 0;
constexpr uint32_t ST_LOCKED = 1;

__device__ __forceinline__ uint32_t tag_of(uint64_t h) { return (static_cast<uint32_t>(h >> 20) & ~3u) | 2u; }

// Workgroup barrier that orders LDS traffic only. __syncthreads() makes hipcc emit `s_waitcnt vmcnt(0)` in front
// of s_barrier, i.e. every wave drains ALL its outstanding global loads and stores at every barrier — in the
// partition kernel that serialised the tile's HBM traffic with its LDS phases (nothing else hides it at one
// workgroup per CU). Here global loads/stores stay in flight across the barrier; the compiler still waits on
// vmcnt where a loaded register is first used.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__global__ void k_store_args(T v, T* dst)
{
  *dst = v;
}

// ------------------------------------------------------------------ record building from columns
// Validity words of one row: keynulls bit c = key column c NULL; valvalid bit v = value column v valid.
__device__ __forceinline__ void row_validity(plan_dev const& p, int64_t row, uint32_t& keynulls, uint32_t& valvalid)
{
  keynulls = 0;
  valvalid = 0;
  for (int c = 0; c < p.ncols; ++c) {
    if (p.cols[c].mask == nullptr) {
      if (c >= p.nkeycols) valvalid |= 1u << (c - p.nkeycols);
      continue;
    }
    bool const v = col_is_valid(p.cols[c], row);
    if (c < p.nkeycols) {
      if (!v) keynulls |= 1u << c;
    } else if (v) {
      valvalid |= 1u << (c - p.nkeycols);
    }
  }
}

__device__ __forceinline__ uint32_t half_bits(plan_dev const& p, int8_t src, int64_t row, uint32_t keynulls,
                                              uint32_t valvalid)
{
  if (src == H_NONE) return 0;
  if (src == H_KEYNULLS) return keynulls;
  if (src == H_VALVALID) return valvalid;
  if ((keynulls >> src) & 1u) return 0;  // NULL key element: data zeroed so equal NULLs compare equal
  return static_cast<uint32_t>(normalize_key_bits(col_load_bits(p.cols[src], row), p.cols[src].cls));
}

__device__ __forceinline__ uint64_t unit_bits(plan_dev const& p, int u, int64_t row, uint32_t keynulls,
                                              uint32_t valvalid)
{
  // one aligned 32-bit scalar load instead of four byte loads
  uint32_t const w = reinterpret_cast<uint32_t const*>(p.unit)[u];
  unit_desc d;
  d.full   = static_cast<int8_t>(w);
  d.lo     = static_cast<int8_t>(w >> 8);
  d.hi     = static_cast<int8_t>(w >> 16);
  d.is_key = static_cast<int8_t>(w >> 24);
  if (d.full) {
    if (d.is_key) {
      if ((keynulls >> d.lo) & 1u) return 0;
      return normalize_key_bits(col_load_bits(p.cols[d.lo], row), p.cols[d.lo].cls);
    }
    return col_load_acc_bits(p.cols[d.lo], row);
  }
  return static_cast<uint64_t>(half_bits(p, d.lo, row, keynulls, valvalid)) |
         (static_cast<uint64_t>(half_bits(p, d.hi, row, keynulls, valvalid)) << 32);
}

// Key units of one row from the columns; false if the row is dropped (null_policy::EXCLUDE).
template <int KUT, bool SIMPLE>
__device__ __forceinline__ bool build_key_units(plan_dev const& p, int64_t row, uint64_t (&key)[KUT], uint32_t& valvalid)
{
  if constexpr (SIMPLE) {
    valvalid = 0xffffffffu;
#pragma unroll
    for (int u = 0; u < KUT; ++u) key[u] = (u < p.KU) ? gload(p.simple_base[u] + row) : 0;
    return true;
  } else {
    uint32_t keynulls;
    row_validity(p, row, keynulls, valvalid);
    if (p.drop_null_keys && keynulls != 0) return false;
#pragma unroll
    for (int u = 0; u < KUT; ++u) key[u] = (u < p.KU) ? unit_bits(p, u, row, keynulls, valvalid) : 0;
    return true;
  }
}

template <int KUT>
__device__ __forceinline__ uint64_t hash_key_units(plan_dev const& p, uint64_t const (&key)[KUT])
{
  uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
  for (int u = 0; u < KUT; ++u)
    if (u < p.KU) h = mix64(h ^ (key[u] & p.key_mask[u]));
  return h;
}

// Column-at-a-time record building for a batch of NR rows per thread: every column / unit descriptor is decoded
// ONCE per batch (scalar work) and the inner loops over the rows are straight typed loads. The row-at-a-time form
// above re-decodes the descriptors for every row and is bound by the CU's scalar ALU (C4: 12.5 ms histogram).
template <int NR>
__device__ __forceinline__ void batch_validity(plan_dev const& p, int64_t const (&row)[NR], bool (&live)[NR],
                                               uint32_t (&keynulls)[NR], uint32_t (&valvalid)[NR])
{
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    keynulls[k] = 0;
    valvalid[k] = 0;
  }
  for (int c = 0; c < p.ncols; ++c) {
    bitmask_type const* mask = p.cols[c].mask;
    int const off            = p.cols[c].offset;
    bool const is_key        = c < p.nkeycols;
    uint32_t const bit       = is_key ? (1u << c) : (1u << (c - p.nkeycols));
    if (mask == nullptr) {
      if (!is_key) {
#pragma unroll
        for (int k = 0; k < NR; ++k) valvalid[k] |= bit;
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      if (!live[k]) continue;
      int64_t const b = static_cast<int64_t>(off) + row[k];
      bool const v    = (gload(mask + (b >> 5)) >> (b & 31)) & 1u;
      if (is_key) keynulls[k] |= v ? 0u : bit;
      else valvalid[k] |= v ? bit : 0u;
    }
  }
  if (p.drop_null_keys) {
#pragma unroll
    for (int k = 0; k < NR; ++k) live[k] = live[k] && keynulls[k] == 0;
  }
}

// raw element bits of column c for NR rows (zero-extended)
template <int NR>
__device__ __forceinline__ void batch_load_bits(device_column const& col, int64_t const (&row)[NR], bool const (&live)[NR],
                                                uint64_t (&out)[NR])
{
  int64_t const off = col.offset;
  switch (col.width) {
    case 1:
#pragma unroll
      for (int k = 0; k < NR; ++k) out[k] = live[k] ? gload(static_cast<uint8_t const*>(col.head) + off + row[k]) : 0;
      break;
    case 2:
#pragma unroll
      for (int k = 0; k < NR; ++k) out[k] = live[k] ? gload(static_cast<uint16_t const*>(col.head) + off + row[k]) : 0;
      break;
    case 4:
#pragma unroll
      for (int k = 0; k < NR; ++k) out[k] = live[k] ? gload(static_cast<uint32_t const*>(col.head) + off + row[k]) : 0;
      break;
    default:
#pragma unroll
      for (int k = 0; k < NR; ++k) out[k] = live[k] ? gload(static_cast<uint64_t const*>(col.head) + off + row[k]) : 0;
  }
}

__device__ __forceinline__ uint64_t to_acc_bits(uint64_t raw, int cls, int width)
{
  switch (cls) {
    case cudf::detail::CLS_SINT:
      switch (width) {
        case 1: return static_cast<uint64_t>(static_cast<int64_t>(static_cast<int8_t>(raw)));
        case 2: return static_cast<uint64_t>(static_cast<int64_t>(static_cast<int16_t>(raw)));
        case 4: return static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(raw)));
        default: return raw;
      }
    case cudf::detail::CLS_BOOL: return raw != 0;
    case cudf::detail::CLS_F32: return __double_as_longlong(static_cast<double>(__uint_as_float(static_cast<uint32_t>(raw))));
    default: return raw;
  }
}

// One 32-bit half of a unit for NR rows.
template <int NR>
__device__ __forceinline__ void batch_half(plan_dev const& p, int8_t src, int64_t const (&row)[NR], bool const (&live)[NR],
                                           uint32_t const (&keynulls)[NR], uint32_t const (&valvalid)[NR], uint32_t (&out)[NR])
{
  if (src == H_NONE) {
#pragma unroll
    for (int k = 0; k < NR; ++k) out[k] = 0;
  } else if (src == H_KEYNULLS) {
#pragma unroll
    for (int k = 0; k < NR; ++k) out[k] = keynulls[k];
  } else if (src == H_VALVALID) {
#pragma unroll
    for (int k = 0; k < NR; ++k) out[k] = valvalid[k];
  } else {
    device_column const col = p.cols[src];
    uint64_t raw[NR];
    batch_load_bits<NR>(col, row, live, raw);
#pragma unroll
    for (int k = 0; k < NR; ++k)
      out[k] = ((keynulls[k] >> src) & 1u) ? 0u : static_cast<uint32_t>(normalize_key_bits(raw[k], col.cls));
  }
}

// Units [0, nunits) of NR rows; rows with live[k] == false are left untouched. UT bounds the static unroll.
template <int NR, int UT>
__device__ __forceinline__ void batch_units(plan_dev const& p, int nunits, int64_t const (&row)[NR], bool (&live)[NR],
                                            uint64_t (&rec)[NR][UT], uint32_t (&valvalid)[NR])
{
  uint32_t keynulls[NR];
  batch_validity<NR>(p, row, live, keynulls, valvalid);
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    if (u >= nunits) break;
    uint32_t const w = reinterpret_cast<uint32_t const*>(p.unit)[u];
    int8_t const full = static_cast<int8_t>(w), lo = static_cast<int8_t>(w >> 8), hi = static_cast<int8_t>(w >> 16),
                 is_key = static_cast<int8_t>(w >> 24);
    if (full) {
      device_column const col = p.cols[lo];
      uint64_t raw[NR];
      batch_load_bits<NR>(col, row, live, raw);
      if (is_key) {
#pragma unroll
        for (int k = 0; k < NR; ++k) rec[k][u] = ((keynulls[k] >> lo) & 1u) ? 0 : normalize_key_bits(raw[k], col.cls);
      } else {
#pragma unroll
        for (int k = 0; k < NR; ++k) rec[k][u] = to_acc_bits(raw[k], col.cls, col.width);
      }
    } else {
      uint32_t l[NR], h[NR];
      batch_half<NR>(p, lo, row, live, keynulls, valvalid, l);
      batch_half<NR>(p, hi, row, live, keynulls, valvalid, h);
#pragma unroll
      for (int k = 0; k < NR; ++k) rec[k][u] = static_cast<uint64_t>(l[k]) | (static_cast<uint64_t>(h[k]) << 32);
    }
  }
}

// ------------------------------------------------------------------ block scan helper
// Exclusive scan of one uint32 per thread across the block; `total` receives the block sum.
// `wave_sums` must hold blockDim.x / 64 entries.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_sums, uint32_t& total)
{
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t const t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wave_sums[wave] = inc;
  lds_barrier();
  if (wave == 0) {
    uint32_t s = lane < nwaves ? wave_sums[lane] : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t const t = __shfl_up(s, o);
      if (lane >= o) s += t;
    }
    if (lane < nwaves) wave_sums[lane] = s;  // inclusive
  }
  lds_barrier();
  uint32_t const wave_off = wave == 0 ? 0 : wave_sums[wave - 1];
  total                   = wave_sums[nwaves - 1];
  lds_barrier();
  return wave_off + inc - v;
}

// ------------------------------------------------------------------ partition: slice bounds
struct slice_range {
  int64_t begin, end;
  int64_t seg_begin;
};
__device__ __forceinline__ slice_range slice_of(part_args const& a, int item)
{
  int const g = item / a.geom.slices, s = item % a.geom.slices;
  int64_t const b = a.from_columns ? 0 : a.seg_offsets[g];
  int64_t const e = a.from_columns ? a.nrows : a.seg_offsets[g + 1];
  int64_t per = (e - b + a.geom.slices - 1) / a.geom.slices;
  per         = (per + 1) & ~int64_t{1};  // even slice starts: rows are loaded in 16-byte pairs on the simple path
  slice_range r;
  r.seg_begin = b;
  r.begin     = min(e, b + per * s);
  r.end       = min(e, r.begin + per);
  return r;
}

// ------------------------------------------------------------------ K_hist
template <bool SIMPLE>
__global__ void __launch_bounds__(1024) k_partition_hist(part_args const* __restrict__ ap)
{
  extern __shared__ uint32_t lds_hist[];
  part_args const& a = *ap;
  plan_dev const& p  = a.plan;
  int const P = a.geom.P, shift = a.geom.shift;
  for (int d = threadIdx.x; d < P; d += blockDim.x) lds_hist[d] = 0;
  __syncthreads();
  slice_range const sr = slice_of(a, blockIdx.x);
  int const U          = p.KU + p.NPAY;
  int const from_cols  = a.from_columns;
  constexpr int R      = 4;
  int64_t const B      = blockDim.x;
  for (int64_t base = sr.begin; base < sr.end; base += R * B) {
    uint64_t key[R][MAX_KU];
    bool keep[R];
    int64_t row[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      row[k]  = base + k * B + threadIdx.x;
      keep[k] = row[k] < sr.end;
    }
    if (SIMPLE && from_cols && p.simple_vec16 && ((sr.begin & 1) == 0)) {
      // two rows per 16-byte load (rows 2m, 2m+1); which lane counts which row does not matter for a histogram
#pragma unroll
      for (int m = 0; m < R / 2; ++m) {
        int64_t const r = base + (static_cast<int64_t>(m) * B + threadIdx.x) * 2;
        keep[2 * m]     = r < sr.end;
        keep[2 * m + 1] = r + 1 < sr.end;
#pragma unroll
        for (int u = 0; u < MAX_KU; ++u) {
          key[2 * m][u] = key[2 * m + 1][u] = 0;
          if (u < p.KU) {
            if (keep[2 * m + 1]) {
              u64x2 const v     = gload(reinterpret_cast<u64x2 const*>(p.simple_base[u] + r));
              key[2 * m][u]     = v.x;
              key[2 * m + 1][u] = v.y;
            } else if (keep[2 * m]) {
              key[2 * m][u] = gload(p.simple_base[u] + r);
            }
          }
        }
      }
    } else if (from_cols && !SIMPLE) {
      uint32_t vv[R];
      batch_units<R, MAX_KU>(p, p.KU, row, keep, key, vv);
    } else {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (keep[k]) {
          if (from_cols) {
            uint32_t vv;
            keep[k] = build_key_units<MAX_KU, SIMPLE>(p, row[k], key[k], vv);
          } else {
#pragma unroll
            for (int u = 0; u < MAX_KU; ++u) key[k][u] = (u < p.KU) ? gload(a.in_records + row[k] * U + u) : 0;
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (keep[k]) {
        uint64_t const h = hash_key_units<MAX_KU>(p, key[k]);
        atomicAdd(&lds_hist[static_cast<uint32_t>(h >> shift) & static_cast<uint32_t>(P - 1)], 1u);
      }
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < P; d += blockDim.x) gstore(a.counts + static_cast<int64_t>(blockIdx.x) * P + d, lds_hist[d]);
}

// ------------------------------------------------------------------ K_scan: one block per segment
__global__ void __launch_bounds__(1024) k_partition_scan(part_args const* __restrict__ ap)
{
  __shared__ uint32_t wave_sums[16];
  extern __shared__ uint32_t lds_tot[];  // P totals, then P exclusive offsets
  part_args const& a = *ap;
  int const P = a.geom.P, S = a.geom.slices, g = blockIdx.x;
  int64_t const seg_begin = a.from_columns ? 0 : a.seg_offsets[g];
  uint32_t* tot  = lds_tot;
  uint32_t* excl = lds_tot + P;
  for (int d = threadIdx.x; d < P; d += blockDim.x) {
    uint32_t t = 0;
    for (int s = 0; s < S; ++s) t += a.counts[(static_cast<int64_t>(g) * S + s) * P + d];
    tot[d] = t;
  }
  __syncthreads();
  // exclusive scan of tot over d: each thread owns a contiguous run of E entries
  int const E = (P + blockDim.x - 1) / blockDim.x;
  uint32_t local = 0;
  for (int k = 0; k < E; ++k) {
    int const d = threadIdx.x * E + k;
    if (d < P) local += tot[d];
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan(local, wave_sums, total);
  for (int k = 0; k < E; ++k) {
    int const d = threadIdx.x * E + k;
    if (d < P) {
      excl[d] = run;
      run += tot[d];
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < P; d += blockDim.x) {
    int64_t running = seg_begin + excl[d];
    a.out_offsets[static_cast<int64_t>(g) * P + d] = running;
    for (int s = 0; s < S; ++s) {
      int64_t const idx = (static_cast<int64_t>(g) * S + s) * P + d;
      a.item_base[idx]  = running;
      running += a.counts[idx];
    }
  }
  if (g == a.geom.nseg - 1 && threadIdx.x == 0) a.out_offsets[static_cast<int64_t>(a.geom.nseg) * P] = seg_begin + total;
}

// ------------------------------------------------------------------ K_scatter
// One workgroup per (segment, slice). Per tile of T = blockDim * RPT rows: rank rows inside their partition
// with an LDS histogram, exclusive-scan the histogram, stage the records in LDS in partition order and write
// them out so that consecutive lanes write consecutive records of one partition (runs of T/P records).
// LDS layout: stage[T*U] u64 | delta[P] i64 | hist[P] u32 | pid[T] u16 | wave_sums[16] u32
// EXACT: the record has exactly UT units (all `u < U` predicates fold away; 16-byte records move as one
// ds_write_b128 / global_store_dwordx4).
template <int UT, int RPT, bool SIMPLE, bool EXACT>
__global__ void __launch_bounds__(1024) k_partition_scatter(part_args const* __restrict__ ap)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  part_args const& a = *ap;
  plan_dev const& p  = a.plan;
  int const P = a.geom.P, shift = a.geom.shift, B = blockDim.x, T = B * RPT;
  int const U  = EXACT ? UT : (p.KU + p.NPAY);  // <= UT
  int const KU = p.KU;
  uint64_t* stage     = reinterpret_cast<uint64_t*>(lds_raw);
  int64_t* delta      = reinterpret_cast<int64_t*>(stage + static_cast<size_t>(T) * U);
  uint32_t* hist      = reinterpret_cast<uint32_t*>(delta + P);
  uint16_t* pid       = reinterpret_cast<uint16_t*>(hist + P);
  uint32_t* wave_sums = reinterpret_cast<uint32_t*>(pid + T + (T & 1));

  int const item       = blockIdx.x;
  slice_range const sr = slice_of(a, item);
  int const from_cols  = a.from_columns;
  uint64_t const* in_records = a.in_records;
  uint64_t* out_records      = a.out_records;
  uint64_t kmask[UT < MAX_KU ? UT : MAX_KU];
#pragma unroll
  for (int u = 0; u < (UT < MAX_KU ? UT : MAX_KU); ++u) kmask[u] = u < KU ? p.key_mask[u] : 0;
  uint64_t const* sbase[UT];
  if constexpr (SIMPLE) {
#pragma unroll
    for (int u = 0; u < UT; ++u) sbase[u] = u < U ? p.simple_base[u] : nullptr;
  }
  // Thread t owns partitions d = t*MAXE + k: their running output cursor lives in registers.
  constexpr int MAXE = 2;  // P <= 2 * B
  int64_t cursor[MAXE], region_end[MAXE];
  int const optimistic = a.optimistic;
  __shared__ int s_abort;
  if (threadIdx.x == 0) s_abort = 0;
#pragma unroll
  for (int k = 0; k < MAXE; ++k) {
    int const d = threadIdx.x * MAXE + k;
    if (optimistic) {
      cursor[k]     = (static_cast<int64_t>(d) * a.geom.slices + item) * a.region_cap;
      region_end[k] = cursor[k] + a.region_cap;
    } else {
      cursor[k]     = d < P ? a.item_base[static_cast<int64_t>(item) * P + d] : 0;
      region_end[k] = INT64_MAX;
    }
    if (d < P) hist[d] = 0;
  }
  lds_barrier();

  uint64_t rec[RPT][UT];
  bool keep[RPT];
  // loads one tile into registers (all loads issued back to back)
  bool const vec16 = SIMPLE && (RPT % 2 == 0) && p.simple_vec16 && ((sr.begin & 1) == 0);
  auto load_tile = [&](int64_t tile) {
    if constexpr (SIMPLE && (RPT % 2 == 0)) {
      if (vec16) {
        // 8-byte-per-lane loads run at 0.54-0.70x the 16-byte rate (the load issue was 51 % of the tile time):
        // each lane takes rows (2m, 2m+1) of every column with one global_load_dwordx4
#pragma unroll
        for (int m = 0; m < RPT / 2; ++m) {
          int64_t const r = tile + (static_cast<int64_t>(m) * B + threadIdx.x) * 2;
          keep[2 * m]     = r < sr.end;
          keep[2 * m + 1] = r + 1 < sr.end;
          if (keep[2 * m + 1]) {
#pragma unroll
            for (int u = 0; u < UT; ++u) {
              if (u < U) {
                u64x2 const v     = gload(reinterpret_cast<u64x2 const*>(sbase[u] + r));
                rec[2 * m][u]     = v.x;
                rec[2 * m + 1][u] = v.y;
              }
            }
          } else if (keep[2 * m]) {
#pragma unroll
            for (int u = 0; u < UT; ++u) rec[2 * m][u] = (u < U) ? gload(sbase[u] + r) : 0;
          }
        }
        return;
      }
    }
    if (!SIMPLE && from_cols) {
      int64_t row[RPT];
      uint32_t vv[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        row[k]  = tile + static_cast<int64_t>(k) * B + threadIdx.x;
        keep[k] = row[k] < sr.end;
      }
      batch_units<RPT, UT>(p, U, row, keep, rec, vv);
      return;
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      int64_t const r = tile + static_cast<int64_t>(k) * B + threadIdx.x;
      keep[k]         = r < sr.end;
      if (keep[k]) {
        if constexpr (SIMPLE) {
          if (vec16) continue;  // loaded two rows at a time below
#pragma unroll
          for (int u = 0; u < UT; ++u) rec[k][u] = (u < U) ? gload(sbase[u] + r) : 0;
        } else if constexpr (EXACT && UT == 2) {
          u64x2 const v = gload(reinterpret_cast<u64x2 const*>(in_records) + r);
          rec[k][0]     = v.x;
          rec[k][1]     = v.y;
        } else {
#pragma unroll
          for (int u = 0; u < UT; ++u) rec[k][u] = (u < U) ? gload(in_records + r * U + u) : 0;
        }
      }
    }
  };
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool const stamp = a.stamps != nullptr && threadIdx.x == 0;
  unsigned long long t_prev = stamp ? __builtin_amdgcn_s_memtime() : 0;
  auto mark = [&](int i) {
    if (stamp) {
      unsigned long long const t = __builtin_amdgcn_s_memtime();
      ph[i] += t - t_prev;
      t_prev = t;
    }
  };
  if (sr.begin < sr.end) load_tile(sr.begin);
  for (int64_t tile = sr.begin; tile < sr.end; tile += T) {
    uint32_t dig[RPT], rank[RPT];
    // phase 1b: hash and rank within (tile, partition)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      if (keep[k]) {
        uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
        for (int u = 0; u < (UT < MAX_KU ? UT : MAX_KU); ++u)
          if (u < KU) h = mix64(h ^ (rec[k][u] & kmask[u]));
        dig[k]  = static_cast<uint32_t>(h >> shift) & static_cast<uint32_t>(P - 1);
        rank[k] = atomicAdd(&hist[dig[k]], 1u);
      }
    }
    mark(0);
    lds_barrier();
    mark(1);
    // phase 2: hist -> exclusive local offsets (in place); delta = global cursor - local offset
    uint32_t hv[MAXE], local = 0;
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      int const d = threadIdx.x * MAXE + k;
      hv[k]       = d < P ? hist[d] : 0;
      local += hv[k];
    }
    uint32_t tile_count;
    uint32_t run = block_exclusive_scan(local, wave_sums, tile_count);
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      int const d = threadIdx.x * MAXE + k;
      if (d < P) {
        hist[d]  = run;
        delta[d] = cursor[k] - static_cast<int64_t>(run);
        cursor[k] += hv[k];
        run += hv[k];
        if (cursor[k] > region_end[k]) s_abort = 1;  // optimistic region too small: nothing of this tile is written
      }
    }
    lds_barrier();
    mark(2);
    if (s_abort) {
      if (threadIdx.x == 0) *a.overflow = 1;
      return;
    }
    // phase 3: stage records in partition order
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      if (keep[k]) {
        uint32_t const pos = hist[dig[k]] + rank[k];
        if constexpr (EXACT && UT == 2) {
          reinterpret_cast<u64x2*>(stage)[pos] = u64x2{rec[k][0], rec[k][1]};
        } else {
#pragma unroll
          for (int u = 0; u < UT; ++u)
            if (u < U) stage[static_cast<size_t>(pos) * U + u] = rec[k][u];
        }
        pid[pos] = static_cast<uint16_t>(dig[k]);
      }
    }
    // the records are staged: the registers are free, so the NEXT tile's loads go out now and fly under the
    // write-out phase (one workgroup per CU: nothing else would hide their latency)
    mark(3);
    if (tile + T < sr.end) load_tile(tile + T);
    mark(4);
    lds_barrier();
    mark(5);
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      int const d = threadIdx.x * MAXE + k;
      if (d < P) hist[d] = 0;
    }
    // phase 4: coalesced write-out; consecutive staged records of one partition go to consecutive slots
    for (uint32_t j = threadIdx.x; j < tile_count; j += B) {
      int64_t const dst = delta[pid[j]] + static_cast<int64_t>(j);
      if constexpr (EXACT && UT == 2) {
        gstore(reinterpret_cast<u64x2*>(out_records) + dst, reinterpret_cast<u64x2 const*>(stage)[j]);
      } else {
#pragma unroll
        for (int u = 0; u < UT; ++u)
          if (u < U) gstore(out_records + dst * U + u, stage[static_cast<size_t>(j) * U + u]);
      }
    }
    mark(6);
    lds_barrier();
    mark(7);
  }
  if (stamp) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a.stamps[static_cast<int64_t>(blockIdx.x) * 8 + i] = ph[i];
  }
  if (optimistic) {
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      int const d = threadIdx.x * MAXE + k;
      if (d < P) a.region_count[static_cast<int64_t>(d) * a.geom.slices + item] = static_cast<int32_t>(cursor[k] - (region_end[k] - a.region_cap));
    }
  }
}

// ------------------------------------------------------------------ K_aggregate
__device__ __forceinline__ uint64_t acc_identity(int op)
{
  switch (op) {
    case MIN_I64: return static_cast<uint64_t>(INT64_MAX);
    case MIN_U64: return UINT64_MAX;
    case MIN_F64: return 0x7ff0000000000000ull;  // +inf
    case MAX_I64: return static_cast<uint64_t>(INT64_MIN);
    case MAX_U64: return 0;
    case MAX_F64: return 0xfff0000000000000ull;  // -inf
    default: return 0;                            // ADD_I64 / ADD_F64
  }
}

__device__ __forceinline__ void lds_merge(uint64_t* slot, int op, uint64_t v)
{
  switch (op) {
    case ADD_I64: atomicAdd(reinterpret_cast<unsigned long long*>(slot), static_cast<unsigned long long>(v)); break;
    case ADD_F64: atomicAdd(reinterpret_cast<double*>(slot), __longlong_as_double(static_cast<long long>(v))); break;
    case MIN_I64: atomicMin(reinterpret_cast<long long*>(slot), static_cast<long long>(v)); break;
    case MIN_U64: atomicMin(reinterpret_cast<unsigned long long*>(slot), static_cast<unsigned long long>(v)); break;
    case MAX_I64: atomicMax(reinterpret_cast<long long*>(slot), static_cast<long long>(v)); break;
    case MAX_U64: atomicMax(reinterpret_cast<unsigned long long*>(slot), static_cast<unsigned long long>(v)); break;
    case MIN_F64:
      __hip_atomic_fetch_min(reinterpret_cast<double*>(slot), __longlong_as_double(static_cast<long long>(v)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    case MAX_F64:
      __hip_atomic_fetch_max(reinterpret_cast<double*>(slot), __longlong_as_double(static_cast<long long>(v)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
  }
}

// Finds or claims the slot of `key` in the LDS table. Returns -1 if the table is saturated.
template <int KUT>
__device__ __forceinline__ int lds_find_or_insert(int KU, uint64_t const (&kmask)[KUT], uint32_t* st, uint64_t* keys,
                                                  int cap, uint64_t const (&key)[KUT], uint64_t h, uint32_t* nfilled,
                                                  int fill_limit, int32_t* overflow_flag)
{
  uint32_t const tag = tag_of(h);
  int slot = static_cast<int>((static_cast<uint64_t>(static_cast<uint32_t>(h)) * static_cast<uint32_t>(cap)) >> 32);
  for (int probes = 0; probes < cap; ++probes) {
    uint32_t s = __hip_atomic_load(&st[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (s == ST_EMPTY) {
      uint32_t const old = atomicCAS(&st[slot], ST_EMPTY, ST_LOCKED);
      if (old == ST_EMPTY) {
#pragma unroll
        for (int u = 0; u < KUT; ++u)
          if (u < KU) keys[static_cast<uint32_t>(u * cap + slot)] = key[u] & kmask[u];
        // publish: key words first, then the tag (LDS executes a wave's accesses in order; the release fence
        // keeps the compiler from reordering and waits for the key stores)
        __hip_atomic_store(&st[slot], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t const n = atomicAdd(nfilled, 1u);
        if (static_cast<int>(n) >= fill_limit) *overflow_flag = 1;
        return slot;
      }
      s = old;
    }
    if (s == ST_LOCKED) {
      --probes;  // owner is publishing: re-read the same slot
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    if (s == tag) {
      bool eq = true;
#pragma unroll
      for (int u = 0; u < KUT; ++u)
        if (u < KU) eq = eq && (keys[static_cast<uint32_t>(u * cap + slot)] == (key[u] & kmask[u]));
      if (eq) return slot;
    }
    slot = slot + 1 == cap ? 0 : slot + 1;
  }
  *overflow_flag = 1;
  return -1;
}

// INPUT: agg_input. KUT: key units held in registers. PAYT: payload units of a RECORD prefetched into
// registers together with the key (0 = payload fetched lazily per accumulator: column input, wide records).
// NACCT: compile-time bound of the accumulator loop (descriptors sit in registers, statically indexed).
// EXACT: the input record has exactly KUT + PAYT units (a 16-byte record is one global_load_dwordx4).
// SIG: compile-time accumulator signature (0 = read the descriptors at run time). For the hot shapes every
// descriptor test folds away and the accumulate step is straight-line ds_* atomics: the generic form spends
// ~200 scalar instructions per 64 rows on descriptor branches and is bound by the CU's single scalar ALU.
// 12 bits per accumulator: op(4) | src(2) | pay+1 (3) | vbit+1 (3); accumulator count in bits 60..63.
constexpr uint64_t sig_acc(int op, int src, int pay, int vbit)
{
  return static_cast<uint64_t>(op) | (static_cast<uint64_t>(src) << 4) | (static_cast<uint64_t>(pay + 1) << 6) |
         (static_cast<uint64_t>(vbit + 1) << 9);
}
constexpr uint64_t make_sig(int n, uint64_t a0 = 0, uint64_t a1 = 0, uint64_t a2 = 0, uint64_t a3 = 0)
{
  return (static_cast<uint64_t>(n) << 60) | a0 | (a1 << 12) | (a2 << 24) | (a3 << 36);
}
constexpr int sig_n(uint64_t s) { return static_cast<int>(s >> 60); }
constexpr int sig_op(uint64_t s, int q) { return static_cast<int>((s >> (12 * q)) & 0xf); }
constexpr int sig_src(uint64_t s, int q) { return static_cast<int>((s >> (12 * q + 4)) & 0x3); }
constexpr int sig_pay(uint64_t s, int q) { return static_cast<int>((s >> (12 * q + 6)) & 0x7) - 1; }
constexpr int sig_vbit(uint64_t s, int q) { return static_cast<int>((s >> (12 * q + 9)) & 0x7) - 1; }

template <int INPUT, int KUT, int PAYT, int NACCT, bool SIMPLE, bool EXACT, uint64_t SIG = 0>
__global__ void __launch_bounds__(1024, (NACCT <= 4 && KUT <= 2) ? 8 : 4) k_aggregate(agg_args const* __restrict__ ap)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  agg_args const& a = *ap;
  plan_dev const& p = a.plan;
  int const cap     = a.geom.cap;
  constexpr bool STATIC_SIG = SIG != 0;
  int const KU = EXACT ? KUT : p.KU, NACC = STATIC_SIG ? sig_n(SIG) : p.NACC;
  uint64_t* keys = reinterpret_cast<uint64_t*>(lds_raw);                 // [KU][cap]
  uint64_t* accs = keys + static_cast<size_t>(KU) * cap;                   // [NACC][cap]
  uint32_t* st   = reinterpret_cast<uint32_t*>(accs + static_cast<size_t>(NACC) * cap);  // [cap]
  __shared__ uint32_t s_nfilled, s_dump;
  __shared__ int32_t s_overflow;

  if (threadIdx.x == 0) {
    s_nfilled  = 0;
    s_dump     = 0;
    s_overflow = 0;
  }
  for (int s = threadIdx.x; s < cap; s += blockDim.x) st[s] = ST_EMPTY;
  for (int q = 0; q < NACC; ++q) {
    uint64_t const id = acc_identity(p.acc[q].op);
    for (int s = threadIdx.x; s < cap; s += blockDim.x) accs[static_cast<size_t>(q) * cap + s] = id;
  }
  uint64_t kmask[KUT];
#pragma unroll
  for (int u = 0; u < KUT; ++u) kmask[u] = u < KU ? p.key_mask[u] : 0;
  // accumulator descriptors live in (scalar) registers for the whole kernel: no memory access per row
  int acc_op[NACCT], acc_src[NACCT], acc_pay[NACCT], acc_vbit[NACCT];
#pragma unroll
  for (int j = 0; j < NACCT; ++j) {
    uint32_t const w = j < NACC ? reinterpret_cast<uint32_t const*>(p.acc)[j] : 0u;
    acc_op[j]        = STATIC_SIG ? sig_op(SIG, j) : static_cast<int8_t>(w);
    acc_src[j]       = STATIC_SIG ? sig_src(SIG, j) : static_cast<int8_t>(w >> 8);
    acc_pay[j]       = STATIC_SIG ? sig_pay(SIG, j) : static_cast<int8_t>(w >> 16);
    acc_vbit[j]      = STATIC_SIG ? sig_vbit(SIG, j) : static_cast<int8_t>(w >> 24);
  }
  __syncthreads();

  int const item  = blockIdx.x;
  int const RU    = KU + p.NPAY;  // raw record units
  int const PU    = KU + NACC;    // partial record units
  int const U     = EXACT ? (KUT + PAYT) : (INPUT == IN_RAW_RECORDS ? RU : PU);
  int const fill_limit = a.geom.fill_limit;
  int const flags_unit = p.flags_unit, flags_hi = p.flags_hi;
  uint64_t const* records = a.records;

  // accumulators of one row whose LDS slot is known
  auto accumulate = [&](int64_t r, int slot, uint64_t const (&pay)[PAYT > 0 ? PAYT : 1], uint32_t valvalid) {
    int last_pay   = -1;
    uint64_t value = 0;
#pragma unroll
    for (int q = 0; q < NACCT; ++q) {
      if (q >= NACC) break;
      uint64_t* tgt = accs + (static_cast<uint32_t>(q) * static_cast<uint32_t>(cap) + static_cast<uint32_t>(slot));
      if constexpr (INPUT == IN_PARTIAL_RECORDS) {
        uint64_t v;
        if constexpr (PAYT > 0) v = q < PAYT ? pay[q < PAYT ? q : 0] : 0;
        else v = gload(records + r * U + KU + q);
        lds_merge(tgt, acc_op[q], v);
      } else {
        if (acc_src[q] == SRC_ONE) {
          lds_merge(tgt, ADD_I64, 1);
          continue;
        }
        bool const valid = acc_vbit[q] < 0 || ((valvalid >> acc_vbit[q]) & 1u);
        if (!valid) continue;
        if (acc_src[q] == SRC_ONE_IF_VALID) {
          lds_merge(tgt, ADD_I64, 1);
          continue;
        }
        if (acc_pay[q] != last_pay) {
          if constexpr (INPUT == IN_COLUMNS) {
            if constexpr (SIMPLE) value = gload(p.simple_base[KU + acc_pay[q]] + r);
            else value = col_load_acc_bits(p.cols[p.nkeycols + acc_pay[q]], r);
          } else if constexpr (PAYT > 0) {
#pragma unroll
            for (int w = 0; w < PAYT; ++w)
              if (acc_pay[q] == w) value = pay[w];
          } else {
            value = gload(records + r * U + KU + acc_pay[q]);
          }
          last_pay = acc_pay[q];
        }
        uint64_t v = value;
        if (acc_src[q] == SRC_SQUARE) {
          if (acc_op[q] == ADD_F64) {
            double const x = __longlong_as_double(static_cast<long long>(v));
            v              = static_cast<uint64_t>(__double_as_longlong(x * x));
          } else {
            v = v * v;
          }
        }
        lds_merge(tgt, acc_op[q], v);
      }
    }
  };
  auto hash_of = [&](uint64_t const (&key)[KUT]) {
    uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
    for (int u = 0; u < KUT; ++u)
      if (u < KU) h = mix64(h ^ (key[u] & kmask[u]));
    return h;
  };
  // one row, unbatched (tail rows)
  auto process = [&](int64_t r, uint64_t const (&key)[KUT], uint64_t const (&pay)[PAYT > 0 ? PAYT : 1], uint32_t valvalid) {
    uint64_t const h = hash_of(key);
    int const slot   = lds_find_or_insert<KUT>(KU, kmask, st, keys, cap, key, h, &s_nfilled, fill_limit, &s_overflow);
    if (slot >= 0) accumulate(r, slot, pay, valvalid);
  };
  // loads one row; false if the row is dropped
  auto load_row = [&](int64_t r, uint64_t (&key)[KUT], uint64_t (&pay)[PAYT > 0 ? PAYT : 1], uint32_t& valvalid) -> bool {
    valvalid = 0xffffffffu;
    if constexpr (INPUT == IN_COLUMNS) {
      return build_key_units<KUT, SIMPLE>(p, r, key, valvalid);
    } else if constexpr (EXACT && KUT == 1 && PAYT == 1) {
      u64x2 const v = gload(reinterpret_cast<u64x2 const*>(records) + r);
      key[0]        = v.x;
      pay[0]        = v.y;
      return true;
    } else {
#pragma unroll
      for (int u = 0; u < KUT; ++u) key[u] = (u < KU) ? gload(records + r * U + u) : 0;
      if constexpr (PAYT > 0) {
#pragma unroll
        for (int v = 0; v < PAYT; ++v) pay[v] = gload(records + r * U + KU + v);
      }
      if (INPUT == IN_RAW_RECORDS && flags_unit >= 0)
        valvalid = gload(reinterpret_cast<uint32_t const*>(records + r * U + flags_unit) + flags_hi);
      return true;
    }
  };

  int nsrc = 1, src0 = item;
  if (a.seg == SEG_STRIDED) {
    src0 = item * a.fan;
    nsrc = min(a.fan, a.nsrc - src0);
    // an upstream kernel (optimistic partition, previous merge round) gave up: its counts are not valid
    if (*a.overflow != 0) nsrc = 0;
  }
  constexpr int R = (KUT + PAYT <= 2) ? 4 : 2;  // rows in flight per thread
  int64_t const B = blockDim.x;
  // Work is dealt to WAVES in batches of W = R*64 consecutive records. One big segment (a partition, a row chunk):
  // the waves interleave batches. Many short segments (the per-slice regions of an optimistic partition, the
  // partial tables of a merge round): each wave takes whole segments, so short segments still run batched.
  constexpr int64_t W = R * 64;
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  bool const multi = a.seg == SEG_STRIDED;
  for (int sidx = multi ? wave : 0; sidx < nsrc; sidx += multi ? nwaves : 1) {
    int64_t begin, end;
    if (a.seg == SEG_ROW_CHUNKS) {
      begin = static_cast<int64_t>(item) * a.chunk;
      end   = min(a.nrows, begin + a.chunk);
    } else if (a.seg == SEG_OFFSETS) {
      begin = a.offsets[item];
      end   = a.offsets[item + 1];
    } else {
      begin = static_cast<int64_t>(src0 + sidx) * a.src_stride;
      end   = begin + min<int64_t>(max(a.src_count[src0 + sidx], 0), a.src_stride);
    }
    int64_t const nbatches = (end - begin) / W;
    // main loop: R full rows per lane, all loads issued before the LDS work
    for (int64_t b = multi ? 0 : wave; b < nbatches; b += multi ? 1 : nwaves) {
      int64_t const base = begin + b * W;
      uint64_t key[R][KUT];
      uint64_t pay[R][PAYT > 0 ? PAYT : 1];
      uint32_t valvalid[R];
      bool keep[R];
#pragma unroll
      for (int k = 0; k < R; ++k) keep[k] = load_row(base + k * 64 + lane, key[k], pay[k], valvalid[k]);
      // Batched first probe: the state word and the stored key words of the home slot of all R rows are read
      // with independent ds_reads (one LDS round trip for the common "group already present" case); only
      // rows that miss walk the full claim/probe protocol.
      uint64_t h[R];
      int slot[R];
      uint32_t s0[R], s1[R];
      uint64_t k0[R][KUT], k1[R][KUT];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        h[k]    = hash_of(key[k]);
        slot[k] = static_cast<int>((static_cast<uint64_t>(static_cast<uint32_t>(h[k])) * static_cast<uint32_t>(cap)) >> 32);
        int const nxt = slot[k] + 1 == cap ? 0 : slot[k] + 1;
        s0[k]   = __hip_atomic_load(&st[slot[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        s1[k]   = __hip_atomic_load(&st[nxt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int u = 0; u < KUT; ++u) {
          k0[k][u] = u < KU ? keys[static_cast<uint32_t>(u * cap + slot[k])] : 0;
          k1[k][u] = u < KU ? keys[static_cast<uint32_t>(u * cap + nxt)] : 0;
        }
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (!keep[k]) continue;
        uint32_t const tag = tag_of(h[k]);
        bool hit0 = s0[k] == tag, hit1 = s1[k] == tag;
#pragma unroll
        for (int u = 0; u < KUT; ++u) {
          if (u < KU) {
            hit0 = hit0 && (k0[k][u] == (key[k][u] & kmask[u]));
            hit1 = hit1 && (k1[k][u] == (key[k][u] & kmask[u]));
          }
        }
        int sl = slot[k];
        if (!hit0) {
          // the second slot only counts when the first is occupied by another key (else the key would sit there)
          if (hit1 && s0[k] >= 2) sl = slot[k] + 1 == cap ? 0 : slot[k] + 1;
          else sl = lds_find_or_insert<KUT>(KU, kmask, st, keys, cap, key[k], h[k], &s_nfilled, fill_limit, &s_overflow);
        }
        if (sl >= 0) accumulate(base + k * 64 + lane, sl, pay[k], valvalid[k]);
      }
    }
    // tail: the < W records after the last full batch
    int64_t const tail_begin = begin + nbatches * W;
    for (int64_t r = tail_begin + (multi ? lane : static_cast<int>(threadIdx.x)); r < end; r += multi ? 64 : B) {
      uint64_t key[KUT];
      uint64_t pay[PAYT > 0 ? PAYT : 1];
      uint32_t valvalid;
      if (load_row(r, key, pay, valvalid)) process(r, key, pay, valvalid);
    }
  }
  __syncthreads();
  // dump the table as compact partial records
  uint64_t* out = a.out_records + static_cast<int64_t>(item) * cap * PU;
  for (int s = threadIdx.x; s < cap; s += blockDim.x) {
    if (st[s] >= 2) {
      uint32_t const pos = atomicAdd(&s_dump, 1u);
      uint64_t* o        = out + static_cast<int64_t>(pos) * PU;
      for (int u = 0; u < KU; ++u) gstore(o + u, keys[static_cast<size_t>(u) * cap + s]);
      for (int q = 0; q < NACC; ++q) gstore(o + KU + q, accs[static_cast<size_t>(q) * cap + s]);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.out_count[item] = static_cast<int32_t>(s_dump);
    if (s_overflow) *a.overflow = 1;
  }
}

// ------------------------------------------------------------------ K_finalize
__device__ __forceinline__ void store_elem(void* base, int64_t i, int width, uint64_t bits)
{
  switch (width) {
    case 1: static_cast<uint8_t*>(base)[i] = static_cast<uint8_t>(bits); break;
    case 2: static_cast<uint16_t*>(base)[i] = static_cast<uint16_t>(bits); break;
    case 4: static_cast<uint32_t*>(base)[i] = static_cast<uint32_t>(bits); break;
    default: static_cast<uint64_t*>(base)[i] = bits;
  }
}

__global__ void __launch_bounds__(256) k_finalize(finalize_args const* __restrict__ fa, uint64_t const* records,
                                                  int64_t cap, int64_t const* prefix, int32_t nitems, int64_t total)
{
  plan_dev const& p     = fa->plan;
  finalize_dev const& f = fa->fin;
  int64_t const o = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  bool const live = o < total;
  int const PU    = p.KU + p.NACC;
  uint64_t const* rec = nullptr;
  if (live) {
    // largest item with prefix[item] <= o
    int lo = 0, hi = nitems - 1;
    while (lo < hi) {
      int const mid = (lo + hi + 1) >> 1;
      if (prefix[mid] <= o) lo = mid; else hi = mid - 1;
    }
    rec = records + (static_cast<int64_t>(lo) * cap + (o - prefix[lo])) * PU;
  }
  int const lane = threadIdx.x & 63;
  for (int c = 0; c < f.nout; ++c) {
    out_desc const d = f.out[c];
    bool valid       = live;
    uint64_t bits    = 0;
    if (live) {
      if (d.kind == OUT_KEY) {
        uint64_t const unit = rec[d.key_unit];
        bits                = d.key_full ? unit : (d.key_hi ? (unit >> 32) : (unit & 0xffffffffull));
        if (d.key_null_bit >= 0) {
          uint64_t const kn = rec[d.keynulls_unit];
          uint32_t const w  = d.keynulls_hi ? static_cast<uint32_t>(kn >> 32) : static_cast<uint32_t>(kn);
          valid             = !((w >> d.key_null_bit) & 1u);
        }
      } else {
        uint64_t const a0 = rec[p.KU + d.a0];
        if (d.valid_acc >= 0) valid = static_cast<int64_t>(rec[p.KU + d.valid_acc]) > 0;
        if (d.kind == OUT_COUNT) {
          bits = a0;
        } else if (d.kind == OUT_M2 || d.kind == OUT_VAR || d.kind == OUT_STD) {
          // reference groupby/common/m2_var_std.cu:48-60,152-187
          auto as_double = [&](uint64_t v) {
            return d.cls == cudf::detail::CLS_F64 ? __longlong_as_double(static_cast<long long>(v))
                                                  : static_cast<double>(static_cast<int64_t>(v));
          };
          int64_t const cnt = static_cast<int64_t>(rec[p.KU + d.a2]);
          double const ssq  = as_double(a0);
          double const sm   = as_double(rec[p.KU + d.a1]);
          double const m2   = cnt > 0 ? ssq - sm * sm / static_cast<double>(cnt) : 0.0;
          double out        = m2;
          if (d.kind != OUT_M2) {
            int64_t const df = cnt - d.ddof;
            valid            = cnt > 0 && df > 0;
            out              = valid ? m2 / static_cast<double>(df) : 0.0;
            if (d.kind == OUT_STD) out = sqrt(out);
          } else {
            valid = true;
          }
          bits = static_cast<uint64_t>(__double_as_longlong(out));
        } else if (d.kind == OUT_MEAN) {
          // MEAN = double(SUM) / COUNT_VALID as FLOAT64 (reference hash_compound_agg_finalizer.cu:92-133)
          double const s = d.cls == cudf::detail::CLS_F64 ? __longlong_as_double(static_cast<long long>(a0))
                           : d.cls == cudf::detail::CLS_UINT ? static_cast<double>(a0)
                                                             : static_cast<double>(static_cast<int64_t>(a0));
          double const n = static_cast<double>(static_cast<int64_t>(rec[p.KU + d.a1]));
          bits           = static_cast<uint64_t>(__double_as_longlong(valid ? s / n : 0.0));
        } else {  // OUT_ACC: accumulator class -> output type
          if (d.cls == cudf::detail::CLS_F64 && d.out_cls == cudf::detail::CLS_F32) {
            bits = __float_as_uint(static_cast<float>(__longlong_as_double(static_cast<long long>(a0))));
          } else {
            bits = a0;  // integers truncate to the output width; FLOAT64 passes through
          }
          if (!valid) bits = 0;
        }
      }
      store_elem(d.data, o, d.width, bits);
    }
    if (d.mask != nullptr) {
      unsigned long long const ballot      = __ballot(valid);  // valid implies live
      unsigned long long const live_ballot = __ballot(live);
      if (live && (lane & 31) == 0) d.mask[o >> 5] = static_cast<uint32_t>(ballot >> (lane & 32));
      int const nulls = __popcll(live_ballot & ~ballot);
      if (lane == 0 && nulls) atomicAdd(d.null_count, nulls);
    }
  }
}

// ------------------------------------------------------------------ K_estimate (linear counting on a sample)
__global__ void __launch_bounds__(256) k_estimate(plan_dev const* __restrict__ pp, int64_t nrows, int64_t sample,
                                                  uint32_t* bitmap, int32_t bits_log2)
{
  plan_dev const& p = *pp;
  int64_t const i   = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= sample) return;
  int64_t const row = sample >= nrows ? i : static_cast<int64_t>((static_cast<__int128>(i) * nrows) / sample);
  uint64_t key[MAX_KU];
  uint32_t vv;
  if (!build_key_units<MAX_KU, false>(p, row, key, vv)) return;
  uint64_t const h   = hash_key_units<MAX_KU>(p, key);
  uint32_t const bit = static_cast<uint32_t>(h >> (64 - bits_log2));
  atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
}
__global__ void __launch_bounds__(256) k_popcount(uint32_t const* bitmap, int64_t nwords, uint32_t* out)
{
  int64_t i     = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  uint32_t acc  = 0;
  for (; i < nwords; i += static_cast<int64_t>(gridDim.x) * blockDim.x) acc += __popc(bitmap[i]);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

inline int next_ut(int u)
{
  for (int c : {2, 3, 4, 6, 8, 12, 16})
    if (u <= c) return c;
  return -1;
}

// Opts a kernel into the full 160 KiB of LDS (static + dynamic) once per process.
void allow_full_lds(void const* fn)
{
  hipFuncAttributes attr{};
  CUDF_HIP_TRY(hipFuncGetAttributes(&attr, fn));
  int const dyn = 160 * 1024 - static_cast<int>(attr.sharedSizeBytes);
  CUDF_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
}
}  // namespace

// ------------------------------------------------------------------ launchers
std::size_t aggregate_lds_bytes(plan_dev const& plan, agg_geom const& g)
{
  return static_cast<std::size_t>(g.cap) * (8u * (plan.KU + plan.NACC) + 4u);
}

std::size_t partition_lds_bytes(plan_dev const& plan, part_geom const& g)
{
  int const U = plan.KU + plan.NPAY;
  std::size_t const T = g.tile_rows;
  return T * U * 8 + static_cast<std::size_t>(g.P) * (8 + 4) + (T + (T & 1)) * 2 + 16 * 4;
}

void store_args(part_args const& a, part_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<part_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_partition_hist(part_args const& a, part_args const* d_args, hipStream_t stream)
{
  int const items = a.geom.nseg * a.geom.slices;
  cudf::detail::prof::scope prof_{"partition_hist", stream};
  if (a.plan.simple && a.from_columns)
    hipLaunchKernelGGL(k_partition_hist<true>, dim3(items), dim3(a.geom.block), a.geom.P * sizeof(uint32_t), stream, d_args);
  else
    hipLaunchKernelGGL(k_partition_hist<false>, dim3(items), dim3(a.geom.block), a.geom.P * sizeof(uint32_t), stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_partition_scan(part_args const& a, part_args const* d_args, hipStream_t stream)
{
  cudf::detail::prof::scope prof_{"partition_scan", stream};
  hipLaunchKernelGGL(k_partition_scan, dim3(a.geom.nseg), dim3(1024), 2 * a.geom.P * sizeof(uint32_t), stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

template <int UT, int RPT, bool SIMPLE, bool EXACT>
static void launch_scatter_t(part_args const& a, part_args const* d_args, hipStream_t stream)
{
  part_geom g  = a.geom;
  g.tile_rows  = g.block * RPT;
  auto const lds = partition_lds_bytes(a.plan, g);
  CUDF_EXPECTS(lds <= 160 * 1024, "partition kernel: LDS budget exceeded (fan-out too large for this record width)");
  static bool attr_set = false;
  if (!attr_set) {
    allow_full_lds(reinterpret_cast<void const*>(&k_partition_scatter<UT, RPT, SIMPLE, EXACT>));
    attr_set = true;
  }
  int const items = g.nseg * g.slices;
  cudf::detail::prof::scope prof_{"partition_scatter", stream};
  hipLaunchKernelGGL((k_partition_scatter<UT, RPT, SIMPLE, EXACT>), dim3(items), dim3(g.block), lds, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_partition_scatter(part_args const& a, part_args const* d_args, hipStream_t stream)
{
  CUDF_EXPECTS(a.geom.P <= 2 * a.geom.block, "partition fan-out exceeds 2x the block size");
  int const U       = a.plan.KU + a.plan.NPAY;
  bool const simple = a.plan.simple && a.from_columns;
  switch (next_ut(U)) {
    case 2:
      if (a.geom.tile_rows == 4 * a.geom.block)  // half-size tile: 64 KiB of LDS, two workgroups per CU
        simple ? launch_scatter_t<2, 4, true, true>(a, d_args, stream) : launch_scatter_t<2, 4, false, true>(a, d_args, stream);
      else
        simple ? launch_scatter_t<2, 8, true, true>(a, d_args, stream) : launch_scatter_t<2, 8, false, true>(a, d_args, stream);
      break;
    case 3: simple ? launch_scatter_t<3, 4, true, true>(a, d_args, stream) : launch_scatter_t<3, 4, false, true>(a, d_args, stream); break;
    case 4: simple ? launch_scatter_t<4, 4, true, true>(a, d_args, stream) : launch_scatter_t<4, 4, false, true>(a, d_args, stream); break;
    case 6: launch_scatter_t<6, 2, false, false>(a, d_args, stream); break;
    case 8: launch_scatter_t<8, 2, false, false>(a, d_args, stream); break;
    case 12: launch_scatter_t<12, 1, false, false>(a, d_args, stream); break;
    case 16: launch_scatter_t<16, 1, false, false>(a, d_args, stream); break;
    default: CUDF_FAIL("record too wide for the partition kernel");
  }
}

template <int INPUT, int KUT, int PAYT, int NACCT, bool SIMPLE, bool EXACT, uint64_t SIG = 0>
static void launch_aggregate_n(agg_args const& a, agg_args* d_args, hipStream_t stream)
{
  auto const lds = aggregate_lds_bytes(a.plan, a.geom);
  static bool attr_set = false;
  if (!attr_set) {
    allow_full_lds(reinterpret_cast<void const*>(&k_aggregate<INPUT, KUT, PAYT, NACCT, SIMPLE, EXACT, SIG>));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_store_args<agg_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"aggregate", stream};
  hipLaunchKernelGGL((k_aggregate<INPUT, KUT, PAYT, NACCT, SIMPLE, EXACT, SIG>), dim3(a.nitems), dim3(a.geom.block), lds,
                     stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

// Signature of a plan's accumulators (0 if it does not fit the static encoding).
static uint64_t plan_sig(plan_dev const& p)
{
  if (p.NACC < 1 || p.NACC > 4) return 0;
  uint64_t a[4] = {0, 0, 0, 0};
  for (int q = 0; q < p.NACC; ++q) {
    auto const& d = p.acc[q];
    if (d.pay > 5 || d.valid_bit > 5) return 0;
    a[q] = sig_acc(d.op, d.src, d.pay, d.valid_bit);
  }
  return make_sig(p.NACC, a[0], a[1], a[2], a[3]);
}
// hot signatures with their own instantiation
constexpr uint64_t SIG_SUMF_CNT = make_sig(2, sig_acc(ADD_F64, SRC_VALUE, 0, -1), sig_acc(ADD_I64, SRC_ONE, -1, -1));
constexpr uint64_t SIG_SUMI_CNT = make_sig(2, sig_acc(ADD_I64, SRC_VALUE, 0, -1), sig_acc(ADD_I64, SRC_ONE, -1, -1));
constexpr uint64_t SIG_SUMF     = make_sig(1, sig_acc(ADD_F64, SRC_VALUE, 0, -1));
constexpr uint64_t SIG_SUMI     = make_sig(1, sig_acc(ADD_I64, SRC_VALUE, 0, -1));
constexpr uint64_t SIG_CNT      = make_sig(1, sig_acc(ADD_I64, SRC_ONE, -1, -1));
// C4: MEAN + MIN + MAX of a nullable float64 column -> SUM, COUNT_VALID, MIN, MAX
constexpr uint64_t SIG_MEAN_MIN_MAX_F_NULLS =
  make_sig(4, sig_acc(ADD_F64, SRC_VALUE, 0, 0), sig_acc(ADD_I64, SRC_ONE_IF_VALID, 0, 0), sig_acc(MIN_F64, SRC_VALUE, 0, 0),
           sig_acc(MAX_F64, SRC_VALUE, 0, 0));

template <int INPUT, int KUT, int PAYT, bool SIMPLE, bool EXACT>
static void launch_aggregate_t(agg_args const& a, agg_args* d_args, hipStream_t stream)
{
  if (a.plan.NACC <= 2) return launch_aggregate_n<INPUT, KUT, PAYT, 2, SIMPLE, EXACT>(a, d_args, stream);
  if (a.plan.NACC <= 4) return launch_aggregate_n<INPUT, KUT, PAYT, 4, SIMPLE, EXACT>(a, d_args, stream);
  return launch_aggregate_n<INPUT, KUT, PAYT, MAX_ACC, SIMPLE, EXACT>(a, d_args, stream);
}

template <int INPUT>
static void launch_aggregate_records(agg_args const& a, agg_args* d_args, hipStream_t stream)
{
  int const KU   = a.plan.KU;
  int const npay = INPUT == IN_RAW_RECORDS ? a.plan.NPAY : a.plan.NACC;
  // hot signatures: descriptors folded at compile time
  if constexpr (INPUT == IN_RAW_RECORDS) {
    uint64_t const sig = plan_sig(a.plan);
    if (KU == 1 && npay == 1 && a.plan.flags_unit < 0) {
      if (sig == SIG_SUMF_CNT) return launch_aggregate_n<INPUT, 1, 1, 2, false, true, SIG_SUMF_CNT>(a, d_args, stream);
      if (sig == SIG_SUMI_CNT) return launch_aggregate_n<INPUT, 1, 1, 2, false, true, SIG_SUMI_CNT>(a, d_args, stream);
      if (sig == SIG_SUMF) return launch_aggregate_n<INPUT, 1, 1, 2, false, true, SIG_SUMF>(a, d_args, stream);
      if (sig == SIG_SUMI) return launch_aggregate_n<INPUT, 1, 1, 2, false, true, SIG_SUMI>(a, d_args, stream);
      if (sig == SIG_CNT) return launch_aggregate_n<INPUT, 1, 1, 2, false, true, SIG_CNT>(a, d_args, stream);
    }
    if (KU == 2 && npay == 1 && sig == SIG_MEAN_MIN_MAX_F_NULLS)
      return launch_aggregate_n<INPUT, 2, 1, 4, false, true, SIG_MEAN_MIN_MAX_F_NULLS>(a, d_args, stream);
  }
  // exact shapes get the payload prefetched with the key; everything else fetches it lazily
  if (KU == 1 && npay == 1) return launch_aggregate_t<INPUT, 1, 1, false, true>(a, d_args, stream);
  if (KU == 1 && npay == 2) return launch_aggregate_t<INPUT, 1, 2, false, true>(a, d_args, stream);
  if (KU == 2 && npay == 1) return launch_aggregate_t<INPUT, 2, 1, false, true>(a, d_args, stream);
  if (KU == 2 && npay == 4) return launch_aggregate_t<INPUT, 2, 4, false, true>(a, d_args, stream);
  if (KU <= 1) return launch_aggregate_t<INPUT, 1, 0, false, false>(a, d_args, stream);
  if (KU <= 2) return launch_aggregate_t<INPUT, 2, 0, false, false>(a, d_args, stream);
  return launch_aggregate_t<INPUT, 4, 0, false, false>(a, d_args, stream);
}

void launch_aggregate(agg_args const& a, agg_args* d_args, hipStream_t stream)
{
  if (a.nitems == 0) return;
  int const KU = a.plan.KU;
  if (a.input == IN_RAW_RECORDS) return launch_aggregate_records<IN_RAW_RECORDS>(a, d_args, stream);
  if (a.input == IN_PARTIAL_RECORDS) return launch_aggregate_records<IN_PARTIAL_RECORDS>(a, d_args, stream);
  bool const simple = a.plan.simple;
  if (KU <= 1) simple ? launch_aggregate_t<IN_COLUMNS, 1, 0, true, false>(a, d_args, stream) : launch_aggregate_t<IN_COLUMNS, 1, 0, false, false>(a, d_args, stream);
  else if (KU <= 2) simple ? launch_aggregate_t<IN_COLUMNS, 2, 0, true, false>(a, d_args, stream) : launch_aggregate_t<IN_COLUMNS, 2, 0, false, false>(a, d_args, stream);
  else simple ? launch_aggregate_t<IN_COLUMNS, 4, 0, true, false>(a, d_args, stream) : launch_aggregate_t<IN_COLUMNS, 4, 0, false, false>(a, d_args, stream);
}

void launch_finalize(finalize_args const& a, finalize_args* d_args, uint64_t const* records, int64_t cap,
                     int64_t const* prefix, int32_t nitems, int64_t total, hipStream_t stream)
{
  if (total == 0) return;
  int const block = 256;
  int64_t const grid = (total + block - 1) / block;
  hipLaunchKernelGGL(k_store_args<finalize_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"finalize", stream};
  hipLaunchKernelGGL(k_finalize, dim3(static_cast<unsigned>(grid)), dim3(block), 0, stream, d_args, records, cap, prefix,
                     nitems, total);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_estimate(plan_dev const& plan, plan_dev* d_plan, int64_t nrows, int64_t sample, uint32_t* bitmap,
                     int32_t bitmap_bits_log2, uint32_t* d_bits_set, hipStream_t stream)
{
  int64_t const nwords = (int64_t{1} << bitmap_bits_log2) / 32;
  CUDF_HIP_TRY(hipMemsetAsync(bitmap, 0, nwords * 4, stream));
  CUDF_HIP_TRY(hipMemsetAsync(d_bits_set, 0, 4, stream));
  int const block = 256;
  hipLaunchKernelGGL(k_store_args<plan_dev>, dim3(1), dim3(1), 0, stream, plan, d_plan);
  cudf::detail::prof::scope prof_{"estimate", stream};
  hipLaunchKernelGGL(k_estimate, dim3(static_cast<unsigned>((sample + block - 1) / block)), dim3(block), 0, stream, d_plan,
                     nrows, sample, bitmap, bitmap_bits_log2);
  hipLaunchKernelGGL(k_popcount, dim3(256), dim3(block), 0, stream, bitmap, nwords, d_bits_set);
  CUDF_HIP_TRY(hipGetLastError());
}

}  // namespace cudf::groupby::detail
