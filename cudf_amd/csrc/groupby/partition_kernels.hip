// SPDX-License-Identifier: Apache-2.0
// gfx950 partition kernels of the hash-groupby engine: histogram, scan, LDS-staged multi-split scatter (engine.hpp).
#include "device_common.hpp"
#include "dense_loader.hpp"
#include "../common/wc_scatter.hpp"

namespace cudf::groupby::detail {
namespace {
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
#ifdef SCATTER_NT_LOAD
#define SCATTER_LOAD gload_stream
#else
#define SCATTER_LOAD gload
#endif
#ifdef SCATTER_NT_STORE
#define SCATTER_STORE gstore_stream
#else
#define SCATTER_STORE gstore
#endif
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
  __shared__ int32_t s_pre[MAX_REGION_LIST + 1];
  region_input rin{};
  int const from_regions = a.from_regions;
  slice_range sr;
  if (from_regions) {
    rin.build(a, item, s_pre);
    sr.seg_begin = 0;
    sr.begin     = 0;
    sr.end       = rin.total();
  } else {
    sr = slice_of(a, item);
  }
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
      // level 1: region (d, item); level 2 (from_regions): region (g*P + d, s) with item = (g, s)
      int64_t const region = from_regions ? (static_cast<int64_t>(item / a.geom.slices) * P + d) * a.geom.slices + item % a.geom.slices
                                          : static_cast<int64_t>(d) * a.geom.slices + item;
      cursor[k]     = region * a.region_cap;
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
          for (int u = 0; u < UT; ++u) rec[k][u] = (u < U) ? SCATTER_LOAD(sbase[u] + r) : 0;
        } else if constexpr (EXACT && UT == 2) {
          int64_t const ri = from_regions ? rin.record_of(r) : r;
          u64x2 const v = SCATTER_LOAD(reinterpret_cast<u64x2 const*>(in_records) + ri);
          rec[k][0]     = v.x;
          rec[k][1]     = v.y;
        } else {
          int64_t const ri = from_regions ? rin.record_of(r) : r;
#pragma unroll
          for (int u = 0; u < UT; ++u) rec[k][u] = (u < U) ? gload(in_records + ri * U + u) : 0;
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
        SCATTER_STORE(reinterpret_cast<u64x2*>(out_records) + dst, reinterpret_cast<u64x2 const*>(stage)[j]);
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
      if (d < P) a.region_count[(region_end[k] - a.region_cap) / a.region_cap] = static_cast<int32_t>(cursor[k] - (region_end[k] - a.region_cap));
    }
  }
}


// ------------------------------------------------------------------ K_scatter, write-combining form
// Optimistic regions, records of exactly UT units: common/wc_scatter.hpp does the work; this kernel supplies the row
// loader (plain columns, generic columns, records, or a strided list of level-1 regions) and the partition digit.
// SRC: where the rows come from - one instantiation each, the generic column reader and the region reader together
// did not fit 128 VGPRs (39-54 spilled registers).
constexpr int WC_SRC_SIMPLE = 0;   // plain 8-byte columns
constexpr int WC_SRC_COLUMNS = 1;  // any fixed-width columns, nulls (records built column-at-a-time)
constexpr int WC_SRC_RECORDS = 2;  // records: one contiguous range, or a strided list of level-1 regions
constexpr int WC_SRC_DENSE_COLS = 3;  // composite dense keys: 1-4 integer key columns + one value column -> {index | validity, value}
// HOT: heavy-hitter keys (part_args::hot_*) are aggregated in a small LDS table and never scattered.
// DENSE: the partition digit comes from the dense key map (part_args::dense) instead of the key hash.
// `chunk`: rows [begin, end) of the input columns (end == 0: all rows) - the chunked pipeline partitions one chunk per launch.
template <int UT, int RPT, int G, int SRC, bool HOT = false, bool DENSE = false>
__global__ void __launch_bounds__(1024) k_partition_scatter_wc(part_args const* __restrict__ ap, chunk_range chunk)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  part_args const& a = *ap;
  plan_dev const& p  = a.plan;
  int const shift = a.geom.shift, B = blockDim.x;
  static_assert(!HOT || (UT == 2 && SRC == WC_SRC_SIMPLE), "heavy hitters: plain 16-byte records");
  static_assert(SRC != WC_SRC_DENSE_COLS || (UT == 2 && DENSE), "composite dense keys: 16-byte records");
  constexpr uint64_t HOT_EMPTY = ~uint64_t{0};
  uint64_t* hkeys = reinterpret_cast<uint64_t*>(lds_raw + (HOT ? a.hot_lds_offset : 0));  // [HOT_SLOTS]
  uint64_t* hsum  = hkeys + HOT_SLOTS;
  uint32_t* hcnt  = reinterpret_cast<uint32_t*>(hsum + HOT_SLOTS);
  __shared__ uint32_t s_hot_dumped;
  bool hot_float = false;
  if constexpr (HOT) {
    for (int q = 0; q < p.NACC; ++q) hot_float = hot_float || p.acc[q].op == ADD_F64;
    for (int t = threadIdx.x; t < HOT_SLOTS; t += B) {
      hkeys[t] = HOT_EMPTY;
      hsum[t]  = 0;
      hcnt[t]  = 0;
    }
    if (threadIdx.x == 0) s_hot_dumped = 0;
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < a.hot_n) {
      uint64_t const key = a.hot_keys[threadIdx.x];
      uint32_t slot      = static_cast<uint32_t>(mix64(0x9e3779b97f4a7c15ull ^ key) >> 20) & (HOT_SLOTS - 1);
      while (atomicCAS(reinterpret_cast<unsigned long long*>(hkeys + slot), HOT_EMPTY, key) != HOT_EMPTY) slot = (slot + 1) & (HOT_SLOTS - 1);
    }
    __syncthreads();
  }
  __shared__ int32_t s_pre[MAX_REGION_LIST + 1];
  constexpr bool SIMPLE = SRC == WC_SRC_SIMPLE;
  region_input rin{};
  int const from_regions = SRC == WC_SRC_RECORDS ? a.from_regions : 0;
  slice_range sr;
  if (from_regions) {
    rin.build(a, blockIdx.x, s_pre);
    sr.seg_begin = 0;
    sr.begin     = 0;
    sr.end       = rin.total();
  } else {
    sr = slice_of(a, blockIdx.x);
    // Tile-cyclic rows (optimistic partition of the input columns): workgroup w takes tiles w, w + slices, ... so that
    // every workgroup sees the whole row range. With a contiguous chunk per workgroup, sorted or clustered keys put a
    // chunk's rows into a few partitions and overflowed their regions.
    if (a.cyclic_tiles) {
      int64_t const cb = chunk.end > 0 ? chunk.begin : 0, ce = chunk.end > 0 ? chunk.end : a.nrows;
      sr.begin = min(ce, cb + static_cast<int64_t>(blockIdx.x) * (RPT * B));
      sr.end   = ce;
    }
  }
  uint64_t const* in_records = a.in_records;
  constexpr int KUM = UT < MAX_KU ? UT : MAX_KU;
  int const KU = p.KU;
  uint64_t kmask[KUM];
#pragma unroll
  for (int u = 0; u < KUM; ++u) kmask[u] = u < KU ? p.key_mask[u] : 0;
  uint32_t const pmask  = static_cast<uint32_t>(a.geom.P - 1);
  uint64_t const* sbase[UT];
  if constexpr (SIMPLE) {
#pragma unroll
    for (int u = 0; u < UT; ++u) sbase[u] = p.simple_base[u];
  }
  auto load_tile = [&](int64_t tile, uint64_t (&rec)[RPT][UT], bool (&keep)[RPT]) {
    if constexpr (SRC == WC_SRC_DENSE_COLS) {
      // composite dense keys: every key column contributes a digit (value - lo) to the row's mixed-radix index; a row with a
      // NULL key is dropped (null_policy::EXCLUDE); a value outside its sampled range voids the attempt (overflow bit 2)
      uint32_t idx32[RPT], valid[RPT];
      uint64_t vbits[RPT];
      bool bad;
      load_dense_composite<RPT>(p, a.dense, tile, B, sr.end, keep, idx32, valid, vbits, bad);
      if (bad) atomicOr(a.overflow, 4);
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        rec[k][0] = static_cast<uint64_t>(idx32[k]) | (static_cast<uint64_t>(valid[k]) << 32);
        rec[k][1] = vbits[k];
      }
    } else if constexpr (SRC == WC_SRC_COLUMNS) {
      int64_t row[RPT];
      uint32_t vv[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        row[k]  = tile + static_cast<int64_t>(k) * B + threadIdx.x;
        keep[k] = row[k] < sr.end;
      }
      // (descriptors from the plan: resolving them once into registers - units_local - measured 7 % SLOWER here)
      batch_units<RPT, UT>(p, UT, row, keep, rec, vv);
    } else {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      int64_t const r = tile + static_cast<int64_t>(k) * B + threadIdx.x;
      keep[k]         = r < sr.end;
      if (keep[k]) {
        if constexpr (SIMPLE) {
#pragma unroll
          for (int u = 0; u < UT; ++u) rec[k][u] = gload(sbase[u] + r);
          if constexpr (HOT) {  // a heavy hitter is accumulated here and leaves the scatter
            uint64_t const key = rec[k][0] & kmask[0];
            uint32_t slot      = static_cast<uint32_t>(mix64(0x9e3779b97f4a7c15ull ^ key) >> 20) & (HOT_SLOTS - 1);
            for (;;) {
              uint64_t const kk = hkeys[slot];
              if (kk == key) {
                if (hot_float) atomicAdd(reinterpret_cast<double*>(hsum + slot), __longlong_as_double(static_cast<long long>(rec[k][1])));
                else atomicAdd(reinterpret_cast<unsigned long long*>(hsum + slot), static_cast<unsigned long long>(rec[k][1]));
                atomicAdd(hcnt + slot, 1u);
                keep[k] = false;
                break;
              }
              if (kk == HOT_EMPTY) break;
              slot = (slot + 1) & (HOT_SLOTS - 1);
            }
          }
        } else {
          int64_t const ri = from_regions ? rin.record_of(r) : r;
          if constexpr (UT == 2) {
            u64x2 const v = gload(reinterpret_cast<u64x2 const*>(in_records) + ri);
            rec[k][0]     = v.x;
            rec[k][1]     = v.y;
          } else {
#pragma unroll
            for (int u = 0; u < UT; ++u) rec[k][u] = gload(in_records + ri * UT + u);
          }
        }
      }
    }
    }
  };
  [[maybe_unused]] uint64_t const dense_lo = a.dense.lo, dense_range = a.dense.range;
  [[maybe_unused]] uint32_t const dense_mult = a.dense.mult, dense_mask = (1u << a.dense.bits) - 1u;
  // dense digit: bits [shift, shift + log2 P) of the scrambled index (level 1: its top bits; level 2: the next ones)
  [[maybe_unused]] int const dense_shift = shift;
  [[maybe_unused]] bool const dense_rec32 = a.dense.nkeys > 0 || SRC == WC_SRC_RECORDS;  // the record holds the index itself
  auto digit_of = [&](uint64_t const (&rec)[UT]) {
    if constexpr (DENSE) {
      uint64_t idx;
      if (dense_rec32) {
        idx = rec[0] & 0xffffffffull;
      } else {
        idx = rec[0] - dense_lo;
        if (idx >= dense_range) {  // the sampled key range was wrong: this call is void (redone by hash)
          atomicOr(a.overflow, 4);
          idx = 0;
        }
      }
      return (((static_cast<uint32_t>(idx) * dense_mult) & dense_mask) >> dense_shift) & pmask;
    }
    uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
    for (int u = 0; u < KUM; ++u)
      if (u < KU) h = mix64(h ^ (rec[u] & kmask[u]));
    return static_cast<uint32_t>(h >> shift) & pmask;
  };
  cudf::detail::wc_scatter_geom g;
  // level 2 (from_regions): item = (seg, s) writes the regions of the global partitions seg*P + d
  int const seg           = from_regions ? blockIdx.x / a.geom.slices : 0;
  int64_t const region0   = static_cast<int64_t>(seg) * a.geom.P * a.geom.slices;
  g.P            = a.geom.P;
  g.slices       = a.geom.slices;
  g.item         = from_regions ? blockIdx.x % a.geom.slices : blockIdx.x;
  g.begin        = sr.begin;
  g.end          = sr.end;
  g.step         = (!from_regions && a.cyclic_tiles) ? static_cast<int64_t>(a.geom.slices) * (RPT * B) : 0;
  g.region_cap   = a.region_cap;
  g.region_count = a.region_count + region0;
  g.overflow     = a.overflow;
  g.out          = a.out_records + region0 * a.region_cap * UT;
  cudf::detail::wc_scatter_slice<RPT, G, UT, SRC != WC_SRC_COLUMNS>(lds_raw, g, load_tile, digit_of);
  if constexpr (HOT) {  // this workgroup's heavy-hitter partials: [key | accumulators in plan order]
    __syncthreads();
    int const PU   = p.KU + p.NACC;
    uint64_t* out  = a.hot_out + static_cast<int64_t>(blockIdx.x) * HOT_SLOTS * PU;
    for (int t = threadIdx.x; t < HOT_SLOTS; t += B) {
      if (hcnt[t] == 0) continue;
      uint32_t const pos = atomicAdd(&s_hot_dumped, 1u);
      gstore(out + static_cast<int64_t>(pos) * PU, hkeys[t]);
      for (int q = 0; q < p.NACC; ++q)
        gstore(out + static_cast<int64_t>(pos) * PU + 1 + q, p.acc[q].src == SRC_VALUE ? hsum[t] : static_cast<uint64_t>(hcnt[t]));
    }
    __syncthreads();
    if (threadIdx.x == 0) a.hot_count[blockIdx.x] = static_cast<int32_t>(s_hot_dumped);
  }
}

}  // namespace

// ------------------------------------------------------------------ launchers
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
  static std::once_flag attr_once;  // (the API is re-entrant across objects: two threads may launch this kernel first)
  std::call_once(attr_once, [] { allow_full_lds(reinterpret_cast<void const*>(&k_partition_scatter<UT, RPT, SIMPLE, EXACT>)); });
  int const items = g.nseg * g.slices;
  cudf::detail::prof::scope prof_{"partition_scatter", stream};
  hipLaunchKernelGGL((k_partition_scatter<UT, RPT, SIMPLE, EXACT>), dim3(items), dim3(g.block), lds, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}


// Tile rows per thread and granule of the write-combining kernel for a record width; 0 = no instantiation.
int wc_rpt(int U) { return U == 2 ? 5 : U == 3 ? 4 : U == 4 ? 3 : 0; }
bool partition_wc_fits(int U, int P, int G, int block)
{
  int const rpt = wc_rpt(U);
  // 24-byte records: 192-byte granules (G = 8), or 96-byte ones (G = 4: whole 32-byte sectors) where the carry area of
  // G = 8 does not fit; 32-byte records: 128-byte granules
  if (rpt == 0 || (G != 4 && G != 8) || (U == 4 && G != 4)) return false;
  // static LDS of the kernel (region prefix list, abort flag) is ~1.1 KB
  return P <= 2 * block && cudf::detail::wc_scatter_lds_bytes(static_cast<std::size_t>(block) * rpt, P, G, U) + 1200 <= (block == 1024 ? 160 : 80) * 1024;
}

std::size_t partition_hot_lds_bytes() { return static_cast<std::size_t>(HOT_SLOTS) * (8 + 8 + 4); }

template <int UT, int RPT, int G, int SRC, bool HOT = false, bool DENSE = false>
static void launch_scatter_wc_src(part_args const& a, part_args const* d_args, hipStream_t stream, chunk_range chunk)
{
  part_geom g  = a.geom;
  g.tile_rows  = g.block * RPT;
  auto const lds = cudf::detail::wc_scatter_lds_bytes(g.tile_rows, g.P, G, UT) + (HOT ? partition_hot_lds_bytes() : 0);
  CUDF_EXPECTS(lds + 1200 <= 160 * 1024, "write-combining partition kernel: LDS budget exceeded");
  CUDF_EXPECTS(!HOT || g.block == 1024, "heavy hitters: 1024-thread scatter workgroups");
  static std::once_flag attr_once;  // (the API is re-entrant across objects: two threads may launch this kernel first)
  std::call_once(attr_once, [] { allow_full_lds(reinterpret_cast<void const*>(&k_partition_scatter_wc<UT, RPT, G, SRC, HOT, DENSE>)); });
  cudf::detail::prof::scope prof_{a.from_columns ? "partition_scatter" : "partition_scatter_level2", stream};
  hipLaunchKernelGGL((k_partition_scatter_wc<UT, RPT, G, SRC, HOT, DENSE>), dim3(g.nseg * g.slices), dim3(g.block), lds, stream, d_args, chunk);
  CUDF_HIP_TRY(hipGetLastError());
}

template <int UT, int RPT, int G>
static void launch_scatter_wc_t(part_args const& a, part_args const* d_args, hipStream_t stream, chunk_range chunk)
{
  if constexpr (UT == 2) {
    if (!a.from_columns && a.use_dense) return launch_scatter_wc_src<UT, RPT, G, WC_SRC_RECORDS, false, true>(a, d_args, stream, chunk);
    // (4 rows per thread: with 5 the column loader spilled 21 registers to scratch)
    if (a.from_columns && a.use_dense && a.dense.nkeys > 0) return launch_scatter_wc_src<UT, 4, G, WC_SRC_DENSE_COLS, false, true>(a, d_args, stream, chunk);
  }
  if (!a.from_columns) return launch_scatter_wc_src<UT, RPT, G, WC_SRC_RECORDS>(a, d_args, stream, chunk);
  if constexpr (UT == 2) {
    if (a.plan.simple && a.use_dense) return launch_scatter_wc_src<UT, RPT, G, WC_SRC_SIMPLE, false, true>(a, d_args, stream, chunk);
    if (a.plan.simple && a.hot_n > 0) return launch_scatter_wc_src<UT, RPT, G, WC_SRC_SIMPLE, true>(a, d_args, stream, chunk);
  }
  if (a.plan.simple) return launch_scatter_wc_src<UT, RPT, G, WC_SRC_SIMPLE>(a, d_args, stream, chunk);
  return launch_scatter_wc_src<UT, RPT, G, WC_SRC_COLUMNS>(a, d_args, stream, chunk);
}

void launch_partition_scatter(part_args const& a, part_args const* d_args, hipStream_t stream, chunk_range chunk)
{
  CUDF_EXPECTS(chunk.end == 0 || (a.wc_granule != 0 && a.from_columns && a.cyclic_tiles), "chunked scatter: write-combining kernel over the input columns only");
  CUDF_EXPECTS(!a.use_dense || (a.wc_granule != 0 && (a.dense.nkeys > 0 || !a.from_columns || (a.plan.simple && a.plan.KU + a.plan.NPAY == 2))),
               "dense keys: 16-byte records only");
  CUDF_EXPECTS(a.geom.P <= 2 * a.geom.block, "partition fan-out exceeds 2x the block size");
  // (composite dense keys and every later level of a dense partition move 16-byte {index, value} records whatever the plan's units)
  int const U       = (a.use_dense && (a.dense.nkeys > 0 || !a.from_columns)) ? 2 : a.plan.KU + a.plan.NPAY;
  bool const simple = a.plan.simple && a.from_columns;
  if (a.wc_granule != 0) {
    CUDF_EXPECTS(a.optimistic && (a.geom.block == 1024 || a.geom.block == 512) && partition_wc_fits(U, a.geom.P, a.wc_granule, a.geom.block),
                 "write-combining scatter: optimistic regions, records of 2-4 units, carry area within the LDS");
    if (U == 2 && a.wc_granule == 4) launch_scatter_wc_t<2, 5, 4>(a, d_args, stream, chunk);
    else if (U == 2) launch_scatter_wc_t<2, 5, 8>(a, d_args, stream, chunk);
    else if (U == 3 && a.wc_granule == 4) launch_scatter_wc_t<3, 4, 4>(a, d_args, stream, chunk);
    else if (U == 3) launch_scatter_wc_t<3, 4, 8>(a, d_args, stream, chunk);
    else launch_scatter_wc_t<4, 3, 4>(a, d_args, stream, chunk);
    return;
  }
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
}  // namespace cudf::groupby::detail
