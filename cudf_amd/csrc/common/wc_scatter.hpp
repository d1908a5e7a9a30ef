// SPDX-License-Identifier: Apache-2.0
// Write-combining radix scatter of fixed-size records (U 8-byte units; 16-byte records have their own 128-bit path)
// into per-(partition, workgroup) regions - shared by the groupby partition passes and the join's probe-side
// partition pass.
//
// A run-per-tile scatter writes, per tile and partition, one run of ~T/P records at an arbitrary 16-byte offset:
// almost every 128-byte line is written in two pieces and relies on the L2 to merge them (with the merge disabled -
// nontemporal stores - the groupby scatter takes 15.9 instead of 11.6 ms; isolated unaligned 128-byte runs write
// at 1.2-1.4 TB/s against 5.8 TB/s aligned, profiles/r1_scatter_align_microbench.txt).
// Here a partition's records leave the workgroup only as whole, aligned GRANULES of G records (64 or 128 bytes):
// the 0..G-1 records that do not fill a granule stay in an LDS carry area and lead the partition's sequence in
// the next tile. Per tile: rank (LDS histogram pre-loaded with the carry counts) -> one packed scan (new records |
// granules) -> stage the new records -> issue the next tile's loads -> write the granules (G lanes per granule,
// source = carry then stage) -> move the unwritten remainder from the stage to the carry area.
// Workgroup `item` appends partition d's records to its own fixed-capacity region [(d*slices + item) * region_cap,
// +region_cap); fill counts go to region_count[d*slices + item]; a region that would overflow raises *overflow
// before anything of that tile is written (the caller redoes the work another way).
// LDS: stage[T] rec | carry[P*(G-1)] rec | meta[P] u64 | delta[P] i64 | hist[P] u32 | gmap[(T+(G-1)P)/G] u16 | sums
#pragma once
#include "device_table.hpp"

#include <cstddef>
#include <cstdint>

namespace cudf::detail {

inline std::size_t wc_scatter_lds_bytes(std::size_t tile_rows, std::size_t P, int G, int U = 2)
{
  std::size_t const gmap_len = (tile_rows + (G - 1) * P) / G + 1;
  return tile_rows * U * 8 + P * (G - 1) * U * 8 + P * (8 + 8 + 4) + ((gmap_len + 3) & ~std::size_t{3}) * 2 + 32 * 8;
}

#if defined(__HIPCC__)
struct wc_scatter_geom {
  int P;                  // partitions (<= 2 * blockDim.x)
  int slices;             // workgroups sharing the output (region index = d * slices + item)
  int item;               // this workgroup
  int64_t begin, end;     // input rows of this workgroup: tiles at begin, begin + step, ... below end
  int64_t step;           // 0 = consecutive tiles (a contiguous slice); else the distance between this workgroup's tiles
  int64_t region_cap;     // records per region (multiple of 8)
  int32_t* region_count;  // [P * slices]
  int32_t* overflow;
  uint64_t* out;          // records of U units
};

// LDS-only barrier (global loads/stores stay in flight across it)
__device__ __forceinline__ void wc_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// load_tile(tile_base, rec, keep): fills rec[k] / keep[k] for row tile_base + k * blockDim.x + threadIdx.x (k < RPT),
// issuing all loads back to back. digit_of(rec[k]) -> partition in [0, P).
// PEEL: the first tile's loads are issued ahead of the loop (two call sites of the loader: fine for a small one);
// otherwise the loop runs one extra leading round that only loads, so that a large loader is instantiated once.
template <int RPT, int G, int U, bool PEEL, class LoadTile, class DigitOf>
__device__ __forceinline__ void wc_scatter_slice(unsigned char* lds_raw, wc_scatter_geom const& g, LoadTile&& load_tile,
                                                 DigitOf&& digit_of)
{
  int const P = g.P, B = blockDim.x, T = B * RPT;
  constexpr int UT = U, CW = G - 1;
  static_assert((G * U) % 2 == 0, "a granule is written in 16-byte chunks");
  uint64_t* stage     = reinterpret_cast<uint64_t*>(lds_raw);           // [T][U]
  uint64_t* carry     = stage + static_cast<size_t>(T) * U;             // [P][G-1][U]
  uint64_t* meta      = carry + static_cast<size_t>(P) * CW * U;
  int64_t* delta      = reinterpret_cast<int64_t*>(meta + P);
  uint32_t* hist      = reinterpret_cast<uint32_t*>(delta + P);
  int const gmap_len  = (T + CW * P) / G + 1;
  uint16_t* gmap      = reinterpret_cast<uint16_t*>(hist + P);
  uint64_t* wave_sums = reinterpret_cast<uint64_t*>(gmap + ((gmap_len + 3) & ~3));
  int const item      = g.item;
  uint64_t* out       = g.out;
  // thread t owns partitions d = t + k*B: output cursor and carry count live in its registers
  constexpr int MAXE = 2;  // P <= 2 * B
  int64_t cursor[MAXE], region_end[MAXE];
  uint32_t ccnt[MAXE];
  __shared__ int s_abort;
  if (threadIdx.x == 0) s_abort = 0;
#pragma unroll
  for (int k = 0; k < MAXE; ++k) {
    int const d   = threadIdx.x + k * B;
    cursor[k]     = (static_cast<int64_t>(d) * g.slices + item) * g.region_cap;
    region_end[k] = cursor[k] + g.region_cap;
    ccnt[k]       = 0;
    if (d < P) hist[d] = 0;
  }
  wc_lds_barrier();

  uint64_t rec[RPT][UT];
  bool keep[RPT];
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = B >> 6;
  // (the generic column reader inlined twice made the kernel 17,000 instructions long)
  if constexpr (PEEL) {
    if (g.begin < g.end) load_tile(g.begin, rec, keep);
  }
  int64_t const step = g.step > 0 ? g.step : T;
  for (int64_t tile = PEEL ? g.begin : g.begin - step; tile < g.end; tile += step) {
    bool const cur = PEEL || tile >= g.begin;  // a tile sits in the registers
    uint32_t tot[MAXE], wr[MAXE], sofs[MAXE], total_gr = 0;
    if (cur) {
    uint32_t dig[RPT], rank[RPT];
    // rank within the partition's sequence: the histogram starts at the carry count
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      if (keep[k]) {
        dig[k]  = digit_of(rec[k]);
        rank[k] = atomicAdd(&hist[dig[k]], 1u);
      }
    }
    wc_lds_barrier();
    // owners: new-record count and granule count per partition; one packed scan gives stage and granule offsets
    uint64_t local = 0;
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      int const d = threadIdx.x + k * B;
      tot[k]      = d < P ? hist[d] : 0;
      wr[k]       = tot[k] & ~static_cast<uint32_t>(G - 1);
      local += static_cast<uint64_t>(tot[k] - ccnt[k]) | (static_cast<uint64_t>(wr[k] / G) << 32);
    }
    unsigned long long inc = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      unsigned long long const t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_sums[wave] = inc;
    wc_lds_barrier();
    if (wave == 0) {
      unsigned long long s = lane < nwaves ? wave_sums[lane] : 0;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        unsigned long long const t = __shfl_up(s, o);
        if (lane >= o) s += t;
      }
      if (lane < nwaves) wave_sums[16 + lane] = s;  // inclusive
    }
    wc_lds_barrier();
    uint64_t run             = (wave == 0 ? 0 : wave_sums[16 + wave - 1]) + inc - local;
    total_gr                 = static_cast<uint32_t>(wave_sums[16 + nwaves - 1] >> 32);
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      int const d = threadIdx.x + k * B;
      if (d < P) {
        uint32_t const s = static_cast<uint32_t>(run), L = static_cast<uint32_t>(run >> 32);
        sofs[k]  = s;
        meta[d]  = static_cast<uint64_t>(s) | (static_cast<uint64_t>(L) << 16) | (static_cast<uint64_t>(ccnt[k]) << 32);
        delta[d] = cursor[k];
        for (uint32_t g = 0; g < wr[k] / G; ++g) gmap[L + g] = static_cast<uint16_t>(d);
        cursor[k] += wr[k];
        if (cursor[k] > region_end[k]) s_abort = 1;  // region too small: nothing of this tile is written
        run += static_cast<uint64_t>(tot[k] - ccnt[k]) | (static_cast<uint64_t>(wr[k] / G) << 32);
        hist[d] = tot[k] - wr[k];  // next tile's ranks start behind the new carry
      }
    }
    wc_lds_barrier();
    if (s_abort) {
      if (threadIdx.x == 0) *g.overflow = 1;
      return;
    }
    // stage the new records in (owner, partition) order
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      if (keep[k]) {
        uint64_t const m   = meta[dig[k]];
        uint32_t const pos = static_cast<uint32_t>(m & 0xffffu) + rank[k] - static_cast<uint32_t>(m >> 32);
        if constexpr (U == 2) {
          reinterpret_cast<u64x2*>(stage)[pos] = u64x2{rec[k][0], rec[k][1]};
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) stage[pos * U + u] = rec[k][u];
        }
      }
    }
    }
    // the registers are free: the next tile's loads fly under the write-out
    if (tile + step < g.end) load_tile(tile + step, rec, keep);
    if (!cur) continue;
    wc_lds_barrier();
    // write-out in 16-byte chunks; sequence index q < carry count comes from the carry area, the rest from the stage
    if constexpr (U == 2) {  // one lane per record, G lanes per granule
      for (uint32_t g = threadIdx.x / G; g < total_gr; g += B / G) {
        int const d        = gmap[g];
        uint64_t const m   = meta[d];
        uint32_t const s   = static_cast<uint32_t>(m & 0xffffu), L = static_cast<uint32_t>(m >> 16) & 0xffffu,
                       c   = static_cast<uint32_t>(m >> 32);
        uint32_t const q   = (g - L) * G + (threadIdx.x % G);
        u64x2 v;
        if (q < c) v = reinterpret_cast<u64x2 const*>(carry)[static_cast<uint32_t>(d) * CW + q];
        else v = reinterpret_cast<u64x2 const*>(stage)[s + q - c];
        gstore(reinterpret_cast<u64x2*>(out) + delta[d] + q, v);
      }
    } else {  // G*U/2 lanes per granule, each moving two consecutive units of the granule's record sequence
      constexpr uint32_t CG = G * U / 2;
      for (uint32_t ch = threadIdx.x; ch < total_gr * CG; ch += B) {
        uint32_t const gi  = ch / CG, ci = ch - gi * CG;
        int const d        = gmap[gi];
        uint64_t const m   = meta[d];
        uint32_t const s   = static_cast<uint32_t>(m & 0xffffu), L = static_cast<uint32_t>(m >> 16) & 0xffffu,
                       c   = static_cast<uint32_t>(m >> 32);
        uint32_t const w0  = (gi - L) * (G * U) + 2 * ci;  // unit index in the partition's sequence of this tile
        uint64_t val[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          uint32_t const w = w0 + j, q = w / U, u = w - q * U;
          val[j] = q < c ? carry[(static_cast<uint32_t>(d) * CW + q) * U + u] : stage[(s + q - c) * U + u];
        }
        gstore(reinterpret_cast<u64x2*>(out + delta[d] * U + w0), u64x2{val[0], val[1]});
      }
    }
    wc_lds_barrier();
    // remainder: sequence [max(written, old carry), total) moves from the stage to the front of the carry area
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      int const d = threadIdx.x + k * B;
      if (d < P) {
        uint32_t const c = ccnt[k];
        for (uint32_t q = wr[k] > c ? wr[k] : c; q < tot[k]; ++q) {
#pragma unroll
          for (int u = 0; u < U; ++u) carry[(static_cast<uint32_t>(d) * CW + q - wr[k]) * U + u] = stage[(sofs[k] + q - c) * U + u];
        }
        ccnt[k] = tot[k] - wr[k];
      }
    }
    // (no barrier: the next tile touches only hist and registers before its first barrier)
  }
  wc_lds_barrier();
  // flush the carried records (the only partial granules of the region): owners publish cursor and carry count, then G
  // lanes per partition write its carried records side by side (16-byte stores; one thread storing a partition's units one
  // by one cost the chunked pipeline, which flushes once per chunk, 4 % of its rows in 8-byte stores)
#pragma unroll
  for (int k = 0; k < MAXE; ++k) {
    int const d = threadIdx.x + k * B;
    if (d < P) {
      if (cursor[k] + ccnt[k] > region_end[k]) {
        *g.overflow = 1;
        ccnt[k]     = 0;
      }
      delta[d] = cursor[k];
      hist[d]  = ccnt[k];
      g.region_count[static_cast<int64_t>(d) * g.slices + item] = static_cast<int32_t>(cursor[k] + ccnt[k] - (region_end[k] - g.region_cap));
    }
  }
  wc_lds_barrier();
  if constexpr (U == 2) {
    for (int idx = threadIdx.x; idx < P * G; idx += B) {
      int const d = idx / G;
      uint32_t const q = static_cast<uint32_t>(idx % G);
      if (q < hist[d]) gstore(reinterpret_cast<u64x2*>(out) + delta[d] + q, reinterpret_cast<u64x2 const*>(carry)[static_cast<uint32_t>(d) * CW + q]);
    }
  } else {
    for (int idx = threadIdx.x; idx < P * G; idx += B) {
      int const d = idx / G;
      uint32_t const q = static_cast<uint32_t>(idx % G);
      if (q < hist[d]) {
#pragma unroll
        for (int u = 0; u < U; ++u) gstore(out + (delta[d] + q) * U + u, carry[(static_cast<uint32_t>(d) * CW + q) * U + u]);
      }
    }
  }
}

#endif  // __HIPCC__

}  // namespace cudf::detail
