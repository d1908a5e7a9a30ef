// SPDX-License-Identifier: Apache-2.0
// The core of the LDS RING scatter, shared by its three users: the dense groupby's scatter of {tag, value} records
// (groupby/dense_ring_kernels.hip), its multi-value form (groupby/dense_multi_kernels.hip) and the joins' scatter of {key, row id}
// records (join/radix_kernels.hip). Measurements: profiles/r2_ring_scatter_microbench.txt; prototype bench_micro/ring_scatter_micro.hip.
//
// Every partition owns a ring of CAP record slots in LDS (one ring per output STREAM: a record leaves as two or more streams of
// 2-, 4- or 8-byte elements so that every stream is written in whole aligned 128-byte granules). A row reserves the next virtual
// position of its partition with ONE returning LDS atomic (`tail`) and writes its record there if the position lies below the
// ring's `limit` (= flushed head + CAP); after a barrier the owner lanes flush every COMPLETE granule to the partition's region
// (global element index = region base + virtual position: nothing ever moves inside LDS); a second barrier ends the tile. A row that
// found its ring full waits one flush round. The kernels keep what differs between them - how a tile's rows are loaded and decoded
// into (partition digit, record words), how many streams there are and how their rings are laid out - and call:
//   plan_flush     the owner lanes' element counts of this flush (and the region-overflow test)
//   flush_stream   one stream's complete granules, rings -> regions
//   commit_flush   the owner lanes' heads and the rings' limits after a flush
//   place_tile     reserve / put / flush / waiting rounds of one tile
// Partitions are dealt to waves: wave w owns partitions [w * PW, (w + 1) * PW), lane l < PW of it is the OWNER lane of partition
// w * PW + l and keeps that partition's flushed heads in registers.
#pragma once
#include "device_table.hpp"

namespace cudf::detail::ring {

// LDS-only barrier: global loads / stores stay in flight across it
__device__ __forceinline__ void barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct flush_plan {
  uint32_t nrec;   // elements of the primary streams (granules of G) this flush moves for the owner lane's partition
  uint32_t nrect;  // elements of the secondary stream (granules of GT; its ring is longer, so its elements may wait for their granule)
  int ab;          // the partition's region would overflow: the workgroup gives up
};

// G / GT: elements per 128-byte granule of the primary / secondary streams. `final`: the partial last granules go as well.
template <uint32_t G, uint32_t GT, typename AbortFlag>
__device__ __forceinline__ flush_plan plan_flush(uint32_t const* tail, bool owner, int dmine, uint32_t head, uint32_t headt, uint32_t CAP,
                                                 uint32_t region_cap, bool final, AbortFlag& s_abort)
{
  flush_plan f{0, 0, 0};
  if (owner) {
    uint32_t const t = tail[dmine], lim = head + CAP;
    uint32_t const c = static_cast<int32_t>(t - lim) < 0 ? t : lim;  // records that made it into the ring
    uint32_t const complete = final ? c : (c & ~(G - 1u));
    f.nrec  = complete - head;
    f.nrect = (final ? c : (c & ~(GT - 1u))) - headt;
    if (complete > region_cap) {
      s_abort = 1;
      f.ab    = 1;
    }
  }
  return f;
}

// One stream of ELEM-sized elements: 8 lanes per granule (16 bytes per lane), 8 partitions per batch of the wave.
// count / from: the owner lane's number of new elements and its flushed head of THIS stream; slot_of(d, pos) = ring index of element
// `pos` of partition d (pos is a multiple of 16 bytes' worth of elements: a lane's 16 bytes never wrap); rbase_of(d) = first element
// of partition d's region in `out`.
template <typename ELEM, typename SlotOf, typename RBaseOf>
__device__ __forceinline__ void flush_stream(ELEM const* ring, ELEM* out, int wave, int PW, int lane, uint32_t count, uint32_t from, int ab,
                                             SlotOf slot_of, RBaseOf rbase_of)
{
  constexpr uint32_t PER_LANE = 16 / sizeof(ELEM), GRANULE = 8 * PER_LANE;
  for (int b = 0; b * 8 < PW; ++b) {
    int const pl = b * 8 + (lane >> 3), sub = lane & 7;
    uint32_t const mr = __shfl(count, pl), mh = __shfl(from, pl);
    int const mab     = __shfl(ab, pl);
    int const d       = wave * PW + pl;
    int64_t const rbase = rbase_of(d);
    for (uint32_t g = 0;; ++g) {
      uint32_t const q = g * GRANULE + sub * PER_LANE;
      bool const act   = pl < PW && q < mr && !mab;
      if (__ballot(act) == 0) break;
      if (act) {
        uint32_t const pos = mh + q;
        ELEM const* src    = ring + slot_of(static_cast<uint32_t>(d), pos);
        if (q + PER_LANE <= mr) {
          gstore(reinterpret_cast<u32x4*>(out + rbase + pos), *reinterpret_cast<u32x4 const*>(src));
        } else {  // (the partial tail of the final flush)
          for (uint32_t e = 0; q + e < mr; ++e) gstore(out + rbase + pos + e, src[e]);
        }
      }
    }
  }
}

__device__ __forceinline__ void commit_flush(uint32_t* limit, bool owner, int dmine, uint32_t& head, uint32_t& headt, flush_plan const& f,
                                             uint32_t CAP)
{
  if (owner) {
    head += f.nrec;
    headt += f.nrect;
    limit[dmine] = head + CAP;
  }
}

// One tile: every kept row k reserves a position in ring d[k]; put(k, pos) writes its record; flush() = the kernel's flush(false).
// after_tile(): once after the tile's first flush round (every thread); wait_round(): thread 0, once per waiting round, between
// two barriers (the dense join counts its rounds there). Returns when no row waits any more or the workgroup has given up.
template <int RPT, typename PendFlag, typename AbortFlag, typename Put, typename Flush, typename AfterTile, typename WaitRound>
__device__ __forceinline__ void place_tile(uint32_t* tail, uint32_t const* limit, bool const (&keep)[RPT], uint32_t const (&d)[RPT],
                                           PendFlag& s_pending, AbortFlag const& s_abort, Put put, Flush flush, AfterTile after_tile,
                                           WaitRound wait_round)
{
  uint32_t pos[RPT], lim[RPT];
  bool pend[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    pos[k] = 0;
    lim[k] = 0;
    if (keep[k]) {
      pos[k] = atomicAdd(&tail[d[k]], 1u);
      lim[k] = limit[d[k]];
    }
  }
  bool any_pend = false;
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    pend[k] = keep[k] && static_cast<int32_t>(pos[k] - lim[k]) >= 0;
    if (keep[k] && !pend[k]) put(k, pos[k]);
    any_pend = any_pend || pend[k];
  }
  if (any_pend) s_pending = 1;
  barrier();
  flush();
  barrier();
  after_tile();
  while (s_pending) {  // a ring was full: its granules are flushed by now, the waiting rows go in
    barrier();
    if (threadIdx.x == 0) {
      s_pending = 0;
      wait_round();
    }
    barrier();
    any_pend = false;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      if (pend[k]) {
        if (static_cast<int32_t>(pos[k] - limit[d[k]]) < 0) {
          put(k, pos[k]);
          pend[k] = false;
        } else {
          any_pend = true;
        }
      }
    }
    if (any_pend) s_pending = 1;
    barrier();
    flush();
    barrier();
    if (s_abort) break;
  }
}

}  // namespace cudf::detail::ring
