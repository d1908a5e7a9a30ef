// SPDX-License-Identifier: Apache-2.0
// Row loader of COMPOSITE dense keys (engine.hpp dense_map, nkeys >= 1), shared by the write-combining scatter
// (partition_kernels.hip) and the ring scatter (dense_ring_kernels.hip): 1-4 integer key columns of any width and one value
// column, all nullable -> per row the mixed-radix index of its key, the validity of its value and the value as its 8-byte
// accumulator class. A row with a NULL key is dropped (null_policy::EXCLUDE, reference groupby/common/utils.cpp:14-32).
#pragma once
#include "device_common.hpp"

namespace cudf::groupby::detail {
namespace {

// The loads of one tile of rows (tile + k * B + threadIdx.x, k < RPT, below `end`), issued by issue_dense_composite and not
// waited for until decode_dense_composite reads them: the first two key columns, their validity words, the value and its
// validity (a third and fourth key column are read in the decode step, one after the other: four columns' worth of rows in
// registers spilled).
// (tried: one wave-wide load of all validity words + cross-lane reads instead of a load per row and mask - 16.0 vs 13.1 ms)
template <int RPT>
struct dense_raw_tile {
  static constexpr int WIDE = 2;
  int64_t tile;
  int64_t end;
  uint64_t raw[WIDE][RPT];
  uint32_t kmw[WIDE][RPT];
  uint64_t vraw[RPT];
  uint32_t vmw[RPT];
};

template <int RPT>
__device__ __forceinline__ void issue_dense_composite(plan_dev const& p, dense_map const& dm, int64_t tile, int B, int64_t end,
                                                      dense_raw_tile<RPT>& t)
{
  constexpr int WIDE = dense_raw_tile<RPT>::WIDE;
  t.tile = tile;
  t.end  = end;
  int64_t row[RPT];
  bool inrange[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    row[k]     = tile + static_cast<int64_t>(k) * B + threadIdx.x;
    inrange[k] = row[k] < end;
  }
  int const nk = dm.nkeys;
#pragma unroll
  for (int c = 0; c < WIDE; ++c) {
    if (c >= nk) break;
    device_column const col = p.cols[dm.key[c].col];
    batch_load_bits<RPT>(col, row, inrange, t.raw[c]);
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      t.kmw[c][k] = 0xffffffffu;
      if (col.mask != nullptr && inrange[k]) t.kmw[c][k] = gload(col.mask + ((static_cast<int64_t>(col.offset) + row[k]) >> 5));
    }
  }
  device_column const vcol = p.cols[dm.value_col];
  batch_load_bits<RPT>(vcol, row, inrange, t.vraw);
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    t.vmw[k] = 0xffffffffu;
    if (vcol.mask != nullptr && inrange[k]) t.vmw[k] = gload(vcol.mask + ((static_cast<int64_t>(vcol.offset) + row[k]) >> 5));
  }
}

// keep[k]: the row exists and no key of it is NULL; idx32: the mixed-radix index of its key; valid / vbits: validity of its value
// and the value as its 8-byte accumulator class; bad: some key of a kept row lay outside its sampled range (the attempt is
// void: overflow bit 2).
template <int RPT>
__device__ __forceinline__ void decode_dense_composite(plan_dev const& p, dense_map const& dm, int B, dense_raw_tile<RPT> const& t,
                                                       bool (&keep)[RPT], uint32_t (&idx32)[RPT], uint32_t (&valid)[RPT],
                                                       uint64_t (&vbits)[RPT], bool& bad)
{
  constexpr int WIDE = dense_raw_tile<RPT>::WIDE;
  int64_t row[RPT];
  bool inrange[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    row[k]     = t.tile + static_cast<int64_t>(k) * B + threadIdx.x;
    keep[k]    = row[k] < t.end;
    inrange[k] = keep[k];
    idx32[k]   = 0;
  }
  int const nk = dm.nkeys;
  bad          = false;
  auto add_digit = [&](dense_key const& dk, int moff, uint64_t rawv, uint32_t mword, int k) {
    if (!((mword >> ((moff + row[k]) & 31)) & 1u)) keep[k] = false;  // NULL key: the row is dropped (EXCLUDE)
    int const sh     = 64 - 8 * dk.width;
    uint64_t const v = dk.is_signed ? static_cast<uint64_t>(static_cast<int64_t>(rawv << sh) >> sh) : rawv;
    uint64_t dig     = v - dk.lo;
    if (dig >= dk.range) {
      bad = bad || keep[k];
      dig = 0;
    }
    idx32[k] += static_cast<uint32_t>(dig) * dk.stride;  // (< 2^30: 32-bit arithmetic)
  };
#pragma unroll
  for (int c = 0; c < WIDE; ++c) {
    if (c >= nk) break;
    dense_key const dk = dm.key[c];
    int const moff     = p.cols[dk.col].offset;
#pragma unroll
    for (int k = 0; k < RPT; ++k)
      if (inrange[k]) add_digit(dk, moff, t.raw[c][k], t.kmw[c][k], k);
  }
  for (int c = WIDE; c < nk; ++c) {
    dense_key const dk      = dm.key[c];
    device_column const col = p.cols[dk.col];
    uint64_t r2[RPT];
    batch_load_bits<RPT>(col, row, inrange, r2);
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      if (!inrange[k]) continue;
      uint32_t const mw = col.mask != nullptr ? gload(col.mask + ((static_cast<int64_t>(col.offset) + row[k]) >> 5)) : 0xffffffffu;
      add_digit(dk, col.offset, r2[k], mw, k);
    }
  }
  device_column const vcol = p.cols[dm.value_col];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    valid[k] = 0;
    vbits[k] = 0;
    if (!inrange[k]) continue;
    valid[k] = (t.vmw[k] >> ((vcol.offset + row[k]) & 31)) & 1u;
    vbits[k] = to_acc_bits(t.vraw[k], vcol.cls, vcol.width);
  }
}

template <int RPT>
__device__ __forceinline__ void load_dense_composite(plan_dev const& p, dense_map const& dm, int64_t tile, int B, int64_t end,
                                                     bool (&keep)[RPT], uint32_t (&idx32)[RPT], uint32_t (&valid)[RPT],
                                                     uint64_t (&vbits)[RPT], bool& bad)
{
  dense_raw_tile<RPT> t;
  issue_dense_composite<RPT>(p, dm, tile, B, end, t);
  decode_dense_composite<RPT>(p, dm, B, t, keep, idx32, valid, vbits, bad);
}

}  // namespace
}  // namespace cudf::groupby::detail
