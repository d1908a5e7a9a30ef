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
  uint64_t raw[WIDE][RPT];
  uint32_t kmw[WIDE][RPT];
  uint64_t vraw[RPT];
  uint32_t vmw[RPT];
};

// raw element bits of one row of a column (zero-extended); the width test is wave-uniform
__device__ __forceinline__ uint64_t load_bits_row(device_column const& col, int64_t row)
{
  int64_t const at = col.offset + row;
  switch (col.width) {
    case 1: return gload(static_cast<uint8_t const*>(col.head) + at);
    case 2: return gload(static_cast<uint16_t const*>(col.head) + at);
    case 4: return gload(static_cast<uint32_t const*>(col.head) + at);
    default: return gload(static_cast<uint64_t const*>(col.head) + at);
  }
}

// Rows K0 <= k < K1 of the tile only: a caller that decodes one half of a tile and at once issues the same half of the tile after
// next keeps loads in flight through the decode phase without a second set of registers.
template <int RPT, int K0 = 0, int K1 = RPT>
__device__ __forceinline__ void issue_dense_composite(plan_dev const& p, dense_map const& dm, int64_t tile, int B, int64_t end,
                                                      dense_raw_tile<RPT>& t)
{
  constexpr int WIDE = dense_raw_tile<RPT>::WIDE;
  int const nk = dm.nkeys;
#pragma unroll
  for (int c = 0; c < WIDE; ++c) {
    if (c >= nk) break;
    device_column const col = p.cols[dm.key[c].col];
#pragma unroll
    for (int k = K0; k < K1; ++k) {
      int64_t const row = tile + static_cast<int64_t>(k) * B + threadIdx.x;
      t.raw[c][k]       = 0;
      t.kmw[c][k]       = 0xffffffffu;
      if (row < end) {
        t.raw[c][k] = load_bits_row(col, row);
        if (col.mask != nullptr) t.kmw[c][k] = gload(col.mask + ((static_cast<int64_t>(col.offset) + row) >> 5));
      }
    }
  }
  device_column const vcol = p.cols[dm.value_col];
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    int64_t const row = tile + static_cast<int64_t>(k) * B + threadIdx.x;
    t.vraw[k]         = 0;
    t.vmw[k]          = 0xffffffffu;
    if (row < end) {
      t.vraw[k] = load_bits_row(vcol, row);
      if (vcol.mask != nullptr) t.vmw[k] = gload(vcol.mask + ((static_cast<int64_t>(vcol.offset) + row) >> 5));
    }
  }
}

// keep[k]: the row exists and no key of it is NULL; idx32: the mixed-radix index of its key; valid / vbits: validity of its value
// and the value as its 8-byte accumulator class; bad: some key of a kept row lay outside its sampled range (the attempt is
// void: overflow bit 2).
// (bad is OR-ed into: the caller clears it before the first part of a tile)
template <int RPT, int K0 = 0, int K1 = RPT>
__device__ __forceinline__ void decode_dense_composite(plan_dev const& p, dense_map const& dm, int64_t tile, int B, int64_t end,
                                                       dense_raw_tile<RPT> const& t, bool (&keep)[RPT], uint32_t (&idx32)[RPT],
                                                       uint32_t (&valid)[RPT], uint64_t (&vbits)[RPT], bool& bad)
{
  constexpr int WIDE = dense_raw_tile<RPT>::WIDE;
  int64_t row[RPT];
  bool inrange[RPT];
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    row[k]     = tile + static_cast<int64_t>(k) * B + threadIdx.x;
    keep[k]    = row[k] < end;
    inrange[k] = keep[k];
    idx32[k]   = 0;
  }
  int const nk = dm.nkeys;
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
    for (int k = K0; k < K1; ++k)
      if (inrange[k]) add_digit(dk, moff, t.raw[c][k], t.kmw[c][k], k);
  }
  for (int c = WIDE; c < nk; ++c) {
    dense_key const dk      = dm.key[c];
    device_column const col = p.cols[dk.col];
    uint64_t r2[RPT];
#pragma unroll
    for (int k = K0; k < K1; ++k) r2[k] = inrange[k] ? load_bits_row(col, row[k]) : 0;
#pragma unroll
    for (int k = K0; k < K1; ++k) {
      if (!inrange[k]) continue;
      uint32_t const mw = col.mask != nullptr ? gload(col.mask + ((static_cast<int64_t>(col.offset) + row[k]) >> 5)) : 0xffffffffu;
      add_digit(dk, col.offset, r2[k], mw, k);
    }
  }
  device_column const vcol = p.cols[dm.value_col];
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    valid[k] = 0;
    vbits[k] = 0;
    if (!inrange[k]) continue;
    valid[k] = (t.vmw[k] >> ((vcol.offset + row[k]) & 31)) & 1u;
    vbits[k] = to_acc_bits(t.vraw[k], vcol.cls, vcol.width);
  }
}

// ---- the same loader with the descriptors of the first two key columns and of the value column resolved ONCE into locals
// (wave-uniform: scalar registers): read through the plan pointer they were re-fetched after every barrier (the barrier's memory
// clobber), 99 scalar instructions per row against 32 on the plain path. Loads are unconditional - a row past the end reads the
// last row, a column without a validity mask reads word 0 of an all-ones dummy - so that a tile's loads go out back to back.
struct dense_col_local {
  unsigned char const* head;  // element (offset + row) at head + (offset + row) * width
  uint32_t const* mask;       // validity words; the all-ones dummy if the column has none
  int64_t offset;
  int64_t mask_on;            // -1: the column has a mask, 0: it has none (word index & mask_on)
  int32_t width;
};
struct dense_local {
  dense_col_local kc[2], vc;
  uint64_t lo[2];
  uint32_t range[2], stride[2];
  int32_t sh[2];  // 64 - 8 * width
  bool is_signed[2];
  int32_t nk, vcls;
};
__device__ __forceinline__ dense_col_local make_col_local(device_column const& c, uint32_t const* ones)
{
  dense_col_local l;
  l.head    = static_cast<unsigned char const*>(c.head);
  l.mask    = c.mask != nullptr ? c.mask : ones;
  l.offset  = c.offset;
  l.mask_on = c.mask != nullptr ? int64_t{-1} : int64_t{0};
  l.width   = c.width;
  return l;
}
__device__ __forceinline__ dense_local make_dense_local(plan_dev const& p, dense_map const& dm, uint32_t const* ones)
{
  dense_local L;
  L.nk = dm.nkeys;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    dense_key const dk = dm.key[c < dm.nkeys ? c : 0];
    L.kc[c]        = make_col_local(p.cols[dk.col], ones);
    L.lo[c]        = dk.lo;
    L.range[c]     = dk.range;
    L.stride[c]    = c < dm.nkeys ? dk.stride : 0;  // (a missing second key contributes nothing)
    L.sh[c]        = 64 - 8 * dk.width;
    L.is_signed[c] = dk.is_signed != 0;
  }
  L.vc   = make_col_local(p.cols[dm.value_col], ones);
  L.vcls = p.cols[dm.value_col].cls;
  return L;
}
template <int RPT>
__device__ __forceinline__ void load_col_local(dense_col_local const& c, int64_t const (&rowc)[RPT], uint64_t (&bits)[RPT], uint32_t (&mw)[RPT])
{
  switch (c.width) {
    case 1:
#pragma unroll
      for (int k = 0; k < RPT; ++k) bits[k] = gload(c.head + c.offset + rowc[k]);
      break;
    case 2:
#pragma unroll
      for (int k = 0; k < RPT; ++k) bits[k] = gload(reinterpret_cast<uint16_t const*>(c.head) + c.offset + rowc[k]);
      break;
    case 4:
#pragma unroll
      for (int k = 0; k < RPT; ++k) bits[k] = gload(reinterpret_cast<uint32_t const*>(c.head) + c.offset + rowc[k]);
      break;
    default:
#pragma unroll
      for (int k = 0; k < RPT; ++k) bits[k] = gload(reinterpret_cast<uint64_t const*>(c.head) + c.offset + rowc[k]);
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k) mw[k] = gload(c.mask + (((c.offset + rowc[k]) >> 5) & c.mask_on));
}
template <int RPT>
__device__ __forceinline__ void issue_dense_local(dense_local const& L, int64_t tile, int B, int64_t end, dense_raw_tile<RPT>& t)
{
  int64_t rowc[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) rowc[k] = max(min(tile + static_cast<int64_t>(k) * B + threadIdx.x, end - 1), int64_t{0});  // (never row -1; aggregate() returns before any launch when the input has no rows)
  load_col_local<RPT>(L.kc[0], rowc, t.raw[0], t.kmw[0]);
  if (L.nk > 1) load_col_local<RPT>(L.kc[1], rowc, t.raw[1], t.kmw[1]);
  load_col_local<RPT>(L.vc, rowc, t.vraw, t.vmw);
}
template <int RPT>
__device__ __forceinline__ void decode_dense_local(plan_dev const& p, dense_map const& dm, dense_local const& L, int64_t tile, int B,
                                                   int64_t end, dense_raw_tile<RPT> const& t, bool (&keep)[RPT], uint32_t (&idx32)[RPT],
                                                   uint32_t (&valid)[RPT], uint64_t (&vbits)[RPT], bool& bad)
{
  int64_t row[RPT];
  bool inrange[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    row[k]     = tile + static_cast<int64_t>(k) * B + threadIdx.x;
    inrange[k] = row[k] < end;
    keep[k]    = inrange[k];
    idx32[k]   = 0;
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c >= L.nk) break;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      uint32_t const bit = (t.kmw[c][k] >> ((L.kc[c].offset + row[k]) & 31)) & 1u;
      uint64_t const raw = t.raw[c][k];
      uint64_t const v   = L.is_signed[c] ? static_cast<uint64_t>(static_cast<int64_t>(raw << L.sh[c]) >> L.sh[c]) : raw;
      uint64_t const dig = v - L.lo[c];
      bool const out     = dig >= L.range[c];
      keep[k]            = keep[k] && bit != 0;  // NULL key: the row is dropped (EXCLUDE)
      bad                = bad || (out && keep[k]);
      idx32[k] += out ? 0u : static_cast<uint32_t>(dig) * L.stride[c];  // (< 2^30: 32-bit arithmetic)
    }
  }
  if (L.nk > 2) {  // a third and fourth key column: read here, one after the other
    for (int c = 2; c < L.nk; ++c) {
      dense_key const dk      = dm.key[c];
      device_column const col = p.cols[dk.col];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        if (!inrange[k]) continue;
        uint64_t const rawv = load_bits_row(col, row[k]);
        uint32_t const mw   = col.mask != nullptr ? gload(col.mask + ((static_cast<int64_t>(col.offset) + row[k]) >> 5)) : 0xffffffffu;
        if (!((mw >> ((col.offset + row[k]) & 31)) & 1u)) keep[k] = false;
        int const sh     = 64 - 8 * dk.width;
        uint64_t const v = dk.is_signed ? static_cast<uint64_t>(static_cast<int64_t>(rawv << sh) >> sh) : rawv;
        uint64_t dig     = v - dk.lo;
        if (dig >= dk.range) {
          bad = bad || keep[k];
          dig = 0;
        }
        idx32[k] += static_cast<uint32_t>(dig) * dk.stride;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    valid[k] = inrange[k] ? (t.vmw[k] >> ((L.vc.offset + row[k]) & 31)) & 1u : 0u;
    vbits[k] = to_acc_bits(t.vraw[k], L.vcls, L.vc.width);
  }
}

// ---- the same loader with the SHAPE of the columns fixed at compile time: key widths, which columns carry a
// validity mask, an 8-byte value. The run-time form above loads a (dummy) validity word for every column and row, switches on the
// width at every load and does all digit arithmetic in 64 bits; here a column without a mask costs no load, a 4-byte key sits
// in one register and its digit is 32-bit arithmetic. W1 == 0: one key column. Offsets, bases, lo / range / stride and the
// signedness stay run-time (wave-uniform) values of `dense_local`.
template <int W0_, bool M0_, int W1_, bool M1_, bool MV_>
struct dense_shape {
  static constexpr int W0 = W0_, W1 = W1_;
  static constexpr bool M0 = M0_, M1 = M1_, MV = MV_;
};
template <int W>
using key_reg_t = std::conditional_t<W == 8, uint64_t, uint32_t>;
template <typename SHAPE, int RPT>
struct dense_static_tile {
  key_reg_t<SHAPE::W0> k0[RPT];
  key_reg_t<SHAPE::W1 == 0 ? 4 : SHAPE::W1> k1[SHAPE::W1 != 0 ? RPT : 1];
  uint32_t m0[SHAPE::M0 ? RPT : 1], m1[SHAPE::M1 ? RPT : 1], mv[SHAPE::MV ? RPT : 1];
  uint64_t v[RPT];
};
template <int W>
__device__ __forceinline__ key_reg_t<W> load_key_static(dense_col_local const& c, int64_t rowc)
{
  if constexpr (W == 8) return gload(reinterpret_cast<uint64_t const*>(c.head) + c.offset + rowc);
  else if constexpr (W == 4) return gload(reinterpret_cast<uint32_t const*>(c.head) + c.offset + rowc);
  else if constexpr (W == 2) return gload(reinterpret_cast<uint16_t const*>(c.head) + c.offset + rowc);
  else return gload(c.head + c.offset + rowc);
}
template <typename SHAPE, int RPT>
__device__ __forceinline__ void issue_dense_static(dense_local const& L, int64_t tile, int B, int64_t end, dense_static_tile<SHAPE, RPT>& t)
{
  int64_t rowc[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) rowc[k] = max(min(tile + static_cast<int64_t>(k) * B + threadIdx.x, end - 1), int64_t{0});  // (never row -1; aggregate() returns before any launch when the input has no rows)
#pragma unroll
  for (int k = 0; k < RPT; ++k) t.k0[k] = load_key_static<SHAPE::W0>(L.kc[0], rowc[k]);
  if constexpr (SHAPE::W1 != 0) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) t.k1[k] = load_key_static<SHAPE::W1>(L.kc[1], rowc[k]);
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k) t.v[k] = gload(reinterpret_cast<uint64_t const*>(L.vc.head) + L.vc.offset + rowc[k]);
  if constexpr (SHAPE::M0) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) t.m0[k] = gload(L.kc[0].mask + ((L.kc[0].offset + rowc[k]) >> 5));
  }
  if constexpr (SHAPE::M1) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) t.m1[k] = gload(L.kc[1].mask + ((L.kc[1].offset + rowc[k]) >> 5));
  }
  if constexpr (SHAPE::MV) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) t.mv[k] = gload(L.vc.mask + ((L.vc.offset + rowc[k]) >> 5));
  }
}
// one key column's digit: keep &= its validity bit, bad |= kept and out of range, idx32 += digit * stride
template <int W, bool M>
__device__ __forceinline__ void digit_static(dense_local const& L, int c, key_reg_t<W> raw, uint32_t mword, int64_t row, bool& keep, bool& bad,
                                             uint32_t& idx32)
{
  if constexpr (M) keep = keep && ((mword >> ((L.kc[c].offset + row) & 31)) & 1u) != 0;  // NULL key: the row is dropped (EXCLUDE)
  uint64_t v;
  if constexpr (W == 8) {
    v = raw;
  } else {
    // (a narrow column: sign- or zero-extension is one 32-bit operation and a select of the high word)
    constexpr int SH   = 32 - 8 * W;
    uint32_t const r32 = static_cast<uint32_t>(raw);
    int32_t const sx   = static_cast<int32_t>(r32 << SH) >> SH;
    v = L.is_signed[c] ? static_cast<uint64_t>(static_cast<int64_t>(sx)) : static_cast<uint64_t>(r32);
  }
  uint64_t const dig = v - L.lo[c];
  bool const out     = dig >= L.range[c];
  bad                = bad || (out && keep);
  idx32 += out ? 0u : static_cast<uint32_t>(dig) * L.stride[c];
}
template <typename SHAPE, int RPT>
__device__ __forceinline__ void decode_dense_static(dense_local const& L, int64_t tile, int B, int64_t end, dense_static_tile<SHAPE, RPT> const& t,
                                                    bool (&keep)[RPT], uint32_t (&idx32)[RPT], uint32_t (&valid)[RPT], uint64_t (&vbits)[RPT], bool& bad)
{
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    int64_t const row = tile + static_cast<int64_t>(k) * B + threadIdx.x;
    bool const inrange = row < end;
    keep[k]  = inrange;
    idx32[k] = 0;
    digit_static<SHAPE::W0, SHAPE::M0>(L, 0, t.k0[k], SHAPE::M0 ? t.m0[k] : 0u, row, keep[k], bad, idx32[k]);
    if constexpr (SHAPE::W1 != 0) digit_static<SHAPE::W1, SHAPE::M1>(L, 1, t.k1[k], SHAPE::M1 ? t.m1[k] : 0u, row, keep[k], bad, idx32[k]);
    if constexpr (SHAPE::MV) valid[k] = inrange ? (t.mv[k] >> ((L.vc.offset + row) & 31)) & 1u : 0u;
    else valid[k] = inrange ? 1u : 0u;
    vbits[k] = t.v[k];  // (an 8-byte value is its own accumulator class: int64 / uint64 / float64 bits)
  }
}
// does a composite dense plan have this shape? (two key columns at most, an 8-byte value of an 8-byte accumulator class)
template <typename SHAPE>
inline bool dense_shape_matches(plan_dev const& p, dense_map const& dm)
{
  if (dm.nkeys != (SHAPE::W1 != 0 ? 2 : 1)) return false;
  device_column const& c0 = p.cols[dm.key[0].col];
  device_column const& cv = p.cols[dm.value_col];
  bool ok = c0.width == SHAPE::W0 && (c0.mask != nullptr) == SHAPE::M0 && cv.width == 8 && (cv.mask != nullptr) == SHAPE::MV &&
            (cv.cls == cudf::detail::CLS_SINT || cv.cls == cudf::detail::CLS_UINT || cv.cls == cudf::detail::CLS_F64);
  if constexpr (SHAPE::W1 != 0) {
    device_column const& c1 = p.cols[dm.key[1].col];
    ok = ok && c1.width == SHAPE::W1 && (c1.mask != nullptr) == SHAPE::M1;
  }
  return ok;
}

template <int RPT>
__device__ __forceinline__ void load_dense_composite(plan_dev const& p, dense_map const& dm, int64_t tile, int B, int64_t end,
                                                     bool (&keep)[RPT], uint32_t (&idx32)[RPT], uint32_t (&valid)[RPT],
                                                     uint64_t (&vbits)[RPT], bool& bad)
{
  dense_raw_tile<RPT> t;
  issue_dense_composite<RPT>(p, dm, tile, B, end, t);
  bad = false;
  decode_dense_composite<RPT>(p, dm, tile, B, end, t, keep, idx32, valid, vbits, bad);
}

}  // namespace
}  // namespace cudf::groupby::detail
