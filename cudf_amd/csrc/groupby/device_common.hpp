// SPDX-License-Identifier: Apache-2.0
// Device helpers shared by the partition and aggregate translation units of the hash-groupby engine
// (record building from columns, key hashing, block scan, LDS-only barrier, launch helpers).
#pragma once
#include "engine.hpp"
#include "../common/profiler.hpp"

#include <cudf/utilities/error.hpp>

#include <mutex>

namespace cudf::groupby::detail {

using cudf::detail::col_is_valid;
using cudf::detail::col_load_acc_bits;
using cudf::detail::col_load_bits;
using cudf::detail::gload;
using cudf::detail::gload_stream;
using cudf::detail::gstore_stream;
using cudf::detail::gstore;
using cudf::detail::mix64;
using cudf::detail::normalize_key_bits;
using cudf::detail::u64x2;
using cudf::detail::u32x4;

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

// ------------------------------------------------------------------ accumulator primitives (LDS)
__device__ __forceinline__ uint64_t acc_identity(int op)
{
  switch (op) {
    case MIN_I64: return static_cast<uint64_t>(INT64_MAX);
    case MIN_U64: return UINT64_MAX;
    case MIN_F64: return 0x7ff0000000000000ull;  // +inf
    case MAX_I64: return static_cast<uint64_t>(INT64_MIN);
    case MAX_U64: return 0;
    case MAX_F64: return 0xfff0000000000000ull;  // -inf
    case MUL_I64: return 1;
    case MUL_F64: return 0x3ff0000000000000ull;  // 1.0
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
    case ANY_U64: *slot = v; break;  // any contributing row will do: a plain store
    case MUL_I64:
    case MUL_F64: {
      // no multiply atomic: compare-and-swap loop (ds_cmpst_rtn_b64), as the reference's product (device_atomics.cuh:226-337)
      unsigned long long* p = reinterpret_cast<unsigned long long*>(slot);
      unsigned long long old = *p, seen;
      do {
        seen = old;
        unsigned long long const next =
          op == MUL_I64 ? seen * static_cast<unsigned long long>(v)
                        : static_cast<unsigned long long>(__double_as_longlong(__longlong_as_double(static_cast<long long>(seen)) *
                                                                                __longlong_as_double(static_cast<long long>(v))));
        old = atomicCAS(p, seen, next);
      } while (old != seen);
      break;
    }
  }
}

// what a raw row contributes to an accumulator, from the row's value (acc_src; SRC_VALUE passes through)
__device__ __forceinline__ uint64_t acc_contribution(int src, int op, uint64_t v)
{
  if (src == SRC_SQUARE) {
    if (op == ADD_F64) {
      double const x = __longlong_as_double(static_cast<long long>(v));
      return static_cast<uint64_t>(__double_as_longlong(x * x));
    }
    return v * v;
  }
  if (src == SRC_LO32) return v & 0xffffffffull;
  if (src == SRC_HI32) return static_cast<uint64_t>(static_cast<int64_t>(v) >> 32);
  return v;
}

// a (op) b on accumulator bit patterns - the wave-level counterpart of lds_merge (float min / max: a NaN never beats a
// number, as ds_min_f64 / ds_max_f64)
__device__ __forceinline__ uint64_t combine_values(int op, uint64_t a, uint64_t b)
{
  auto f = [](uint64_t x) { return __longlong_as_double(static_cast<long long>(x)); };
  auto u = [](double x) { return static_cast<uint64_t>(__double_as_longlong(x)); };
  switch (op) {
    case ADD_I64: return a + b;
    case ADD_F64: return u(f(a) + f(b));
    case MIN_I64: return static_cast<uint64_t>(min(static_cast<long long>(a), static_cast<long long>(b)));
    case MIN_U64: return min(a, b);
    case MAX_I64: return static_cast<uint64_t>(max(static_cast<long long>(a), static_cast<long long>(b)));
    case MAX_U64: return max(a, b);
    case MIN_F64: return u(fmin(f(a), f(b)));
    case MAX_F64: return u(fmax(f(a), f(b)));
    case MUL_I64: return a * b;
    case MUL_F64: return u(f(a) * f(b));
    default: return a;  // ANY_U64
  }
}

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

// Signature of a plan's accumulators (0 if it does not fit the static encoding).
inline uint64_t plan_sig(plan_dev const& p)
{
  if (p.NACC < 1 || p.NACC > 4) return 0;
  uint64_t a[4] = {0, 0, 0, 0};
  for (int q = 0; q < p.NACC; ++q) {
    auto const& d = p.acc[q];
    if (d.pay > 5 || d.valid_bit > 5 || d.src > 3) return 0;
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
// SUM + COUNT_VALID (and MEAN) of a NULLABLE column: both accumulators test the row's validity bit
constexpr uint64_t SIG_SUMF_CNT_NULLS = make_sig(2, sig_acc(ADD_F64, SRC_VALUE, 0, 0), sig_acc(ADD_I64, SRC_ONE_IF_VALID, 0, 0));
constexpr uint64_t SIG_SUMI_CNT_NULLS = make_sig(2, sig_acc(ADD_I64, SRC_VALUE, 0, 0), sig_acc(ADD_I64, SRC_ONE_IF_VALID, 0, 0));
// C4: MEAN + MIN + MAX of a nullable float64 column -> SUM, COUNT_VALID, MIN, MAX
constexpr uint64_t SIG_MEAN_MIN_MAX_F_NULLS =
  make_sig(4, sig_acc(ADD_F64, SRC_VALUE, 0, 0), sig_acc(ADD_I64, SRC_ONE_IF_VALID, 0, 0), sig_acc(MIN_F64, SRC_VALUE, 0, 0),
           sig_acc(MAX_F64, SRC_VALUE, 0, 0));

// the same without nulls: SUM, COUNT_ALL (as the count of MEAN), MIN, MAX
constexpr uint64_t SIG_MEAN_MIN_MAX_F =
  make_sig(4, sig_acc(ADD_F64, SRC_VALUE, 0, -1), sig_acc(ADD_I64, SRC_ONE, -1, -1), sig_acc(MIN_F64, SRC_VALUE, 0, -1),
           sig_acc(MAX_F64, SRC_VALUE, 0, -1));

// two / three plain float64 value columns: {a: SUM + COUNT (or MEAN), b: SUM (, c: SUM)} and the plain sums
constexpr uint64_t SIG_SUMF_CNT_SUMF =
  make_sig(3, sig_acc(ADD_F64, SRC_VALUE, 0, -1), sig_acc(ADD_I64, SRC_ONE, -1, -1), sig_acc(ADD_F64, SRC_VALUE, 1, -1));
constexpr uint64_t SIG_SUMF_SUMF = make_sig(2, sig_acc(ADD_F64, SRC_VALUE, 0, -1), sig_acc(ADD_F64, SRC_VALUE, 1, -1));
constexpr uint64_t SIG_SUMF_CNT_SUMF_SUMF =
  make_sig(4, sig_acc(ADD_F64, SRC_VALUE, 0, -1), sig_acc(ADD_I64, SRC_ONE, -1, -1), sig_acc(ADD_F64, SRC_VALUE, 1, -1),
           sig_acc(ADD_F64, SRC_VALUE, 2, -1));
constexpr uint64_t SIG_SUMF_SUMF_SUMF =
  make_sig(3, sig_acc(ADD_F64, SRC_VALUE, 0, -1), sig_acc(ADD_F64, SRC_VALUE, 1, -1), sig_acc(ADD_F64, SRC_VALUE, 2, -1));

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
  if (src == H_ROWID) return static_cast<uint32_t>(row);
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
  } else if (src == H_ROWID) {
#pragma unroll
    for (int k = 0; k < NR; ++k) out[k] = static_cast<uint32_t>(row[k]);
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
  // the data loads are predicated on the row being in range only - not on its validity bits - so that they do not wait
  // for the mask loads (a second exposed HBM round trip per tile); dropped rows are simply never used
  bool inrange[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) inrange[k] = live[k];
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
      batch_load_bits<NR>(col, row, inrange, raw);
      if (is_key) {
#pragma unroll
        for (int k = 0; k < NR; ++k) rec[k][u] = ((keynulls[k] >> lo) & 1u) ? 0 : normalize_key_bits(raw[k], col.cls);
      } else {
#pragma unroll
        for (int k = 0; k < NR; ++k) rec[k][u] = to_acc_bits(raw[k], col.cls, col.width);
      }
    } else {
      uint32_t l[NR], h[NR];
      batch_half<NR>(p, lo, row, inrange, keynulls, valvalid, l);
      batch_half<NR>(p, hi, row, inrange, keynulls, valvalid, h);
#pragma unroll
      for (int k = 0; k < NR; ++k) rec[k][u] = static_cast<uint64_t>(l[k]) | (static_cast<uint64_t>(h[k]) << 32);
    }
  }
}

// ------------------------------------------------------------------ hoisted record builder
// batch_units() reads its column and unit descriptors from the plan (device memory) on every call: scalar loads, field
// extraction and pointer arithmetic that the compiler cannot hoist out of the tile / batch loop (the loop stores to
// global memory) - 70-90 scalar instructions per row-lane against 25 on the plain-column path (SQ_INSTS_SALU). A kernel
// that builds records in a loop resolves the descriptors ONCE into this register-resident form (static indices only).
struct half_local {
  void const* head;  // column data (element i at head + (offset + i) * width); unused for H_* sources
  int32_t offset;
  int8_t src;        // column index (>= 0) or H_NONE / H_KEYNULLS / H_VALVALID / H_ROWID
  int8_t width;
  int8_t cls;
};
template <int UT>
struct units_local {
  bitmask_type const* mask[MAX_LOCAL_COLS];
  int32_t moff[MAX_LOCAL_COLS];
  int32_t ncols, nkeycols, drop_null_keys;
  half_local lo[UT], hi[UT];
  int8_t full[UT], is_key[UT];

  __device__ __forceinline__ void load(plan_dev const& p, int nunits)
  {
    ncols          = p.ncols;
    nkeycols       = p.nkeycols;
    drop_null_keys = p.drop_null_keys;
#pragma unroll
    for (int c = 0; c < MAX_LOCAL_COLS; ++c) {
      mask[c] = c < p.ncols ? p.cols[c].mask : nullptr;
      moff[c] = c < p.ncols ? p.cols[c].offset : 0;
    }
    auto resolve = [&](int8_t src) {
      half_local h{nullptr, 0, src, 0, 0};
      if (src >= 0) {
        device_column const c = p.cols[src];
        h.head   = c.head;
        h.offset = c.offset;
        h.width  = static_cast<int8_t>(c.width);
        h.cls    = static_cast<int8_t>(c.cls);
      }
      return h;
    };
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      unit_desc const d = u < nunits ? p.unit[u] : unit_desc{0, H_NONE, H_NONE, 0};
      full[u]   = d.full;
      is_key[u] = d.is_key;
      lo[u]     = resolve(d.lo);
      hi[u]     = resolve(d.full ? H_NONE : d.hi);
    }
  }
};

template <int NR>
__device__ __forceinline__ void load_bits_local(half_local const& h, int64_t const (&row)[NR], bool const (&live)[NR], uint64_t (&out)[NR])
{
  int64_t const off = h.offset;
  switch (h.width) {
    case 1:
#pragma unroll
      for (int k = 0; k < NR; ++k) out[k] = live[k] ? gload(static_cast<uint8_t const*>(h.head) + off + row[k]) : 0;
      break;
    case 2:
#pragma unroll
      for (int k = 0; k < NR; ++k) out[k] = live[k] ? gload(static_cast<uint16_t const*>(h.head) + off + row[k]) : 0;
      break;
    case 4:
#pragma unroll
      for (int k = 0; k < NR; ++k) out[k] = live[k] ? gload(static_cast<uint32_t const*>(h.head) + off + row[k]) : 0;
      break;
    default:
#pragma unroll
      for (int k = 0; k < NR; ++k) out[k] = live[k] ? gload(static_cast<uint64_t const*>(h.head) + off + row[k]) : 0;
  }
}

template <int NR>
__device__ __forceinline__ void half_local_bits(half_local const& h, int64_t const (&row)[NR], bool const (&live)[NR],
                                                uint32_t const (&keynulls)[NR], uint32_t const (&valvalid)[NR], uint32_t (&out)[NR])
{
  if (h.src == H_NONE) {
#pragma unroll
    for (int k = 0; k < NR; ++k) out[k] = 0;
  } else if (h.src == H_KEYNULLS) {
#pragma unroll
    for (int k = 0; k < NR; ++k) out[k] = keynulls[k];
  } else if (h.src == H_VALVALID) {
#pragma unroll
    for (int k = 0; k < NR; ++k) out[k] = valvalid[k];
  } else if (h.src == H_ROWID) {
#pragma unroll
    for (int k = 0; k < NR; ++k) out[k] = static_cast<uint32_t>(row[k]);
  } else {
    uint64_t raw[NR];
    load_bits_local<NR>(h, row, live, raw);
#pragma unroll
    for (int k = 0; k < NR; ++k) out[k] = ((keynulls[k] >> h.src) & 1u) ? 0u : static_cast<uint32_t>(normalize_key_bits(raw[k], h.cls));
  }
}

// Same contract as batch_units(), descriptors from `L` (loaded once by the caller).
template <int NR, int UT>
__device__ __forceinline__ void batch_units_local(units_local<UT> const& L, int nunits, int64_t const (&row)[NR], bool (&live)[NR],
                                                  uint64_t (&rec)[NR][UT], uint32_t (&valvalid)[NR])
{
  bool inrange[NR];
  uint32_t keynulls[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    inrange[k]  = live[k];
    keynulls[k] = 0;
    valvalid[k] = 0;
  }
#pragma unroll
  for (int c = 0; c < MAX_LOCAL_COLS; ++c) {
    if (c >= L.ncols) break;
    bool const is_key  = c < L.nkeycols;
    uint32_t const bit = is_key ? (1u << c) : (1u << (c - L.nkeycols));
    if (L.mask[c] == nullptr) {
      if (!is_key) {
#pragma unroll
        for (int k = 0; k < NR; ++k) valvalid[k] |= bit;
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      if (!inrange[k]) continue;
      int64_t const b = static_cast<int64_t>(L.moff[c]) + row[k];
      bool const v    = (gload(L.mask[c] + (b >> 5)) >> (b & 31)) & 1u;
      if (is_key) keynulls[k] |= v ? 0u : bit;
      else valvalid[k] |= v ? bit : 0u;
    }
  }
  if (L.drop_null_keys) {
#pragma unroll
    for (int k = 0; k < NR; ++k) live[k] = live[k] && keynulls[k] == 0;
  }
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    if (u >= nunits) break;
    if (L.full[u]) {
      uint64_t raw[NR];
      load_bits_local<NR>(L.lo[u], row, inrange, raw);
      if (L.is_key[u]) {
#pragma unroll
        for (int k = 0; k < NR; ++k) rec[k][u] = ((keynulls[k] >> L.lo[u].src) & 1u) ? 0 : normalize_key_bits(raw[k], L.lo[u].cls);
      } else {
#pragma unroll
        for (int k = 0; k < NR; ++k) rec[k][u] = to_acc_bits(raw[k], L.lo[u].cls, L.lo[u].width);
      }
    } else {
      uint32_t l[NR], h[NR];
      half_local_bits<NR>(L.lo[u], row, inrange, keynulls, valvalid, l);
      half_local_bits<NR>(L.hi[u], row, inrange, keynulls, valvalid, h);
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

// Virtual row range over a strided list of level-1 regions (part_args::from_regions). `pre` is a workgroup-shared
// prefix array of MAX_REGION_LIST + 1 entries filled by build(); record_of(v) maps virtual row v to its record index.
constexpr int MAX_REGION_LIST = 256;
struct region_input {
  int32_t* pre;
  int nreg;
  int64_t first;   // record index of region 0's base
  int64_t stride;  // records between the bases of consecutive listed regions
  __device__ __forceinline__ void build(part_args const& a, int item, int32_t* lds_pre)
  {
    int const g = item / a.geom.slices, s = item % a.geom.slices;
    pre    = lds_pre;
    nreg   = (a.in_slices - s + a.geom.slices - 1) / a.geom.slices;
    first  = (static_cast<int64_t>(g) * a.in_slices + s) * a.in_region_cap;
    stride = static_cast<int64_t>(a.geom.slices) * a.in_region_cap;
    if (threadIdx.x == 0) {
      int32_t run = 0;
      for (int j = 0; j < nreg; ++j) {
        pre[j] = run;
        // (clamped: a level-1 workgroup that gave up left its counts unwritten)
        run += static_cast<int32_t>(min(static_cast<int64_t>(max(a.in_region_count[static_cast<int64_t>(g) * a.in_slices + s + static_cast<int64_t>(j) * a.geom.slices], 0)), a.in_region_cap));
      }
      pre[nreg] = run;
    }
    __syncthreads();
  }
  __device__ __forceinline__ int64_t total() const { return pre[nreg]; }
  __device__ __forceinline__ int64_t record_of(int64_t v) const
  {
    int lo = 0, hi = nreg;  // invariant: pre[lo] <= v < pre[hi]
    while (hi - lo > 1) {
      int const mid = (lo + hi) >> 1;
      if (pre[mid] <= v) lo = mid; else hi = mid;
    }
    return first + static_cast<int64_t>(lo) * stride + (v - pre[lo]);
  }
};

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
}  // namespace cudf::groupby::detail
