// SPDX-License-Identifier: Apache-2.0
// Device-side view of Arrow-layout columns plus the row primitives every kernel on the path shares:
// element loads, validity, the reference-compatible MurmurHash3_x86_32 row hash and row equality.
// Plays the role of the reference's column_device_view / table_device_view
// (cpp/include/cudf/column/column_device_view_base.cuh:163,246; table/table_device_view.cuh) and of
// detail/row_operator/{hashing,equality}.cuh — restated for 64-lane waves, passed BY VALUE in kernel
// arguments (no host->device descriptor copy + sync as in row_operators.cu:847-860).
#pragma once
#include <cudf/column/column_view.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf/types.hpp>
#include <hip/hip_runtime.h>

namespace cudf::detail {

constexpr int MAX_COLS = 16;  // fixed-width columns per table on the fast path (kernel-argument resident)

enum elem_class : int32_t { CLS_NONE = 0, CLS_SINT = 1, CLS_UINT = 2, CLS_F32 = 3, CLS_F64 = 4, CLS_BOOL = 5 };

struct device_column {
  void const* head;           // element i at head + (offset + i) * width
  bitmask_type const* mask;   // bit (offset + i), LSB-first, 1 = valid; nullptr = all valid
  int32_t offset;
  int32_t width;              // 1, 2, 4, 8
  int32_t cls;                // elem_class
  int32_t type;               // type_id
};

struct device_table {
  device_column col[MAX_COLS];
  int32_t ncols;
  int32_t nrows;
};

inline elem_class class_of(type_id t)
{
  switch (t) {
    case type_id::INT8: case type_id::INT16: case type_id::INT32: case type_id::INT64:
    case type_id::TIMESTAMP_DAYS: case type_id::TIMESTAMP_SECONDS: case type_id::TIMESTAMP_MILLISECONDS:
    case type_id::TIMESTAMP_MICROSECONDS: case type_id::TIMESTAMP_NANOSECONDS: case type_id::DURATION_DAYS:
    case type_id::DURATION_SECONDS: case type_id::DURATION_MILLISECONDS: case type_id::DURATION_MICROSECONDS:
    case type_id::DURATION_NANOSECONDS: case type_id::DECIMAL32: case type_id::DECIMAL64: return CLS_SINT;
    case type_id::UINT8: case type_id::UINT16: case type_id::UINT32: case type_id::UINT64: return CLS_UINT;
    case type_id::FLOAT32: return CLS_F32;
    case type_id::FLOAT64: return CLS_F64;
    case type_id::BOOL8: return CLS_BOOL;
    default: return CLS_NONE;
  }
}

inline device_column make_device_column(column_view const& c)
{
  device_column d{};
  d.head   = c.head();
  d.mask   = c.nullable() ? c.null_mask() : nullptr;
  d.offset = c.offset();
  d.width  = static_cast<int32_t>(size_of_id(c.type().id()));
  d.cls    = class_of(c.type().id());
  d.type   = static_cast<int32_t>(c.type().id());
  return d;
}

// Throws if the table cannot be described on the fast path (non fixed-width, > 8-byte, too many columns).
device_table make_device_table(table_view const& t);

#if defined(__HIPCC__)

// Pointers read out of an argument struct are generic ("flat") to the compiler; flat loads count on BOTH
// vmcnt and lgkmcnt and so serialise against LDS traffic. Everything the path reads from HBM goes through
// these explicit global-address-space accessors instead (global_load_* / global_store_*).
#define CUDF_AMD_GLOBAL_AS __attribute__((address_space(1)))
// one 16-byte record: moves as global_load/store_dwordx4 and ds_read/write_b128
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ T gload(T const* p)
{
  return *(CUDF_AMD_GLOBAL_AS T const*)(p);
}
template <typename T>
__device__ __forceinline__ void gstore(T* p, T v)
{
  *(CUDF_AMD_GLOBAL_AS T*)(p) = v;
}

// streaming (read-once / write-once) variants: the `nt` cache policy keeps such lines from displacing the
// partially written lines the partition scatter needs the L2 to merge
template <typename T>
__device__ __forceinline__ T gload_stream(T const* p)
{
  return __builtin_nontemporal_load((CUDF_AMD_GLOBAL_AS T const*)(p));
}
template <typename T>
__device__ __forceinline__ void gstore_stream(T* p, T v)
{
  __builtin_nontemporal_store(v, (CUDF_AMD_GLOBAL_AS T*)(p));
}

__device__ __forceinline__ bool col_is_valid(device_column const& c, int64_t i)
{
  if (c.mask == nullptr) return true;
  int64_t const b = static_cast<int64_t>(c.offset) + i;
  return (gload(c.mask + (b >> 5)) >> (b & 31)) & 1u;
}

// Raw element bits, zero-extended to 64 bits.
__device__ __forceinline__ uint64_t col_load_bits(device_column const& c, int64_t i)
{
  int64_t const e = static_cast<int64_t>(c.offset) + i;
  switch (c.width) {
    case 1: return gload(static_cast<uint8_t const*>(c.head) + e);
    case 2: return gload(static_cast<uint16_t const*>(c.head) + e);
    case 4: return gload(static_cast<uint32_t const*>(c.head) + e);
    default: return gload(static_cast<uint64_t const*>(c.head) + e);
  }
}

// Bits used for hashing / equality: bool -> 0/1; floats normalised (-0 -> +0, NaN -> canonical quiet NaN),
// reference hashing/detail/hash_functions.cuh:19-37 and row_operator/equality.cuh:59-89.
__device__ __forceinline__ uint64_t normalize_key_bits(uint64_t bits, int32_t cls)
{
  if (cls == CLS_BOOL) return bits != 0;
  if (cls == CLS_F32) {
    uint32_t b = static_cast<uint32_t>(bits);
    if ((b & 0x7fffffffu) == 0) return 0;
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0x7fc00000u;
    return b;
  }
  if (cls == CLS_F64) {
    if ((bits & 0x7fffffffffffffffull) == 0) return 0;
    if ((bits & 0x7fffffffffffffffull) > 0x7ff0000000000000ull) return 0x7ff8000000000000ull;
  }
  return bits;
}

// Element converted to its 8-byte accumulator class: SINT -> sign-extended int64, UINT -> zero-extended,
// BOOL -> 0/1, F32/F64 -> double bits (targets: reference detail/aggregation/aggregation.hpp:878-978).
__device__ __forceinline__ uint64_t col_load_acc_bits(device_column const& c, int64_t i)
{
  uint64_t const raw = col_load_bits(c, i);
  switch (c.cls) {
    case CLS_SINT:
      switch (c.width) {
        case 1: return static_cast<uint64_t>(static_cast<int64_t>(static_cast<int8_t>(raw)));
        case 2: return static_cast<uint64_t>(static_cast<int64_t>(static_cast<int16_t>(raw)));
        case 4: return static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(raw)));
        default: return raw;
      }
    case CLS_BOOL: return raw != 0;
    case CLS_F32: return __double_as_longlong(static_cast<double>(__uint_as_float(static_cast<uint32_t>(raw))));
    default: return raw;
  }
}

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// MurmurHash3_x86_32 of a 1/2/4/8-byte little-endian value (public algorithm; reference wraps
// cuco::murmurhash3_32 in hashing/detail/murmurhash3_x86_32.cuh:21-45).
__device__ __forceinline__ uint32_t murmur3_32_bits(uint64_t bits, int width, uint32_t seed)
{
  constexpr uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  uint32_t h = seed;
  auto body  = [&](uint32_t k) {
    k *= c1; k = rotl32(k, 15); k *= c2;
    h ^= k; h = rotl32(h, 13); h = h * 5 + 0xe6546b64u;
  };
  if (width == 8) {
    body(static_cast<uint32_t>(bits));
    body(static_cast<uint32_t>(bits >> 32));
  } else if (width == 4) {
    body(static_cast<uint32_t>(bits));
  } else {  // 1- or 2-byte tail
    uint32_t k = static_cast<uint32_t>(bits) & (width == 1 ? 0xffu : 0xffffu);
    k *= c1; k = rotl32(k, 15); k *= c2;
    h ^= k;
  }
  h ^= static_cast<uint32_t>(width);
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}

__device__ __forceinline__ uint32_t hash_combine32(uint32_t lhs, uint32_t rhs)
{
  return lhs ^ (rhs + 0x9e3779b9u + (lhs << 6) + (lhs >> 2));
}

// Element hash: null -> UINT32_MAX (reference row_operator/hashing.cuh:54-73).
__device__ __forceinline__ uint32_t element_hash(device_column const& c, int64_t i, uint32_t seed)
{
  if (!col_is_valid(c, i)) return 0xffffffffu;
  return murmur3_32_bits(normalize_key_bits(col_load_bits(c, i), c.cls), c.width, seed);
}

// Row hash: first column's hash is the init, the rest folded with hash_combine (hashing.cuh:118-134).
__device__ __forceinline__ uint32_t row_hash(device_table const& t, int64_t i, uint32_t seed)
{
  if (t.ncols == 0) return seed;
  uint32_t h = element_hash(t.col[0], i, seed);
  for (int c = 1; c < t.ncols; ++c) h = hash_combine32(h, element_hash(t.col[c], i, seed));
  return h;
}

__device__ __forceinline__ bool row_has_null(device_table const& t, int64_t i)
{
  for (int c = 0; c < t.ncols; ++c)
    if (!col_is_valid(t.col[c], i)) return true;
  return false;
}

// Row equality between row i of a and row j of b (same schema): both null -> nulls_equal, one null -> false,
// NaN == NaN, -0 == +0 (reference row_operator/equality.cuh:128-142,244-262).
__device__ __forceinline__ bool rows_equal(device_table const& a, int64_t i, device_table const& b, int64_t j,
                                           bool nulls_equal)
{
  for (int c = 0; c < a.ncols; ++c) {
    bool const va = col_is_valid(a.col[c], i), vb = col_is_valid(b.col[c], j);
    if (!va || !vb) {
      if (!(va == vb && nulls_equal)) return false;
      continue;
    }
    if (normalize_key_bits(col_load_bits(a.col[c], i), a.col[c].cls) !=
        normalize_key_bits(col_load_bits(b.col[c], j), b.col[c].cls))
      return false;
  }
  return true;
}

// 64-bit mixer used by the groupby / join engines for partition digits and table slots (the engines' own
// hash; results do not depend on it — SURVEY.md §8c).
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
  x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32;
  return x;
}
#endif  // __HIPCC__

}  // namespace cudf::detail
