// SPDX-License-Identifier: Apache-2.0
// Source-compatible subset of the libcudf type vocabulary for the hash-groupby / hash-join hot path.
// Mirrors (names, enumerator order and meaning) reference cpp/include/cudf/types.hpp:76-77 (size_type,
// bitmask_type), :99-148 (order, null_policy, nan_policy, null_equality, null_order, sorted),
// :162-168 (mask_state), :185-217 (type_id), :279-340 (data_type). Written from scratch for HIP/gfx950.
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CUDF_HOST_DEVICE __host__ __device__
#else
#define CUDF_HOST_DEVICE
#endif
#define CUDF_EXPORT __attribute__((visibility("default")))

namespace cudf {

using size_type         = int32_t;   // row index type
using bitmask_type      = uint32_t;  // validity bitmask word, LSB-first, 1 = valid
using hash_value_type   = uint32_t;
using thread_index_type = int64_t;

enum class order : bool { ASCENDING, DESCENDING };
enum class null_policy : bool { EXCLUDE, INCLUDE };
enum class nan_policy : bool { NAN_IS_NULL, NAN_IS_VALID };
enum class nan_equality { ALL_EQUAL, UNEQUAL };
enum class null_equality : bool { EQUAL, UNEQUAL };
enum class null_order : bool { AFTER, BEFORE };
enum class sorted : bool { NO, YES };
enum class mask_state : int32_t { UNALLOCATED, UNINITIALIZED, ALL_VALID, ALL_NULL };
enum class out_of_bounds_policy : bool { NULLIFY, DONT_CHECK };
// quantile interpolation between the two neighbouring order statistics (reference types.hpp:173-180)
enum class interpolation : int32_t { LINEAR, LOWER, HIGHER, MIDPOINT, NEAREST, NEAREST_HALF_UP };

enum class type_id : int32_t {
  EMPTY,
  INT8,
  INT16,
  INT32,
  INT64,
  UINT8,
  UINT16,
  UINT32,
  UINT64,
  FLOAT32,
  FLOAT64,
  BOOL8,
  TIMESTAMP_DAYS,
  TIMESTAMP_SECONDS,
  TIMESTAMP_MILLISECONDS,
  TIMESTAMP_MICROSECONDS,
  TIMESTAMP_NANOSECONDS,
  DURATION_DAYS,
  DURATION_SECONDS,
  DURATION_MILLISECONDS,
  DURATION_MICROSECONDS,
  DURATION_NANOSECONDS,
  DICTIONARY32,
  STRING,
  LIST,
  DECIMAL32,
  DECIMAL64,
  DECIMAL128,
  STRUCT,
  NUM_TYPE_IDS
};

class data_type {
 public:
  data_type() = default;
  CUDF_HOST_DEVICE explicit constexpr data_type(type_id id) : _id{id} {}
  explicit data_type(type_id id, int32_t scale) : _id{id}, _fixed_point_scale{scale} {}
  [[nodiscard]] CUDF_HOST_DEVICE constexpr type_id id() const noexcept { return _id; }
  [[nodiscard]] CUDF_HOST_DEVICE constexpr int32_t scale() const noexcept { return _fixed_point_scale; }

 private:
  type_id _id{type_id::EMPTY};
  int32_t _fixed_point_scale{};
};

constexpr bool operator==(data_type const& lhs, data_type const& rhs)
{
  return lhs.id() == rhs.id() && lhs.scale() == rhs.scale();
}
constexpr bool operator!=(data_type const& lhs, data_type const& rhs) { return !(lhs == rhs); }

// Size in bytes of one element of a fixed-width type; 0 for non-fixed-width types
// (reference: cudf::size_of, cpp/src/utilities/type_dispatcher / traits).
CUDF_HOST_DEVICE constexpr std::size_t size_of_id(type_id id)
{
  switch (id) {
    case type_id::INT8:
    case type_id::UINT8:
    case type_id::BOOL8: return 1;
    case type_id::INT16:
    case type_id::UINT16: return 2;
    case type_id::INT32:
    case type_id::UINT32:
    case type_id::FLOAT32:
    case type_id::TIMESTAMP_DAYS:
    case type_id::DURATION_DAYS:
    case type_id::DECIMAL32: return 4;
    case type_id::INT64:
    case type_id::UINT64:
    case type_id::FLOAT64:
    case type_id::TIMESTAMP_SECONDS:
    case type_id::TIMESTAMP_MILLISECONDS:
    case type_id::TIMESTAMP_MICROSECONDS:
    case type_id::TIMESTAMP_NANOSECONDS:
    case type_id::DURATION_SECONDS:
    case type_id::DURATION_MILLISECONDS:
    case type_id::DURATION_MICROSECONDS:
    case type_id::DURATION_NANOSECONDS:
    case type_id::DECIMAL64: return 8;
    case type_id::DECIMAL128: return 16;
    default: return 0;
  }
}
std::size_t size_of(data_type t);
inline bool is_fixed_width(data_type t) { return size_of_id(t.id()) != 0; }

// type_id <-> C++ type mapping for the numeric types the hot path instantiates.
template <typename T> constexpr type_id type_to_id() { return type_id::EMPTY; }
template <> constexpr type_id type_to_id<int8_t>() { return type_id::INT8; }
template <> constexpr type_id type_to_id<int16_t>() { return type_id::INT16; }
template <> constexpr type_id type_to_id<int32_t>() { return type_id::INT32; }
template <> constexpr type_id type_to_id<int64_t>() { return type_id::INT64; }
template <> constexpr type_id type_to_id<uint8_t>() { return type_id::UINT8; }
template <> constexpr type_id type_to_id<uint16_t>() { return type_id::UINT16; }
template <> constexpr type_id type_to_id<uint32_t>() { return type_id::UINT32; }
template <> constexpr type_id type_to_id<uint64_t>() { return type_id::UINT64; }
template <> constexpr type_id type_to_id<float>() { return type_id::FLOAT32; }
template <> constexpr type_id type_to_id<double>() { return type_id::FLOAT64; }
template <> constexpr type_id type_to_id<bool>() { return type_id::BOOL8; }

}  // namespace cudf
