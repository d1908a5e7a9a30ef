// SPDX-License-Identifier: Apache-2.0
// Validity-bitmask helpers: LSB-first 32-bit words, bit set = valid
// (reference cpp/include/cudf/utilities/bit.hpp:47-104).
#pragma once
#include <cudf/types.hpp>

namespace cudf {
namespace detail {
template <typename T> constexpr CUDF_HOST_DEVICE std::size_t size_in_bits() { return sizeof(T) * 8; }
}  // namespace detail

constexpr CUDF_HOST_DEVICE size_type word_index(size_type bit_index)
{
  return bit_index / static_cast<size_type>(detail::size_in_bits<bitmask_type>());
}
constexpr CUDF_HOST_DEVICE size_type intra_word_index(size_type bit_index)
{
  return bit_index % static_cast<size_type>(detail::size_in_bits<bitmask_type>());
}
CUDF_HOST_DEVICE inline bool bit_is_set(bitmask_type const* bitmask, size_type bit_index)
{
  return bitmask[word_index(bit_index)] & (bitmask_type{1} << intra_word_index(bit_index));
}
CUDF_HOST_DEVICE inline bool bit_value_or(bitmask_type const* bitmask, size_type bit_index, bool default_value)
{
  return bitmask != nullptr ? bit_is_set(bitmask, bit_index) : default_value;
}
constexpr CUDF_HOST_DEVICE size_type num_bitmask_words(size_type number_of_bits)
{
  return (number_of_bits + 31) / 32;
}
}  // namespace cudf
