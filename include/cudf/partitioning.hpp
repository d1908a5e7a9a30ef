// SPDX-License-Identifier: Apache-2.0
// cudf::hash_partition (reference cpp/include/cudf/partitioning.hpp; implementation
// cpp/src/partitioning/partitioning.cu:569-760,925-975): reorders `input` so that rows of the same partition
// are contiguous and returns num_partitions + 1 row offsets: partition i = rows [offsets[i], offsets[i+1]), the last offset
// is the row count (partitioning.hpp:84-101).
// Partition of a row = row_hash(columns_to_hash) % num_partitions with the MurmurHash3_x86_32 row hash
// (hash_id::HASH_MURMUR3, default seed 0), nulls hashing to UINT32_MAX.
#pragma once
#include <cudf/table/table.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <memory>
#include <utility>
#include <vector>

namespace cudf {
enum class hash_id { HASH_IDENTITY = 0, HASH_MURMUR3 };
constexpr uint32_t DEFAULT_HASH_SEED = 0;

std::pair<std::unique_ptr<table>, std::vector<size_type>> hash_partition(
  table_view const& input,
  std::vector<size_type> const& columns_to_hash,
  int num_partitions,
  hash_id hash_function             = hash_id::HASH_MURMUR3,
  uint32_t seed                     = DEFAULT_HASH_SEED,
  stream_ref stream                 = get_default_stream(),
  rmm::device_async_resource_ref mr = get_current_device_resource_ref());

// The same with the rows hashed by the columns of a separate `keys` table (reference partitioning.hpp:118-145); throws
// std::invalid_argument when `keys` has columns and a different number of rows.
std::pair<std::unique_ptr<table>, std::vector<size_type>> hash_partition(
  table_view const& input,
  table_view const& keys,
  int num_partitions,
  hash_id hash_function             = hash_id::HASH_MURMUR3,
  uint32_t seed                     = DEFAULT_HASH_SEED,
  stream_ref stream                 = get_default_stream(),
  rmm::device_async_resource_ref mr = get_current_device_resource_ref());

namespace hashing {
// Row hash column (UINT32) — reference cudf::hashing::murmurhash3_x86_32 (cpp/include/cudf/hashing.hpp).
std::unique_ptr<column> murmurhash3_x86_32(table_view const& input, uint32_t seed = DEFAULT_HASH_SEED,
                                           stream_ref stream                 = get_default_stream(),
                                           rmm::device_async_resource_ref mr = get_current_device_resource_ref());
}  // namespace hashing
}  // namespace cudf
