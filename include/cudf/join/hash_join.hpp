// SPDX-License-Identifier: Apache-2.0
// cudf::hash_join — build once on `right`, probe many times. Signatures follow reference
// cpp/include/cudf/join/hash_join.hpp:62 (nullable_join: YES == false!), :71 (class), ctor pair, inner_join,
// left_join, full_join, *_join_size. Probes are const and may run concurrently on different streams.
// Lifetime (as in libcudf): a probe returns without synchronising its stream; the hash_join object - whose tables the probe kernels
// read and which go back to the pool on the BUILD stream when it is destroyed - must outlive, or be stream-synchronised with, every
// probe stream it was used on (the Python mirror synchronises the last probe stream before cudf_amd_hash_join_destroy). Against a
// build side that took the radix partitions, *_join_size runs the radix join's count pass; every other probe that the partitioned
// paths do not take (small probe sides, full joins, match contexts) builds the open-addressing table lazily on first use, on the
// probe's stream with the memory resource given to the constructor - which therefore has to stay alive as long as the object.
#pragma once
#include <cudf/join/join.hpp>
#include <cudf/utilities/span.hpp>
#include <memory>
#include <optional>

namespace cudf {
namespace detail {
class hash_join_impl;
}

// NOTE: YES has the value `false` (reference hash_join.hpp:62) — compare against the enumerator.
enum class nullable_join : bool { YES, NO };

class hash_join {
 public:
  hash_join() = delete;
  ~hash_join();
  hash_join(hash_join const&)            = delete;
  hash_join(hash_join&&)                 = delete;
  hash_join& operator=(hash_join const&) = delete;
  hash_join& operator=(hash_join&&)      = delete;

  hash_join(table_view const& right, null_equality compare_nulls,
            stream_ref stream                 = get_default_stream(),
            rmm::device_async_resource_ref mr = get_current_device_resource_ref());
  hash_join(table_view const& right, nullable_join has_nulls, null_equality compare_nulls, double load_factor,
            stream_ref stream                 = get_default_stream(),
            rmm::device_async_resource_ref mr = get_current_device_resource_ref());

  [[nodiscard]] join_index_pair inner_join(table_view const& left,
                                           std::optional<std::size_t> output_size = {},
                                           stream_ref stream                      = get_default_stream(),
                                           rmm::device_async_resource_ref mr = get_current_device_resource_ref()) const;
  [[nodiscard]] join_index_pair left_join(table_view const& left,
                                          std::optional<std::size_t> output_size = {},
                                          stream_ref stream                      = get_default_stream(),
                                          rmm::device_async_resource_ref mr = get_current_device_resource_ref()) const;
  [[nodiscard]] join_index_pair full_join(table_view const& left,
                                          std::optional<std::size_t> output_size = {},
                                          stream_ref stream                      = get_default_stream(),
                                          rmm::device_async_resource_ref mr = get_current_device_resource_ref()) const;
  [[nodiscard]] std::size_t inner_join_size(table_view const& left, stream_ref stream = get_default_stream()) const;
  [[nodiscard]] std::size_t left_join_size(table_view const& left, stream_ref stream = get_default_stream()) const;
  [[nodiscard]] std::size_t full_join_size(table_view const& left, stream_ref stream = get_default_stream(),
                                           rmm::device_async_resource_ref mr = get_current_device_resource_ref()) const;

  // Chunked probing (reference hash_join.hpp:276-441): per-left-row match counts, then the join of a row range of the
  // left table; returned left indices refer to the complete left table. Left/full contexts count a row without a
  // match as 1 (its JoinNoMatch pair). partitioned_full_join emits the probe side only; the unmatched right rows are
  // appended once by finalize_partitioned_full_join from the collected partial results.
  [[nodiscard]] join_match_context inner_join_match_context(
    table_view const& left, stream_ref stream = get_default_stream(),
    rmm::device_async_resource_ref mr = get_current_device_resource_ref()) const;
  [[nodiscard]] join_match_context left_join_match_context(
    table_view const& left, stream_ref stream = get_default_stream(),
    rmm::device_async_resource_ref mr = get_current_device_resource_ref()) const;
  [[nodiscard]] join_match_context full_join_match_context(
    table_view const& left, stream_ref stream = get_default_stream(),
    rmm::device_async_resource_ref mr = get_current_device_resource_ref()) const;
  [[nodiscard]] join_index_pair partitioned_inner_join(
    join_partition_context const& context, stream_ref stream = get_default_stream(),
    rmm::device_async_resource_ref mr = get_current_device_resource_ref()) const;
  [[nodiscard]] join_index_pair partitioned_left_join(
    join_partition_context const& context, stream_ref stream = get_default_stream(),
    rmm::device_async_resource_ref mr = get_current_device_resource_ref()) const;
  [[nodiscard]] join_index_pair partitioned_full_join(
    join_partition_context const& context, stream_ref stream = get_default_stream(),
    rmm::device_async_resource_ref mr = get_current_device_resource_ref()) const;
  [[nodiscard]] static join_index_pair finalize_partitioned_full_join(
    cudf::host_span<cudf::device_span<size_type const> const> left_partials,
    cudf::host_span<cudf::device_span<size_type const> const> right_partials, size_type left_table_num_rows,
    size_type right_table_num_rows, stream_ref stream = get_default_stream(),
    rmm::device_async_resource_ref mr = get_current_device_resource_ref());

 private:
  std::unique_ptr<detail::hash_join_impl const> _impl;
};
}  // namespace cudf
