// SPDX-License-Identifier: Apache-2.0
// cudf::inner_join / left_join / full_join — index-pair joins on key tables.
// Signatures follow reference cpp/include/cudf/join/join.hpp:160-166 (inner), :72 (JoinNoMatch); semantics
// SURVEY.md Appendix A rules 13-19: every matching (left_row, right_row) pair exactly once, order unspecified.
#pragma once
#include <cudf/table/table_view.hpp>
#include <cudf/types.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <rmm/device_uvector.hpp>
#include <limits>
#include <memory>
#include <utility>

namespace cudf {

enum class join_kind : int32_t { INNER_JOIN = 0, LEFT_JOIN = 1, FULL_JOIN = 2, LEFT_SEMI_JOIN = 3, LEFT_ANTI_JOIN = 4 };

constexpr size_type JoinNoMatch = std::numeric_limits<size_type>::min();

// Per-left-row match counts of a probe (reference join.hpp:81-108) and a row range of that probe (:120-125);
// used by hash_join::*_join_match_context / partitioned_*_join to chunk a large probe side.
struct join_match_context {
  table_view _left_table;                                          // the left table of the probe (non-owning)
  std::unique_ptr<rmm::device_uvector<size_type>> _match_counts;  // matches in the right table per left row
  join_match_context(table_view const& left_table, std::unique_ptr<rmm::device_uvector<size_type>> match_counts)
    : _left_table{left_table}, _match_counts{std::move(match_counts)}
  {
  }
  join_match_context(join_match_context const&)            = delete;
  join_match_context& operator=(join_match_context const&) = delete;
  join_match_context(join_match_context&&)                 = default;
  join_match_context& operator=(join_match_context&&)      = default;
  virtual ~join_match_context()                            = default;
};
struct join_partition_context {
  std::unique_ptr<join_match_context> left_table_context;  // from a *_join_match_context call
  size_type left_start_idx;                                // first left row of this partition
  size_type left_end_idx;                                  // one past its last left row
};

using join_index_pair = std::pair<std::unique_ptr<rmm::device_uvector<size_type>>,
                                  std::unique_ptr<rmm::device_uvector<size_type>>>;

join_index_pair inner_join(table_view const& left_keys,
                           table_view const& right_keys,
                           null_equality compare_nulls       = null_equality::EQUAL,
                           stream_ref stream                 = get_default_stream(),
                           rmm::device_async_resource_ref mr = get_current_device_resource_ref());

join_index_pair left_join(table_view const& left_keys,
                          table_view const& right_keys,
                          null_equality compare_nulls       = null_equality::EQUAL,
                          stream_ref stream                 = get_default_stream(),
                          rmm::device_async_resource_ref mr = get_current_device_resource_ref());

join_index_pair full_join(table_view const& left_keys,
                          table_view const& right_keys,
                          null_equality compare_nulls       = null_equality::EQUAL,
                          stream_ref stream                 = get_default_stream(),
                          rmm::device_async_resource_ref mr = get_current_device_resource_ref());
}  // namespace cudf
