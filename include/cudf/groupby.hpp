// SPDX-License-Identifier: Apache-2.0
// cudf::groupby::groupby — the drop-in boundary for the hash-groupby path.
// Signatures follow reference cpp/include/cudf/groupby.hpp:54-57 (aggregation_request), :81-84
// (aggregation_result), :91-96 (non-copyable/non-movable), :121-125 (ctor), :181-184 (aggregate).
// Semantics (SURVEY.md Appendix A rules 1-12): group order unspecified; results[i].results[j] answers
// requests[i].aggregations[j]; result rows align with the returned key rows; keys is a non-owning view.
#pragma once
#include <cudf/aggregation.hpp>
#include <cudf/column/column.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <rmm/resource_ref.hpp>
#include <memory>
#include <span>
#include <utility>
#include <vector>

namespace cudf {
namespace groupby {

struct aggregation_request {
  column_view values;
  std::vector<std::unique_ptr<groupby_aggregation>> aggregations;
};

struct aggregation_result {
  std::vector<std::unique_ptr<column>> results{};
};

// Which kernel family served the last aggregate() call (repo addition, for tests and benchmarks).
// (SORT: the sort-based groupby - MEDIAN / QUANTILE / NUNIQUE / NTH_ELEMENT, or pre-sorted keys whose runs are not the distinct keys)
enum class hash_path : int32_t { NONE = 0, LDS_SINGLE_PASS = 1, PARTITIONED_LDS = 2, GLOBAL_TABLE = 3, DENSE_DIRECT = 4, SORT = 5 };

class groupby {
 public:
  groupby() = delete;
  ~groupby();
  groupby(groupby const&)            = delete;
  groupby(groupby&&)                 = delete;
  groupby& operator=(groupby const&) = delete;
  groupby& operator=(groupby&&)      = delete;

  explicit groupby(table_view const& keys,
                   null_policy null_handling                      = null_policy::EXCLUDE,
                   sorted keys_are_sorted                         = sorted::NO,
                   std::vector<order> const& column_order         = {},
                   std::vector<null_order> const& null_precedence = {});

  // (reference groupby.hpp:181-184; a std::vector<aggregation_request> converts implicitly)
  std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> aggregate(
    std::span<aggregation_request const> requests,
    stream_ref stream                 = get_default_stream(),
    rmm::device_async_resource_ref mr = get_current_device_resource_ref());

  [[nodiscard]] hash_path last_path() const noexcept { return _last_path; }

 private:
  table_view _keys;
  null_policy _include_null_keys{null_policy::EXCLUDE};
  sorted _keys_are_sorted{sorted::NO};
  std::vector<order> _column_order{};
  std::vector<null_order> _null_precedence{};
  hash_path _last_path{hash_path::NONE};
};
}  // namespace groupby
}  // namespace cudf
