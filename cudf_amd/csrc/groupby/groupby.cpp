// SPDX-License-Identifier: Apache-2.0
// Host side of cudf::groupby::groupby for the hash path: request validation, aggregation flattening, record
// planning, strategy selection and kernel orchestration.
// Reference counterparts: cpp/src/groupby/groupby.cu:39-70,186-236 (ctor, validation, dispatch),
// cpp/src/groupby/hash/groupby.cu:33-147 (hash dispatch), extract_single_pass_aggs.cpp:26-177 (flattening),
// output_utils.cu:49-224 (result columns), hash_compound_agg_finalizer.cu:92-133 (MEAN).
#include "call.hpp"
#include "../common/profiler.hpp"

#include <cudf/groupby.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/utilities/error.hpp>

#include <algorithm>
#include <string>

namespace cudf {

// ---------------------------------------------------------------------------------------------------------
// aggregation descriptors (reference cpp/src/aggregation/aggregation.cpp factories)
namespace detail {
class simple_aggregation final : public groupby_aggregation, public reduce_aggregation {
 public:
  explicit simple_aggregation(aggregation::Kind k) : aggregation{k} {}
  [[nodiscard]] std::unique_ptr<aggregation> clone() const override
  {
    return std::make_unique<simple_aggregation>(*this);
  }
};

data_type target_type(data_type source, aggregation::Kind k)
{
  auto const id  = source.id();
  auto const cls = class_of(id);
  bool const plain_numeric = id >= type_id::INT8 && id <= type_id::BOOL8;
  bool const is_duration   = id >= type_id::DURATION_DAYS && id <= type_id::DURATION_NANOSECONDS;
  bool const is_decimal    = id == type_id::DECIMAL32 || id == type_id::DECIMAL64;
  auto const invalid       = data_type{type_id::EMPTY};
  if (cls == CLS_NONE) return invalid;
  switch (k) {
    case aggregation::MIN:
    case aggregation::MAX: return source;
    case aggregation::COUNT_VALID:
    case aggregation::COUNT_ALL:
    case aggregation::ARGMAX:
    case aggregation::ARGMIN: return data_type{type_id::INT32};
    case aggregation::MEAN:
      if (plain_numeric) return data_type{type_id::FLOAT64};
      if (is_duration || is_decimal) return source;
      return invalid;
    case aggregation::SUM:
      if (cls == CLS_F32 || cls == CLS_F64) return source;
      if (plain_numeric) return data_type{type_id::INT64};
      if (is_duration || is_decimal) return source;
      return invalid;
    case aggregation::SUM_OVERFLOW:
      // signed integers (not bool) and decimals: struct {sum: source type, overflow: bool}
      // (reference detail/aggregation/aggregation.hpp:981-995)
      if ((cls == CLS_SINT && plain_numeric) || is_decimal) return data_type{type_id::STRUCT};
      return invalid;
    case aggregation::PRODUCT:
    case aggregation::SUM_OF_SQUARES:
      if (cls == CLS_F32 || cls == CLS_F64) return source;
      if (plain_numeric) return data_type{type_id::INT64};
      return invalid;
    case aggregation::M2:
    case aggregation::VARIANCE:
    case aggregation::STD: return plain_numeric ? data_type{type_id::FLOAT64} : invalid;
    // sort-groupby kinds (reference detail/aggregation/aggregation.hpp:1015-1049)
    case aggregation::NTH_ELEMENT: return source;
    case aggregation::MEDIAN:
    case aggregation::QUANTILE: return data_type{type_id::FLOAT64};
    case aggregation::NUNIQUE: return data_type{type_id::INT32};
    default: return invalid;
  }
}
bool is_valid_aggregation(data_type source, aggregation::Kind k) { return target_type(source, k).id() != type_id::EMPTY; }

device_table make_device_table(table_view const& t)
{
  CUDF_EXPECTS(t.num_columns() <= MAX_COLS, "Too many columns for the hash path (limit 16).");
  device_table d{};
  d.ncols = t.num_columns();
  d.nrows = t.num_rows();
  for (int c = 0; c < d.ncols; ++c) {
    auto const& col = t.column(c);
    auto const w    = size_of_id(col.type().id());
    CUDF_EXPECTS(w >= 1 && w <= 8, "Only fixed-width columns of at most 8 bytes are supported on the hash path.");
    d.col[c] = make_device_column(col);
    if (!col.has_nulls()) d.col[c].mask = nullptr;
  }
  return d;
}
}  // namespace detail

#define CUDF_AMD_FACTORY(fn, kind)                                               \
  template <typename Base>                                                       \
  std::unique_ptr<Base> fn()                                                     \
  {                                                                              \
    return std::make_unique<detail::simple_aggregation>(aggregation::kind);      \
  }                                                                              \
  template std::unique_ptr<aggregation> fn<aggregation>();                       \
  template std::unique_ptr<groupby_aggregation> fn<groupby_aggregation>();
CUDF_AMD_FACTORY(make_sum_aggregation, SUM)
CUDF_AMD_FACTORY(make_sum_overflow_aggregation, SUM_OVERFLOW)
CUDF_AMD_FACTORY(make_product_aggregation, PRODUCT)
CUDF_AMD_FACTORY(make_min_aggregation, MIN)
CUDF_AMD_FACTORY(make_max_aggregation, MAX)
CUDF_AMD_FACTORY(make_sum_of_squares_aggregation, SUM_OF_SQUARES)
CUDF_AMD_FACTORY(make_mean_aggregation, MEAN)
CUDF_AMD_FACTORY(make_m2_aggregation, M2)
CUDF_AMD_FACTORY(make_argmax_aggregation, ARGMAX)
CUDF_AMD_FACTORY(make_argmin_aggregation, ARGMIN)
CUDF_AMD_FACTORY(make_median_aggregation, MEDIAN)
#undef CUDF_AMD_FACTORY

template <typename Base>
std::unique_ptr<Base> make_count_aggregation(null_policy null_handling)
{
  return std::make_unique<detail::simple_aggregation>(null_handling == null_policy::INCLUDE ? aggregation::COUNT_ALL
                                                                                            : aggregation::COUNT_VALID);
}
template std::unique_ptr<aggregation> make_count_aggregation<aggregation>(null_policy);
template std::unique_ptr<groupby_aggregation> make_count_aggregation<groupby_aggregation>(null_policy);
template <typename Base>
std::unique_ptr<Base> make_variance_aggregation(size_type ddof)
{
  return std::make_unique<detail::ddof_aggregation>(aggregation::VARIANCE, ddof);
}
template std::unique_ptr<aggregation> make_variance_aggregation<aggregation>(size_type);
template std::unique_ptr<groupby_aggregation> make_variance_aggregation<groupby_aggregation>(size_type);
template <typename Base>
std::unique_ptr<Base> make_std_aggregation(size_type ddof)
{
  return std::make_unique<detail::ddof_aggregation>(aggregation::STD, ddof);
}
template std::unique_ptr<aggregation> make_std_aggregation<aggregation>(size_type);
template std::unique_ptr<groupby_aggregation> make_std_aggregation<groupby_aggregation>(size_type);
template <typename Base>
std::unique_ptr<Base> make_nth_element_aggregation(size_type n, null_policy null_handling)
{
  return std::make_unique<detail::nth_element_aggregation>(n, null_handling);
}
template <typename Base>
std::unique_ptr<Base> make_quantile_aggregation(std::vector<double> const& quantiles, interpolation interp)
{
  return std::make_unique<detail::quantile_aggregation>(quantiles, interp);
}
template std::unique_ptr<aggregation> make_quantile_aggregation<aggregation>(std::vector<double> const&, interpolation);
template std::unique_ptr<groupby_aggregation> make_quantile_aggregation<groupby_aggregation>(std::vector<double> const&, interpolation);
template <typename Base>
std::unique_ptr<Base> make_nunique_aggregation(null_policy null_handling)
{
  return std::make_unique<detail::nunique_aggregation>(null_handling);
}
template std::unique_ptr<aggregation> make_nunique_aggregation<aggregation>(null_policy);
template std::unique_ptr<groupby_aggregation> make_nunique_aggregation<groupby_aggregation>(null_policy);
template std::unique_ptr<aggregation> make_nth_element_aggregation<aggregation>(size_type, null_policy);
template std::unique_ptr<groupby_aggregation> make_nth_element_aggregation<groupby_aggregation>(size_type, null_policy);


namespace groupby {

groupby::groupby(table_view const& keys, null_policy null_handling, sorted keys_are_sorted,
                 std::vector<order> const& column_order, std::vector<null_order> const& null_precedence)
  : _keys{keys}, _include_null_keys{null_handling}, _keys_are_sorted{keys_are_sorted}, _column_order{column_order},
    _null_precedence{null_precedence}
{
}
groupby::~groupby() = default;

namespace {
// empty input: typed empty outputs (reference groupby.cu:233, :87-182)
std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> empty_results(table_view const& keys,
                                                                                 std::span<aggregation_request const> requests)
{
    std::vector<aggregation_result> res;
    for (auto const& r : requests) {
      aggregation_result ar;
      for (auto const& a : r.aggregations) {
        if (a->kind == aggregation::SUM_OVERFLOW) {
          std::vector<std::unique_ptr<column>> children;
          children.push_back(make_empty_column(r.values.type()));
          children.push_back(make_empty_column(data_type{type_id::BOOL8}));
          ar.results.push_back(std::make_unique<column>(data_type{type_id::STRUCT}, 0, rmm::device_buffer{}, rmm::device_buffer{}, 0, std::move(children)));
          continue;
        }
        ar.results.push_back(make_empty_column(cudf::detail::target_type(r.values.type(), a->kind)));
      }
      res.push_back(std::move(ar));
    }
  return {empty_like(keys), std::move(res)};
}
}  // namespace

std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> groupby::aggregate(
  std::span<aggregation_request const> requests, stream_ref stream, rmm::device_async_resource_ref mr)
{
  CUDF_FUNC_RANGE();  // (reference groupby.cu:224)
  using namespace detail;
  // keys_are_sorted == YES: the reference takes its sort-based path WITHOUT a sort (groupby.cu:64-69) - the groups are the RUNS of equal
  // adjacent keys, also where the caller's claim is wrong (keys_tests.cpp:187-213 pins a NULL key that appears in two runs as two
  // groups). Where the claim holds, runs and distinct keys are the same groups and the hash path answers 5 x faster (1B sorted rows:
  // 5 ms against 24 ms on the run labels): the hash kinds are answered by it and the answer is kept if its group count equals the
  // number of runs (one coalesced pass over the keys); otherwise the call is redone on the runs.
  // reference groupby.cu:225-229
  CUDF_EXPECTS(std::all_of(requests.begin(), requests.end(),
                           [this](auto const& r) { return r.values.size() == _keys.num_rows(); }),
               "Size mismatch between request values and groupby keys.");
  // reference groupby.cu:186-201
  CUDF_EXPECTS(std::all_of(requests.begin(), requests.end(),
                           [](auto const& r) {
                             return std::all_of(r.aggregations.begin(), r.aggregations.end(), [&r](auto const& a) {
                               return cudf::detail::is_valid_aggregation(r.values.type(), a->kind);
                             });
                           }),
               "Invalid type/aggregation combination.");
  bool needs_sort = false;
  for (auto const& r : requests)
    for (auto const& a : r.aggregations) {
      CUDF_EXPECTS(is_engine_kind(a->kind) || is_sort_kind(a->kind), "Unsupported groupby aggregation on this path.");
      needs_sort = needs_sort || is_sort_kind(a->kind);
    }

  if (_keys.num_rows() == 0) {
    _last_path = hash_path::NONE;
    return empty_results(_keys, requests);
  }
  // one kind without a hash implementation takes the whole call down the sort-based path (reference groupby.cu:64-69,
  // hash/groupby.cu can_use_hash_groupby)
  if (needs_sort) {
    _last_path = hash_path::SORT;
    return sort_aggregate(_keys, _include_null_keys, _keys_are_sorted == sorted::YES, requests, stream, mr);
  }
  // pre-sorted keys whose runs are not the distinct keys (the claim was wrong): the call again, on the runs
  // (null keys to drop: the reference sorts such keys whatever the caller claims, sort_helper.cu:51-54 - groups by key, as the hash path's)
  auto const on_runs_if_needed = [&](std::pair<std::unique_ptr<table>, std::vector<aggregation_result>>&& answer) {
    if (_keys_are_sorted == sorted::YES && !(_include_null_keys == null_policy::EXCLUDE && cudf::has_nulls(_keys)) &&
        count_key_runs(_keys, stream) != static_cast<int64_t>(answer.first->num_rows())) {
      _last_path = hash_path::SORT;
      return sort_aggregate(_keys, _include_null_keys, true, requests, stream, mr);
    }
    return std::move(answer);
  };
  // ARGMIN / ARGMAX over one integer key column of a small range: MIN / MAX (on whatever path those take) + one lookup pass (arg_lookup.hip)
  if (auto answered = arg_by_lookup(_keys, _include_null_keys, requests, stream, mr, &_last_path)) return on_runs_if_needed(std::move(*answered));
  // plan -> estimate -> attempts (one executor per path: call.hpp) -> result columns
  aggregate_call call{_keys, _include_null_keys, requests, stream.value()};
  try {
    call.run();
  } catch (...) {
    _last_path = call.path;
    throw;
  }
  _last_path = call.path;
  return on_runs_if_needed(call.finalize(_keys, requests, stream, mr));
}

}  // namespace groupby
}  // namespace cudf
