// SPDX-License-Identifier: Apache-2.0
// Aggregation descriptors for groupby requests. Kind values are numerically identical to the reference
// (cpp/include/cudf/aggregation.hpp:78-121) because pylibcudf passes them through as ints; factories
// mirror :212-367. Constructible on this path: the hash-groupby kinds (SUM, SUM_OVERFLOW, PRODUCT, MIN, MAX,
// COUNT_VALID, COUNT_ALL, SUM_OF_SQUARES, MEAN, M2, VARIANCE, STD, ARGMAX, ARGMIN) and the sort-groupby kinds
// MEDIAN, QUANTILE, NUNIQUE, NTH_ELEMENT; the rest of the enum is kept so that values line up.
#pragma once
#include <cudf/types.hpp>
#include <cudf/utilities/error.hpp>
#include <functional>
#include <memory>
#include <vector>

namespace cudf {

class aggregation {
 public:
  enum Kind : int32_t {
    SUM = 0,
    SUM_OVERFLOW,
    PRODUCT,
    MIN,
    MAX,
    COUNT_VALID,
    COUNT_ALL,
    ANY,
    ALL,
    SUM_OF_SQUARES,
    MEAN,
    M2,
    VARIANCE,
    STD,
    MEDIAN,
    QUANTILE,
    ARGMAX,
    ARGMIN,
    NUNIQUE,
    NTH_ELEMENT,
    ROW_NUMBER,
    EWMA,
    RANK,
    COLLECT_LIST,
    COLLECT_SET,
    LEAD,
    LAG,
    PTX,
    CUDA,
    HOST_UDF,
    MERGE_LISTS,
    MERGE_SETS,
    MERGE_M2,
    COVARIANCE,
    CORRELATION,
    TDIGEST,
    MERGE_TDIGEST,
    HISTOGRAM,
    MERGE_HISTOGRAM,
    BITWISE_AGG,
    TOP_K,
    INVALID
  };

  aggregation() = delete;
  aggregation(Kind kind_) : kind{kind_} { CUDF_EXPECTS(is_valid(), "Invalid aggregation kind"); }
  Kind kind;
  virtual ~aggregation() = default;
  [[nodiscard]] bool is_valid() const { return kind >= 0 && kind < Kind::INVALID; }
  [[nodiscard]] virtual bool is_equal(aggregation const& other) const { return kind == other.kind; }
  [[nodiscard]] virtual size_t do_hash() const { return std::hash<int>{}(kind); }
  [[nodiscard]] virtual std::unique_ptr<aggregation> clone() const = 0;
};

class groupby_aggregation : public virtual aggregation {};
class reduce_aggregation : public virtual aggregation {};

template <typename Base = aggregation> std::unique_ptr<Base> make_sum_aggregation();
// SUM with overflow detection: struct {sum: source type, overflow: bool} (reference aggregation.hpp:214-217)
template <typename Base = aggregation> std::unique_ptr<Base> make_sum_overflow_aggregation();
template <typename Base = aggregation> std::unique_ptr<Base> make_product_aggregation();
template <typename Base = aggregation> std::unique_ptr<Base> make_min_aggregation();
template <typename Base = aggregation> std::unique_ptr<Base> make_max_aggregation();
template <typename Base = aggregation>
std::unique_ptr<Base> make_count_aggregation(null_policy null_handling = null_policy::EXCLUDE);
template <typename Base = aggregation> std::unique_ptr<Base> make_sum_of_squares_aggregation();
template <typename Base = aggregation> std::unique_ptr<Base> make_mean_aggregation();
template <typename Base = aggregation> std::unique_ptr<Base> make_m2_aggregation();
template <typename Base = aggregation> std::unique_ptr<Base> make_variance_aggregation(size_type ddof = 1);
template <typename Base = aggregation> std::unique_ptr<Base> make_std_aggregation(size_type ddof = 1);
template <typename Base = aggregation> std::unique_ptr<Base> make_argmax_aggregation();
template <typename Base = aggregation> std::unique_ptr<Base> make_argmin_aggregation();
// Kinds without a hash implementation: a request holding one takes the whole call down the sort-based groupby
// (reference cpp/src/groupby/groupby.cu:64-69, cpp/src/groupby/sort/aggregate.cpp:355-440).
template <typename Base = aggregation>
std::unique_ptr<Base> make_nth_element_aggregation(size_type n, null_policy null_handling = null_policy::INCLUDE);
template <typename Base = aggregation> std::unique_ptr<Base> make_median_aggregation();
template <typename Base = aggregation>
std::unique_ptr<Base> make_quantile_aggregation(std::vector<double> const& quantiles,
                                                interpolation interp = interpolation::LINEAR);
template <typename Base = aggregation>
std::unique_ptr<Base> make_nunique_aggregation(null_policy null_handling = null_policy::EXCLUDE);

namespace detail {
// Accumulator/result type of `k` applied to a column of type `source`
// (reference cpp/include/cudf/detail/aggregation/aggregation.hpp:878-978).
data_type target_type(data_type source, aggregation::Kind k);
// True if (source type, kind) is a legal request (reference cpp/src/groupby/groupby.cu:186-214).
bool is_valid_aggregation(data_type source, aggregation::Kind k);
}  // namespace detail
}  // namespace cudf
