// SPDX-License-Identifier: Apache-2.0
// Host side of one cudf::groupby::groupby::aggregate call on the hash path, split into an immutable PLAN (host_plan: record
// layout, accumulators, result columns; planner_env: the switches of the planner, read once per call) and one EXECUTOR per path
// (methods of aggregate_call, one translation unit per family):
//   plan.cpp         planner_env::load, build_plan (columns, key units, payload units, accumulators)
//   estimate.cpp     geometry, the sample pass (group estimate, key ranges, heavy hitters, clustered rows), escalation
//   paths_dense.cpp  T (one direct-address table), D (ring scatter / write-combining scatter + direct-address tables)
//   paths_hash.cpp   S (single pass), A (local pre-aggregation), P (radix partition: optimistic one / two levels, exact)
//   finalize.cpp     partial records -> typed result columns
//   groupby.cpp      the public entry: validation, empty input, the attempt loop
// Reference counterparts: cpp/src/groupby/groupby.cu:39-70,186-236, hash/groupby.cu:33-147, compute_groupby.cu:51-155,
// compute_single_pass_aggs.cuh:32-164, extract_single_pass_aggs.cpp:26-177, output_utils.cu:49-224.
#pragma once
#include "engine.hpp"

#include <cudf/groupby.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/utilities/error.hpp>

#include <chrono>
#include <optional>
#include <span>
#include <string>
#include <vector>

namespace cudf::detail {
// (groupby.cpp) the ddof-carrying aggregation descriptor and the (source type, kind) -> result type table
// (reference cpp/src/aggregation/aggregation.cpp, detail/aggregation/aggregation.hpp:981-1180)
class ddof_aggregation final : public groupby_aggregation, public reduce_aggregation {
 public:
  ddof_aggregation(aggregation::Kind k, size_type ddof) : aggregation{k}, _ddof{ddof} {}
  [[nodiscard]] bool is_equal(aggregation const& other) const override
  {
    auto const* o = dynamic_cast<ddof_aggregation const*>(&other);
    return o != nullptr && aggregation::is_equal(other) && o->_ddof == _ddof;
  }
  [[nodiscard]] size_t do_hash() const override { return aggregation::do_hash() ^ std::hash<int>{}(_ddof); }
  [[nodiscard]] std::unique_ptr<aggregation> clone() const override { return std::make_unique<ddof_aggregation>(*this); }
  size_type _ddof;
};
// the sort-groupby kinds' descriptors (reference detail/aggregation/aggregation.hpp:269-283, :335-376)
class nth_element_aggregation final : public groupby_aggregation, public reduce_aggregation {
 public:
  nth_element_aggregation(size_type n, null_policy null_handling)
    : aggregation{aggregation::NTH_ELEMENT}, _n{n}, _null_handling{null_handling}
  {
  }
  [[nodiscard]] bool is_equal(aggregation const& other) const override
  {
    auto const* o = dynamic_cast<nth_element_aggregation const*>(&other);
    return o != nullptr && aggregation::is_equal(other) && o->_n == _n && o->_null_handling == _null_handling;
  }
  [[nodiscard]] size_t do_hash() const override
  {
    return aggregation::do_hash() ^ std::hash<int>{}(_n) ^ std::hash<int>{}(static_cast<int>(_null_handling));
  }
  [[nodiscard]] std::unique_ptr<aggregation> clone() const override { return std::make_unique<nth_element_aggregation>(*this); }
  size_type _n;
  null_policy _null_handling;
};
class quantile_aggregation final : public groupby_aggregation, public reduce_aggregation {
 public:
  quantile_aggregation(std::vector<double> const& q, interpolation i) : aggregation{aggregation::QUANTILE}, _quantiles{q}, _interpolation{i}
  {
  }
  [[nodiscard]] bool is_equal(aggregation const& other) const override
  {
    auto const* o = dynamic_cast<quantile_aggregation const*>(&other);
    return o != nullptr && aggregation::is_equal(other) && o->_interpolation == _interpolation && o->_quantiles == _quantiles;
  }
  [[nodiscard]] size_t do_hash() const override
  {
    size_t h = aggregation::do_hash() ^ std::hash<int>{}(static_cast<int>(_interpolation));
    for (double q : _quantiles) h ^= std::hash<double>{}(q) + 0x9e3779b9u + (h << 6) + (h >> 2);
    return h;
  }
  [[nodiscard]] std::unique_ptr<aggregation> clone() const override { return std::make_unique<quantile_aggregation>(*this); }
  std::vector<double> _quantiles;
  interpolation _interpolation;
};
class nunique_aggregation final : public groupby_aggregation, public reduce_aggregation {
 public:
  explicit nunique_aggregation(null_policy null_handling) : aggregation{aggregation::NUNIQUE}, _null_handling{null_handling} {}
  [[nodiscard]] bool is_equal(aggregation const& other) const override
  {
    auto const* o = dynamic_cast<nunique_aggregation const*>(&other);
    return o != nullptr && aggregation::is_equal(other) && o->_null_handling == _null_handling;
  }
  [[nodiscard]] size_t do_hash() const override { return aggregation::do_hash() ^ std::hash<int>{}(static_cast<int>(_null_handling)); }
  [[nodiscard]] std::unique_ptr<aggregation> clone() const override { return std::make_unique<nunique_aggregation>(*this); }
  null_policy _null_handling;
};
data_type target_type(data_type source, aggregation::Kind k);
bool is_valid_aggregation(data_type source, aggregation::Kind k);
}  // namespace cudf::detail

namespace cudf::groupby::detail {

// Every switch of the planner. Read ONCE per call - a call never sees two values of one switch - and not
// once per process: the test suite flips them between calls of one process (forced paths, shrunken tables).
struct planner_env {
  int64_t lds_kb, agg_block, big_min_rows, estimate_min_rows, forced_p, s_items, preagg_min_pct, dense_log2p, scatter_block, rpt;
  int64_t one_table_min_rows, sample_div;
  int64_t dense_nsplit, wc_g, slices, plan_load_pct;  // -1: not set (the default depends on the plan)
  bool dense, dense_composite, dense_one_table, dense_ring, dense_multi, hot, preagg, optimistic, optimistic2, exact, wc, cyclic,
    stamps, debug, no_simple, vec16, trace, collapse_runs;
  int64_t static_shapes;  // composite dense keys: 0 = run-time loader, 1 / 2 = compiled column shapes with one / two tiles of loads in flight
  static planner_env load();
};

struct result_spec {  // one per (request, aggregation)
  aggregation::Kind kind;
  data_type target;
  int value_idx;  // distinct value column
  int a0{-1}, a1{-1}, a2{-1}, valid_acc{-1};
  int ddof{1};
  bool nullable{false};
  int acc_cls{0};
};

struct host_plan {
  plan_dev dev{};
  std::vector<column_view> value_cols;  // distinct
  std::vector<result_spec> results;     // flattened in request order
  // key column c -> (unit, half: 0 lo / 1 hi / 2 full)
  int key_unit[MAX_COLS]{};
  int key_half[MAX_COLS]{};
  int key_raw_vidx[MAX_COLS];  // float key column -> its slot among the value columns (-1: not a float key)
  int key_acc[MAX_COLS];       // ... -> the ANY_U64 accumulator carrying a representative row's bits
  int keynulls_unit{-1}, keynulls_hi{0};
};

host_plan build_plan(table_view const& keys, null_policy policy, std::span<aggregation_request const> requests, planner_env const& env);
// Heavy-hitter handling covers plans whose accumulators are SUMs of the single value column and row COUNTs (no nulls).
bool hot_plan_ok(plan_dev const& p);
bool is_engine_kind(aggregation::Kind k);
// MEDIAN / QUANTILE / NUNIQUE / NTH_ELEMENT: served by the sort-based groupby (sort_groupby.hip)
inline bool is_sort_kind(aggregation::Kind k)
{
  return k == aggregation::MEDIAN || k == aggregation::QUANTILE || k == aggregation::NUNIQUE || k == aggregation::NTH_ELEMENT;
}
// ARGMIN / ARGMAX over one integer key column of a small range as MIN / MAX + one lookup pass over the rows (arg_lookup.hip); nullopt: not
// such a call (the engine's own ARGMIN / ARGMAX). *path = the path of the MIN / MAX call.
std::optional<std::pair<std::unique_ptr<table>, std::vector<aggregation_result>>> arg_by_lookup(table_view const& keys, null_policy include_null_keys,
                                                                                              std::span<aggregation_request const> requests,
                                                                                              stream_ref stream, rmm::device_async_resource_ref mr,
                                                                                              hash_path* path);
// (sort_groupby.hip) the number of runs of equal adjacent key rows, nulls equal nulls
int64_t count_key_runs(table_view const& keys, stream_ref stream);
// The sort-based groupby (reference cpp/src/groupby/sort/aggregate.cpp:798-830, sort_helper.cu): radix-sorts the rows by key,
// labels the groups and serves every request of the call; the unique keys come back in ascending order, nulls last.
std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> sort_aggregate(table_view const& keys, null_policy include_null_keys,
                                                                                  bool keys_are_sorted,
                                                                                  std::span<aggregation_request const> requests,
                                                                                  stream_ref stream, rmm::device_async_resource_ref mr);
double hyperloglog_estimate(std::vector<uint32_t> const& regs);
// Page-locked host staging for a call's small read-backs (per thread; the two buffers stay valid next to each other).
int32_t* pinned_ints(std::size_t count);
unsigned char* pinned_bytes(std::size_t count);

// CUDF_AMD_GB_TRACE=1: host-side timeline of a call (microseconds since entry at every mark), printed to stderr when the call returns.
struct call_trace {
  bool on;
  std::chrono::steady_clock::time_point t0;
  std::string line;
  explicit call_trace(bool enabled) : on{enabled}, t0{std::chrono::steady_clock::now()} {}
  void mark(char const* what)
  {
    if (!on) return;
    auto const us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    line += std::string(line.empty() ? "" : " | ") + what + " " + std::to_string(us);
  }
  ~call_trace()
  {
    if (on) {
      mark("return");
      fprintf(stderr, "[cudf_amd] groupby trace (us): %s\n", line.c_str());
    }
  }
};

struct scratch {  // stream-ordered temporaries from the current device resource
  hipStream_t stream;
  rmm::device_async_resource_ref mr;
  std::vector<rmm::device_buffer> bufs;
  template <typename T>
  T* alloc(std::size_t n)
  {
    bufs.emplace_back(std::max<std::size_t>(n, 1) * sizeof(T), stream, mr);
    return static_cast<T*>(bufs.back().data());
  }
};

// What one path made of one attempt.
enum class outcome {
  skip,           // the path does not apply to this call (nothing was launched): the next path in order takes the attempt
  done,           // partial / d_count / nitems / final_cap / h_count are final
  retry_free,     // the path ruled itself out (a key outside the sampled range, a region overflow, ...): redo, not counted
  retry_counted   // a table overflowed and escalate() planned more tables: redo, counted against the attempt limit
};

class aggregate_call {
 public:
  aggregate_call(table_view const& keys, null_policy policy, std::span<aggregation_request const> requests, hipStream_t stream);
  // estimate + attempts; leaves the partial records of the successful attempt for finalize()
  void run();
  std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> finalize(table_view const& keys,
                                                                              std::span<aggregation_request const> requests,
                                                                              stream_ref stream, rmm::device_async_resource_ref mr);
  hash_path path{hash_path::NONE};

 private:
  // ---- the plan (immutable after the constructor)
  planner_env const env;
  call_trace trace;
  host_plan hp;
  plan_dev const& p;
  int64_t const n;
  hipStream_t const s;
  int const RU, PU;
  agg_geom ag{};
  bool dense_signed{false}, dense_candidate{false}, dense_composite{false};
  // ---- what the sample pass found (estimate.cpp)
  double est_groups{0};
  uint64_t h_range[2]{0, 0};             // sample minimum / maximum of a single plain key column (bit patterns)
  int64_t h_ranges[2 * MAX_KU]{};        // composite keys: per key column, as int64
  bool ranges_known{false};
  double adjacent_equal{0.0};            // share of the sampled rows whose successor row carries the same key
  std::vector<uint64_t> hot_keys;        // heavy hitters (aggregated inside the scatter workgroups)
  double hot_mass{0.0};                  // their share of the sampled rows
  double skew_m2{0.0};                   // sum of the squared row shares of the other keys (0: not measured)
  // ---- state that attempts change
  scratch sc;
  int32_t* d_overflow{nullptr};
  bool allow_dense{false}, allow_optimistic{true}, pre_failed{false}, counted_all{false};
  double safety{1.3};
  // ---- result of the successful attempt
  uint64_t* partial{nullptr};  // partial records: item i at [i*cap, i*cap + count[i])
  int32_t* d_count{nullptr};
  int32_t nitems{0};
  int64_t final_cap{0};        // records per work item in `partial` (0: ag.cap)
  std::vector<int32_t> h_count;

  struct attempt_plan {  // per attempt: the tables the estimate asks for
    double need;         // hash tables needed at the planned load
    bool fits_one_table;
    agg_args aa;         // plan + geometry + overflow flag, the rest filled by the path
  };

  // estimate.cpp
  void estimate();
  int32_t overflow_and_counts();  // one stream synchronisation returns the overflow flag and the per-item group counts
  void escalate();
  void fresh_scratch();           // drops the attempt's temporaries (a path ruled itself out)
  bool dense_map_from_sample(dense_map& dm, bool tight = false) const;
  attempt_plan plan_attempt() const;
  outcome run_attempt(int attempt);
  // paths_dense.cpp
  outcome try_dense_one_table();
  outcome try_dense_ring();
  outcome try_dense_ring_multi();
  outcome try_dense_wc();
  // paths_hash.cpp
  outcome run_single_pass(attempt_plan& ap);
  outcome try_preaggregate(attempt_plan& ap);
  outcome run_partitioned(attempt_plan& ap);
  struct partition_plan {
    int64_t P1, P2;
    int log2P1, log2P2;
  };
  outcome try_optimistic_one_level(attempt_plan& ap, partition_plan const& pp, part_args& pa, part_args* d_pa);
  outcome try_optimistic_two_level(attempt_plan& ap, partition_plan const& pp, part_args& pa, part_args* d_pa);
  int32_t wc_granule_for(int64_t P) const;
  bool setup_hot(part_args& pa, int64_t P);
  void merge_hot(part_args const& pa, agg_args const& aa);
  void exact_pipeline(part_args& pa, part_args* d_pa, plan_dev const& pplan, int64_t nrec, int units, agg_input in_mode, int64_t P1,
                      int64_t P2, int log2P1, int log2P2, agg_args& aa);
};

}  // namespace cudf::groupby::detail
