// SPDX-License-Identifier: Apache-2.0
// extern "C" shim over the C++ boundary (include/cudf_amd_c.h). Translates views, maps exceptions to status
// codes (pylibcudf does the same mapping in exception_handler.pxd:34-64).
#include <cudf_amd_c.h>

#include "common/profiler.hpp"

#include <cudf/copying.hpp>
#include <cudf/distributed.hpp>
#include <cudf/groupby.hpp>
#include <cudf/partitioning.hpp>
#include <cudf/interop.hpp>
#include <cudf/join/hash_join.hpp>
#include <cudf/join/join.hpp>
#include <cudf/table/table.hpp>
#include <cudf/utilities/error.hpp>

#include <algorithm>
#include <cstring>
#include <new>
#include <mutex>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>

struct cudf_amd_table_s {
  std::vector<std::unique_ptr<cudf::column>> cols;
};
struct cudf_amd_comm_s {
  std::unique_ptr<cudf::distributed::communicator> comm;
};
struct cudf_amd_hash_join_s {
  std::unique_ptr<cudf::hash_join> hj;
};

namespace {
thread_local std::string g_last_error;

rmm::mr::statistics_resource_adaptor& stats_mr()
{
  static rmm::mr::pool_memory_resource base{};
  static rmm::mr::statistics_resource_adaptor stats{&base};
  return stats;
}
void ensure_resource()
{
  static bool const once = [] {
    rmm::mr::set_current_device_resource(&stats_mr());
    return true;
  }();
  (void)once;
}

template <typename F>
cudf_amd_status guarded(F&& f)
{
  try {
    ensure_resource();
    f();
    return CUDF_AMD_OK;
  } catch (cudf::data_type_error const& e) {
    g_last_error = e.what();
    return CUDF_AMD_DATA_TYPE_ERROR;
  } catch (std::invalid_argument const& e) {
    g_last_error = e.what();
    return CUDF_AMD_INVALID_ARGUMENT;
  } catch (std::out_of_range const& e) {
    g_last_error = e.what();
    return CUDF_AMD_OUT_OF_RANGE;
  } catch (cudf::logic_error const& e) {
    g_last_error = e.what();
    return CUDF_AMD_LOGIC_ERROR;
  } catch (cudf::hip_error const& e) {
    g_last_error = e.what();
    return CUDF_AMD_DEVICE_ERROR;
  } catch (std::bad_alloc const& e) {
    g_last_error = "std::bad_alloc: out of device memory";
    return CUDF_AMD_BAD_ALLOC;
  } catch (std::exception const& e) {
    g_last_error = e.what();
    return CUDF_AMD_OTHER_ERROR;
  }
}

cudf::column_view to_view(cudf_amd_column_view const& c)
{
  auto const id = static_cast<cudf::type_id>(c.type_id);
  cudf::data_type const t =
    (id == cudf::type_id::DECIMAL32 || id == cudf::type_id::DECIMAL64 || id == cudf::type_id::DECIMAL128)
      ? cudf::data_type{id, c.scale}
      : cudf::data_type{id};
  return cudf::column_view{t, c.size, c.data, c.null_mask, c.null_count, c.offset};
}
cudf::table_view to_table(cudf_amd_column_view const* cols, int32_t n)
{
  std::vector<cudf::column_view> v;
  v.reserve(n);
  for (int32_t i = 0; i < n; ++i) v.push_back(to_view(cols[i]));
  return cudf::table_view{v};
}
hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

std::unique_ptr<cudf::groupby_aggregation> make_agg(cudf_amd_aggregation_request const& req, int32_t k)
{
  int32_t const kind    = req.kinds[k];
  int32_t const* param  = req.params ? req.params + k : nullptr;
  int32_t const* param2 = req.params2 ? req.params2 + k : nullptr;
  auto const policy     = [&](cudf::null_policy dflt) { return param2 ? (*param2 ? cudf::null_policy::INCLUDE : cudf::null_policy::EXCLUDE) : dflt; };
  using A = cudf::aggregation;
  switch (kind) {
    case A::SUM: return cudf::make_sum_aggregation<cudf::groupby_aggregation>();
    case A::SUM_OVERFLOW: return cudf::make_sum_overflow_aggregation<cudf::groupby_aggregation>();
    case A::PRODUCT: return cudf::make_product_aggregation<cudf::groupby_aggregation>();
    case A::MIN: return cudf::make_min_aggregation<cudf::groupby_aggregation>();
    case A::MAX: return cudf::make_max_aggregation<cudf::groupby_aggregation>();
    case A::COUNT_VALID: return cudf::make_count_aggregation<cudf::groupby_aggregation>(cudf::null_policy::EXCLUDE);
    case A::COUNT_ALL: return cudf::make_count_aggregation<cudf::groupby_aggregation>(cudf::null_policy::INCLUDE);
    case A::SUM_OF_SQUARES: return cudf::make_sum_of_squares_aggregation<cudf::groupby_aggregation>();
    case A::MEAN: return cudf::make_mean_aggregation<cudf::groupby_aggregation>();
    case A::M2: return cudf::make_m2_aggregation<cudf::groupby_aggregation>();
    case A::VARIANCE: return cudf::make_variance_aggregation<cudf::groupby_aggregation>(param ? *param : 1);
    case A::STD: return cudf::make_std_aggregation<cudf::groupby_aggregation>(param ? *param : 1);
    case A::ARGMAX: return cudf::make_argmax_aggregation<cudf::groupby_aggregation>();
    case A::ARGMIN: return cudf::make_argmin_aggregation<cudf::groupby_aggregation>();
    case A::MEDIAN: return cudf::make_median_aggregation<cudf::groupby_aggregation>();
    case A::NTH_ELEMENT:
      return cudf::make_nth_element_aggregation<cudf::groupby_aggregation>(param ? *param : 0, policy(cudf::null_policy::INCLUDE));
    case A::NUNIQUE: return cudf::make_nunique_aggregation<cudf::groupby_aggregation>(policy(cudf::null_policy::EXCLUDE));
    case A::QUANTILE: {
      CUDF_EXPECTS(req.quantiles != nullptr && req.quantile_offsets != nullptr, "QUANTILE needs quantiles and quantile_offsets.",
                   std::invalid_argument);
      CUDF_EXPECTS(!param2 || (*param2 >= 0 && *param2 <= 5), "QUANTILE: unknown interpolation.", std::invalid_argument);
      std::vector<double> q(req.quantiles + req.quantile_offsets[k], req.quantiles + req.quantile_offsets[k + 1]);
      return cudf::make_quantile_aggregation<cudf::groupby_aggregation>(q, param2 ? static_cast<cudf::interpolation>(*param2)
                                                                                  : cudf::interpolation::LINEAR);
    }
    default: CUDF_FAIL("Unsupported aggregation kind in the C ABI.", std::invalid_argument);
  }
}
}  // namespace

extern "C" {

const char* cudf_amd_last_error(void) { return g_last_error.c_str(); }
const char* cudf_amd_version(void) { return "cudf_amd 0.3.0 (gfx950; libcudf 26.10 API subset)"; }
// Bumped whenever a struct layout or the meaning of an argument of this header changes (2: cudf_amd_aggregation_request gained
// `params`, cudf_amd_hash_partition writes num_partitions + 1 offsets; 3: loopback communicators, shuffle_join; 4: the sort-groupby kinds' parameters).
int32_t cudf_amd_abi_version(void) { return CUDF_AMD_ABI_VERSION; }

// sizes of the live cudf_amd_malloc allocations: the resource's deallocate takes the size, the C ABI's free does not
static std::mutex g_alloc_mu;
static std::unordered_map<void*, std::size_t> g_alloc_sizes;
cudf_amd_status cudf_amd_malloc(void** ptr, size_t bytes, void* stream)
{
  return guarded([&] {
    *ptr = cudf::get_current_device_resource_ref().allocate_async(bytes, as_stream(stream));
    std::lock_guard<std::mutex> g{g_alloc_mu};
    g_alloc_sizes[*ptr] = bytes;
  });
}
cudf_amd_status cudf_amd_free(void* ptr, void* stream)
{
  return guarded([&] {
    if (ptr == nullptr) return;
    std::size_t bytes = 0;
    {
      std::lock_guard<std::mutex> g{g_alloc_mu};
      auto it = g_alloc_sizes.find(ptr);
      CUDF_EXPECTS(it != g_alloc_sizes.end(), "cudf_amd_free: pointer was not allocated by cudf_amd_malloc", std::invalid_argument);
      bytes = it->second;
      g_alloc_sizes.erase(it);
    }
    cudf::get_current_device_resource_ref().deallocate_async(ptr, bytes, as_stream(stream));
  });
}
cudf_amd_status cudf_amd_memcpy(void* dst, const void* src, size_t bytes, int32_t kind, void* stream)
{
  return guarded([&] {
    auto const k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (bytes) CUDF_HIP_TRY(hipMemcpyAsync(dst, src, bytes, k, as_stream(stream)));
  });
}
cudf_amd_status cudf_amd_memset(void* dst, int32_t value, size_t bytes, void* stream)
{
  return guarded([&] {
    if (bytes) CUDF_HIP_TRY(hipMemsetAsync(dst, value, bytes, as_stream(stream)));
  });
}
cudf_amd_status cudf_amd_stream_synchronize(void* stream)
{
  return guarded([&] { CUDF_HIP_TRY(hipStreamSynchronize(as_stream(stream))); });
}
cudf_amd_status cudf_amd_memory_stats(uint64_t* current_bytes, uint64_t* peak_bytes)
{
  return guarded([&] {
    *current_bytes = stats_mr().current_bytes();
    *peak_bytes    = stats_mr().peak_bytes();
  });
}

cudf_amd_status cudf_amd_profile_enable(int32_t on)
{
  return guarded([&] { cudf::detail::prof::enable(on != 0); });
}
cudf_amd_status cudf_amd_profile_reset(void)
{
  return guarded([&] { cudf::detail::prof::reset(); });
}
cudf_amd_status cudf_amd_profile_report(char* buf, size_t n)
{
  return guarded([&] {
    std::string out;
    for (auto const& k : cudf::detail::prof::collect())
      out += k.name + " " + std::to_string(k.launches) + " " + std::to_string(k.total_ms) + "\n";
    if (n > 0) {
      auto const len = std::min(out.size(), n - 1);
      std::memcpy(buf, out.data(), len);
      buf[len] = 0;
    }
  });
}

int32_t cudf_amd_table_num_columns(cudf_amd_table_t t) { return t ? static_cast<int32_t>(t->cols.size()) : 0; }
int32_t cudf_amd_table_num_rows(cudf_amd_table_t t) { return (t && !t->cols.empty()) ? t->cols.front()->size() : 0; }
cudf_amd_status cudf_amd_table_column(cudf_amd_table_t t, int32_t i, cudf_amd_column_view* out)
{
  return guarded([&] {
    CUDF_EXPECTS(t != nullptr && i >= 0 && i < static_cast<int32_t>(t->cols.size()), "column index out of range",
                 std::invalid_argument);
    auto const v    = t->cols[i]->view();
    out->type_id    = static_cast<int32_t>(v.type().id());
    out->size       = v.size();
    out->data       = v.head();
    out->null_mask  = v.null_mask();
    out->null_count = v.null_count();
    out->offset     = v.offset();
    out->scale      = v.type().scale();
  });
}
static void fill_view(cudf::column_view const& v, cudf_amd_column_view* out)
{
  out->type_id    = static_cast<int32_t>(v.type().id());
  out->size       = v.size();
  out->data       = v.head();
  out->null_mask  = v.null_mask();
  out->null_count = v.null_count();
  out->offset     = v.offset();
  out->scale      = v.type().scale();
}
int32_t cudf_amd_table_column_num_children(cudf_amd_table_t t, int32_t i)
{
  return (t && i >= 0 && i < static_cast<int32_t>(t->cols.size())) ? t->cols[i]->num_children() : 0;
}
cudf_amd_status cudf_amd_table_column_child(cudf_amd_table_t t, int32_t i, int32_t j, cudf_amd_column_view* out)
{
  return guarded([&] {
    CUDF_EXPECTS(t != nullptr && i >= 0 && i < static_cast<int32_t>(t->cols.size()), "column index out of range", std::invalid_argument);
    auto const v = t->cols[i]->view();
    CUDF_EXPECTS(j >= 0 && j < v.num_children(), "child index out of range", std::invalid_argument);
    fill_view(v.child(j), out);
  });
}
void cudf_amd_table_free(cudf_amd_table_t t) { delete t; }

cudf_amd_status cudf_amd_groupby_aggregate(const cudf_amd_column_view* keys, int32_t num_keys,
                                           int32_t include_null_keys, int32_t keys_are_sorted,
                                           const cudf_amd_aggregation_request* requests, int32_t num_requests,
                                           void* stream, cudf_amd_table_t* out_keys, cudf_amd_table_t* out_results,
                                           int32_t* out_path)
{
  return guarded([&] {
    *out_keys    = nullptr;
    *out_results = nullptr;
    auto const kt = to_table(keys, num_keys);
    std::vector<cudf::groupby::aggregation_request> reqs(num_requests);
    for (int32_t r = 0; r < num_requests; ++r) {
      reqs[r].values = to_view(requests[r].values);
      for (int32_t k = 0; k < requests[r].num_kinds; ++k)
        reqs[r].aggregations.push_back(make_agg(requests[r], k));
    }
    cudf::groupby::groupby gb{kt, include_null_keys ? cudf::null_policy::INCLUDE : cudf::null_policy::EXCLUDE,
                              keys_are_sorted ? cudf::sorted::YES : cudf::sorted::NO};
    auto [ukeys, results] = gb.aggregate(reqs, cudf::stream_ref{as_stream(stream)});
    auto kh               = std::make_unique<cudf_amd_table_s>();
    kh->cols              = ukeys->release();
    auto rh               = std::make_unique<cudf_amd_table_s>();
    for (auto& r : results)
      for (auto& c : r.results) rh->cols.push_back(std::move(c));
    if (out_path) *out_path = static_cast<int32_t>(gb.last_path());
    *out_keys    = kh.release();
    *out_results = rh.release();
  });
}

static cudf_amd_table_t indices_to_table(cudf::join_index_pair&& p)
{
  auto t = std::make_unique<cudf_amd_table_s>();
  t->cols.push_back(std::make_unique<cudf::column>(std::move(*p.first), rmm::device_buffer{}, 0));
  t->cols.push_back(std::make_unique<cudf::column>(std::move(*p.second), rmm::device_buffer{}, 0));
  return t.release();
}

cudf_amd_status cudf_amd_join(const cudf_amd_column_view* left_keys, int32_t num_left,
                              const cudf_amd_column_view* right_keys, int32_t num_right, int32_t nulls_equal,
                              int32_t kind, void* stream, cudf_amd_table_t* out_indices)
{
  return guarded([&] {
    *out_indices  = nullptr;
    auto const l  = to_table(left_keys, num_left);
    auto const r  = to_table(right_keys, num_right);
    auto const eq = nulls_equal ? cudf::null_equality::EQUAL : cudf::null_equality::UNEQUAL;
    cudf::stream_ref const s{as_stream(stream)};
    CUDF_EXPECTS(kind >= 0 && kind <= 2, "join kind must be 0 (inner), 1 (left) or 2 (full)", std::invalid_argument);
    auto p = kind == 0 ? cudf::inner_join(l, r, eq, s) : kind == 1 ? cudf::left_join(l, r, eq, s) : cudf::full_join(l, r, eq, s);
    *out_indices = indices_to_table(std::move(p));
  });
}

cudf_amd_status cudf_amd_hash_join_create(const cudf_amd_column_view* right_keys, int32_t num_right, int32_t has_nulls,
                                          int32_t nulls_equal, double load_factor, void* stream,
                                          cudf_amd_hash_join_t* out)
{
  return guarded([&] {
    *out          = nullptr;
    auto const r  = to_table(right_keys, num_right);
    auto const eq = nulls_equal ? cudf::null_equality::EQUAL : cudf::null_equality::UNEQUAL;
    cudf::stream_ref const s{as_stream(stream)};
    auto h = std::make_unique<cudf_amd_hash_join_s>();
    if (has_nulls < 0) h->hj = std::make_unique<cudf::hash_join>(r, eq, s);
    else h->hj = std::make_unique<cudf::hash_join>(r, has_nulls ? cudf::nullable_join::YES : cudf::nullable_join::NO, eq, load_factor, s);
    *out = h.release();
  });
}
void cudf_amd_hash_join_destroy(cudf_amd_hash_join_t h) { delete h; }

cudf_amd_status cudf_amd_hash_join_probe(cudf_amd_hash_join_t h, const cudf_amd_column_view* left_keys, int32_t num_left,
                                         int32_t kind, int64_t output_size, void* stream, cudf_amd_table_t* out_indices)
{
  return guarded([&] {
    *out_indices = nullptr;
    CUDF_EXPECTS(h != nullptr && h->hj != nullptr, "null hash_join handle", std::invalid_argument);
    auto const l = to_table(left_keys, num_left);
    cudf::stream_ref const s{as_stream(stream)};
    std::optional<std::size_t> sz;
    if (output_size >= 0) sz = static_cast<std::size_t>(output_size);
    CUDF_EXPECTS(kind >= 0 && kind <= 2, "join kind must be 0 (inner), 1 (left) or 2 (full)", std::invalid_argument);
    auto p = kind == 0 ? h->hj->inner_join(l, sz, s) : kind == 1 ? h->hj->left_join(l, sz, s) : h->hj->full_join(l, sz, s);
    *out_indices = indices_to_table(std::move(p));
  });
}

cudf_amd_status cudf_amd_hash_join_size(cudf_amd_hash_join_t h, const cudf_amd_column_view* left_keys, int32_t num_left,
                                        int32_t kind, void* stream, uint64_t* out_size)
{
  return guarded([&] {
    CUDF_EXPECTS(h != nullptr && h->hj != nullptr, "null hash_join handle", std::invalid_argument);
    auto const l = to_table(left_keys, num_left);
    cudf::stream_ref const s{as_stream(stream)};
    CUDF_EXPECTS(kind >= 0 && kind <= 2, "join kind must be 0 (inner), 1 (left) or 2 (full)", std::invalid_argument);
    *out_size = kind == 0 ? h->hj->inner_join_size(l, s) : kind == 1 ? h->hj->left_join_size(l, s) : h->hj->full_join_size(l, s);
  });
}

cudf_amd_status cudf_amd_hash_join_match_counts(cudf_amd_hash_join_t h, const cudf_amd_column_view* left_keys,
                                                int32_t num_left, int32_t kind, void* stream, cudf_amd_table_t* out_counts)
{
  return guarded([&] {
    *out_counts = nullptr;
    CUDF_EXPECTS(h != nullptr && h->hj != nullptr, "null hash_join handle", std::invalid_argument);
    CUDF_EXPECTS(kind >= 0 && kind <= 2, "join kind must be 0 (inner), 1 (left) or 2 (full)", std::invalid_argument);
    auto const l = to_table(left_keys, num_left);
    cudf::stream_ref const s{as_stream(stream)};
    auto ctx = kind == 0 ? h->hj->inner_join_match_context(l, s)
                         : kind == 1 ? h->hj->left_join_match_context(l, s) : h->hj->full_join_match_context(l, s);
    auto t = std::make_unique<cudf_amd_table_s>();
    t->cols.push_back(std::make_unique<cudf::column>(std::move(*ctx._match_counts), rmm::device_buffer{}, 0));
    *out_counts = t.release();
  });
}

cudf_amd_status cudf_amd_hash_join_probe_range(cudf_amd_hash_join_t h, const cudf_amd_column_view* left_keys,
                                               int32_t num_left, const int32_t* match_counts, int32_t kind,
                                               int32_t left_start, int32_t left_end, void* stream,
                                               cudf_amd_table_t* out_indices)
{
  return guarded([&] {
    *out_indices = nullptr;
    CUDF_EXPECTS(h != nullptr && h->hj != nullptr, "null hash_join handle", std::invalid_argument);
    CUDF_EXPECTS(kind >= 0 && kind <= 2, "join kind must be 0 (inner), 1 (left) or 2 (full)", std::invalid_argument);
    auto const l = to_table(left_keys, num_left);
    cudf::stream_ref const s{as_stream(stream)};
    cudf::join_partition_context ctx{nullptr, left_start, left_end};
    if (match_counts != nullptr) {
      // the context only has to carry the counts the caller already holds: a non-owning copy is not expressible
      // with device_uvector, so the counts are duplicated (4 bytes per left row)
      auto counts = std::make_unique<rmm::device_uvector<cudf::size_type>>(static_cast<std::size_t>(l.num_rows()), s.value(),
                                                                          cudf::get_current_device_resource_ref());
      if (l.num_rows() > 0)
        CUDF_HIP_TRY(hipMemcpyAsync(counts->data(), match_counts, static_cast<std::size_t>(l.num_rows()) * sizeof(cudf::size_type),
                                    hipMemcpyDeviceToDevice, s.value()));
      ctx.left_table_context = std::make_unique<cudf::join_match_context>(l, std::move(counts));
    }
    auto p = kind == 0 ? h->hj->partitioned_inner_join(ctx, s)
                       : kind == 1 ? h->hj->partitioned_left_join(ctx, s) : h->hj->partitioned_full_join(ctx, s);
    *out_indices = indices_to_table(std::move(p));
  });
}

cudf_amd_status cudf_amd_hash_join_finalize_full(const int32_t* const* left_partials, const int32_t* const* right_partials,
                                                 const uint64_t* partial_sizes, int32_t num_partials,
                                                 int32_t left_num_rows, int32_t right_num_rows, void* stream,
                                                 cudf_amd_table_t* out_indices)
{
  return guarded([&] {
    *out_indices = nullptr;
    CUDF_EXPECTS(num_partials >= 0, "negative number of partial results", std::invalid_argument);
    std::vector<cudf::device_span<cudf::size_type const>> lp, rp;
    for (int32_t i = 0; i < num_partials; ++i) {
      lp.emplace_back(left_partials[i], static_cast<std::size_t>(partial_sizes[i]));
      rp.emplace_back(right_partials[i], static_cast<std::size_t>(partial_sizes[i]));
    }
    cudf::stream_ref const s{as_stream(stream)};
    *out_indices = indices_to_table(cudf::hash_join::finalize_partitioned_full_join(lp, rp, left_num_rows, right_num_rows, s));
  });
}

cudf_amd_status cudf_amd_from_arrow(const struct ArrowSchema* schema, const struct ArrowArray* array, void* stream,
                                    cudf_amd_table_t* out_table)
{
  return guarded([&] {
    *out_table = nullptr;
    cudf::stream_ref const s{as_stream(stream)};
    auto tbl = cudf::from_arrow(schema, array, s);
    auto t   = std::make_unique<cudf_amd_table_s>();
    t->cols  = tbl->release();
    *out_table = t.release();
  });
}

cudf_amd_status cudf_amd_to_arrow_host(const cudf_amd_column_view* columns, int32_t num_columns, const char* const* names,
                                       void* stream, struct ArrowSchema* out_schema, struct ArrowArray* out_array)
{
  return guarded([&] {
    CUDF_EXPECTS(out_schema != nullptr && out_array != nullptr, "output ArrowSchema / ArrowArray must not be NULL",
                 std::invalid_argument);
    auto const tv = to_table(columns, num_columns);
    std::vector<cudf::column_metadata> meta;
    for (int32_t i = 0; i < num_columns; ++i) meta.emplace_back(names != nullptr && names[i] != nullptr ? names[i] : "");
    cudf::stream_ref const s{as_stream(stream)};
    auto schema = cudf::to_arrow_schema(tv, meta);
    auto arr    = cudf::to_arrow_host(tv, s);
    *out_schema = *schema;          // move: the caller's copy now owns the contents
    schema->release = nullptr;
    *out_array  = arr->array;
    arr->array.release = nullptr;
  });
}

cudf_amd_status cudf_amd_hash_partition(const cudf_amd_column_view* input, int32_t num_columns,
                                        const int32_t* columns_to_hash, int32_t num_hash_columns, int32_t num_partitions,
                                        uint32_t seed, void* stream, cudf_amd_table_t* out_table, int32_t* out_offsets)
{
  return guarded([&] {
    *out_table   = nullptr;
    auto const t = to_table(input, num_columns);
    std::vector<cudf::size_type> cols(columns_to_hash, columns_to_hash + num_hash_columns);
    auto [tbl, offs] = cudf::hash_partition(t, cols, num_partitions, cudf::hash_id::HASH_MURMUR3, seed,
                                            cudf::stream_ref{as_stream(stream)});
    for (size_t i = 0; i < offs.size(); ++i) out_offsets[i] = offs[i];
    auto h  = std::make_unique<cudf_amd_table_s>();
    h->cols = tbl->release();
    *out_table = h.release();
  });
}

cudf_amd_status cudf_amd_comm_unique_id(uint8_t* out_id_128_bytes)
{
  return guarded([&] {
    auto const id = cudf::distributed::communicator::make_unique_id();
    std::memcpy(out_id_128_bytes, id.data(), id.size());
  });
}
cudf_amd_status cudf_amd_comm_create(const uint8_t* id_128_bytes, int32_t world_size, int32_t rank, cudf_amd_comm_t* out)
{
  return guarded([&] {
    *out = nullptr;
    cudf::distributed::unique_id id{};
    std::memcpy(id.data(), id_128_bytes, id.size());
    auto h  = std::make_unique<cudf_amd_comm_s>();
    h->comm = std::make_unique<cudf::distributed::communicator>(id, world_size, rank);
    *out    = h.release();
  });
}
void cudf_amd_comm_destroy(cudf_amd_comm_t comm) { delete comm; }
cudf_amd_status cudf_amd_comm_create_loopback(int32_t world_size, cudf_amd_comm_t* out_comms)
{
  return guarded([&] {
    for (int32_t r = 0; r < world_size; ++r) out_comms[r] = nullptr;
    auto ends = cudf::distributed::communicator::make_loopback(world_size);
    for (int32_t r = 0; r < world_size; ++r) {
      auto h        = std::make_unique<cudf_amd_comm_s>();
      h->comm       = std::move(ends[r]);
      out_comms[r]  = h.release();
    }
  });
}
cudf_amd_status cudf_amd_comm_set_max_message_bytes(cudf_amd_comm_t comm, int64_t bytes)
{
  return guarded([&] {
    CUDF_EXPECTS(comm != nullptr && comm->comm != nullptr, "null communicator", std::invalid_argument);
    comm->comm->set_max_message_bytes(bytes);
  });
}
cudf_amd_status cudf_amd_plan_exchange(const int64_t* counts, int32_t world_size, int32_t rank, int64_t* out_recv_count,
                                       int64_t* out_recv_offset, int64_t* out_biggest)
{
  // (host arithmetic only: no device resource is touched, so it runs on a box without a GPU)
  try {
    CUDF_EXPECTS(world_size >= 1 && counts != nullptr, "plan_exchange: a world x world count matrix", std::invalid_argument);
    std::vector<int64_t> m(counts, counts + static_cast<std::size_t>(world_size) * world_size);
    auto const ep = cudf::distributed::plan_exchange(m, world_size, rank);
    for (int32_t p = 0; p < world_size; ++p) out_recv_count[p] = ep.recv_count[p];
    for (int32_t p = 0; p <= world_size; ++p) out_recv_offset[p] = ep.recv_offset[p];
    *out_biggest = ep.biggest;
    return CUDF_AMD_OK;
  } catch (std::invalid_argument const& e) {
    g_last_error = e.what();
    return CUDF_AMD_INVALID_ARGUMENT;
  } catch (std::exception const& e) {
    g_last_error = e.what();
    return CUDF_AMD_OTHER_ERROR;
  }
}
cudf_amd_status cudf_amd_shuffle_join(cudf_amd_comm_t comm, const cudf_amd_column_view* left_keys, int32_t num_left,
                                      const cudf_amd_column_view* right_keys, int32_t num_right, int32_t nulls_equal, void* stream,
                                      cudf_amd_table_t* out_global_row_ids)
{
  return guarded([&] {
    *out_global_row_ids = nullptr;
    CUDF_EXPECTS(comm != nullptr && comm->comm != nullptr, "null communicator", std::invalid_argument);
    auto [l, r] = cudf::distributed::shuffle_join(to_table(left_keys, num_left), to_table(right_keys, num_right), *comm->comm,
                                                  nulls_equal ? cudf::null_equality::EQUAL : cudf::null_equality::UNEQUAL,
                                                  cudf::stream_ref{as_stream(stream)});
    auto h = std::make_unique<cudf_amd_table_s>();
    h->cols.push_back(std::move(l));
    h->cols.push_back(std::move(r));
    *out_global_row_ids = h.release();
  });
}
cudf_amd_status cudf_amd_range_partition(const cudf_amd_column_view* input, int32_t num_columns, const int32_t* key_columns,
                                         int32_t num_key_columns, int32_t num_destinations, void* stream,
                                         cudf_amd_table_t* out_table, int32_t* out_offsets)
{
  return guarded([&] {
    *out_table = nullptr;
    std::vector<cudf::size_type> keys(key_columns, key_columns + num_key_columns);
    auto [tbl, offs] = cudf::distributed::range_partition(to_table(input, num_columns), keys, num_destinations, cudf::stream_ref{as_stream(stream)});
    for (size_t i = 0; i < offs.size(); ++i) out_offsets[i] = offs[i];
    auto h  = std::make_unique<cudf_amd_table_s>();
    h->cols = tbl->release();
    *out_table = h.release();
  });
}
cudf_amd_status cudf_amd_shuffle(cudf_amd_comm_t comm, const cudf_amd_column_view* input, int32_t num_columns,
                                 const int32_t* key_columns, int32_t num_key_columns, void* stream, cudf_amd_table_t* out_table)
{
  return guarded([&] {
    *out_table = nullptr;
    CUDF_EXPECTS(comm != nullptr && comm->comm != nullptr, "null communicator", std::invalid_argument);
    std::vector<cudf::size_type> keys(key_columns, key_columns + num_key_columns);
    auto tbl = cudf::distributed::shuffle(to_table(input, num_columns), keys, *comm->comm, cudf::stream_ref{as_stream(stream)});
    auto h   = std::make_unique<cudf_amd_table_s>();
    h->cols  = tbl->release();
    *out_table = h.release();
  });
}
cudf_amd_status cudf_amd_shuffle_groupby(cudf_amd_comm_t comm, const cudf_amd_column_view* keys, int32_t num_keys,
                                         int32_t include_null_keys, const cudf_amd_aggregation_request* requests,
                                         int32_t num_requests, void* stream, cudf_amd_table_t* out_keys,
                                         cudf_amd_table_t* out_results)
{
  return guarded([&] {
    *out_keys    = nullptr;
    *out_results = nullptr;
    CUDF_EXPECTS(comm != nullptr && comm->comm != nullptr, "null communicator", std::invalid_argument);
    auto const kt = to_table(keys, num_keys);
    std::vector<cudf::groupby::aggregation_request> reqs(num_requests);
    for (int32_t r = 0; r < num_requests; ++r) {
      reqs[r].values = to_view(requests[r].values);
      for (int32_t k = 0; k < requests[r].num_kinds; ++k)
        reqs[r].aggregations.push_back(make_agg(requests[r], k));
    }
    auto [ukeys, results] = cudf::distributed::shuffle_groupby(
      kt, reqs, *comm->comm, include_null_keys ? cudf::null_policy::INCLUDE : cudf::null_policy::EXCLUDE, cudf::stream_ref{as_stream(stream)});
    auto kh  = std::make_unique<cudf_amd_table_s>();
    kh->cols = ukeys->release();
    auto rh  = std::make_unique<cudf_amd_table_s>();
    for (auto& r : results)
      for (auto& c : r.results) rh->cols.push_back(std::move(c));
    *out_keys    = kh.release();
    *out_results = rh.release();
  });
}

cudf_amd_status cudf_amd_combine_groupby(cudf_amd_comm_t comm, const cudf_amd_column_view* keys, int32_t num_keys,
                                         int32_t include_null_keys, const cudf_amd_aggregation_request* requests,
                                         int32_t num_requests, void* stream, cudf_amd_table_t* out_keys,
                                         cudf_amd_table_t* out_results)
{
  return guarded([&] {
    *out_keys    = nullptr;
    *out_results = nullptr;
    CUDF_EXPECTS(comm != nullptr && comm->comm != nullptr, "null communicator", std::invalid_argument);
    auto const kt = to_table(keys, num_keys);
    std::vector<cudf::groupby::aggregation_request> reqs(num_requests);
    for (int32_t r = 0; r < num_requests; ++r) {
      reqs[r].values = to_view(requests[r].values);
      for (int32_t k = 0; k < requests[r].num_kinds; ++k)
        reqs[r].aggregations.push_back(make_agg(requests[r], k));
    }
    auto [ukeys, results] = cudf::distributed::combine_groupby(
      kt, reqs, *comm->comm, include_null_keys ? cudf::null_policy::INCLUDE : cudf::null_policy::EXCLUDE, cudf::stream_ref{as_stream(stream)});
    auto kh  = std::make_unique<cudf_amd_table_s>();
    kh->cols = ukeys->release();
    auto rh  = std::make_unique<cudf_amd_table_s>();
    for (auto& r : results)
      for (auto& c : r.results) rh->cols.push_back(std::move(c));
    *out_keys    = kh.release();
    *out_results = rh.release();
  });
}

cudf_amd_status cudf_amd_murmurhash3_x86_32(const cudf_amd_column_view* input, int32_t num_columns, uint32_t seed,
                                            void* stream, cudf_amd_table_t* out_column)
{
  return guarded([&] {
    *out_column = nullptr;
    auto h      = std::make_unique<cudf_amd_table_s>();
    h->cols.push_back(cudf::hashing::murmurhash3_x86_32(to_table(input, num_columns), seed, cudf::stream_ref{as_stream(stream)}));
    *out_column = h.release();
  });
}

cudf_amd_status cudf_amd_gather(const cudf_amd_column_view* source, int32_t num_columns,
                                const cudf_amd_column_view* gather_map, int32_t nullify, void* stream,
                                cudf_amd_table_t* out_table)
{
  return guarded([&] {
    *out_table = nullptr;
    auto tbl   = cudf::gather(to_table(source, num_columns), to_view(*gather_map),
                              nullify ? cudf::out_of_bounds_policy::NULLIFY : cudf::out_of_bounds_policy::DONT_CHECK,
                              cudf::stream_ref{as_stream(stream)});
    auto h     = std::make_unique<cudf_amd_table_s>();
    h->cols    = tbl->release();
    *out_table = h.release();
  });
}
}  // extern "C"
