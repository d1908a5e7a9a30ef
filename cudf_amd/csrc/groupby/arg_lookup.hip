// SPDX-License-Identifier: Apache-2.0
// ARGMIN / ARGMAX over one integer key column of a small range, without the hash tables' second sweep.
// The hash engine answers these kinds by aggregating {key, value, row id} records in LDS hash tables and then walking every row a second
// time to find the first row that attains its group's extreme (aggregate_kernel.inl): 1B rows on 500K groups take 26 ms for ARGMIN and
// 37 ms for ARGMIN + ARGMAX, against 9.3 ms for MIN + MAX on the dense-key path (profiles/r4_exotic_shapes.txt, bench_micro/arg_probe.py).
// Here the request is answered in two steps instead:
//   1. the same call with every ARGMIN replaced by MIN and every ARGMAX by MAX (whatever path the engine picks: the dense one for such keys);
//   2. one streaming pass over the key and value columns: a row looks its group up in a direct-address table key -> group (4 bytes per
//      key of the range: L2-resident for the shapes this is for), compares its value with the group's extreme and, if it attains it,
//      lowers the group's row index with an atomic minimum - the SMALLEST row index among the rows that attain the extreme, as the engine's
//      sweep finds it (reference: ARGMIN / ARGMAX of groupby/hash, device_aggregators.cuh:309-361, ties by arrival order there).
// A group without a valid value is null (as its MIN / MAX is); a group whose valid values are all NaN attains no extreme: -1, the engine's
// rule (tests/test_groupby_gpu.py::test_all_nan_group_min_max_argmin_argmax_pinned).
#include "call.hpp"
#include "../common/profiler.hpp"

#include <cudf/groupby.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/utilities/error.hpp>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <limits>

namespace cudf::groupby::detail {
namespace {
using cudf::detail::CLS_BOOL;
using cudf::detail::CLS_F32;
using cudf::detail::CLS_F64;
using cudf::detail::CLS_SINT;
using cudf::detail::CLS_UINT;
using cudf::detail::col_is_valid;
using cudf::detail::col_load_bits;
using cudf::detail::device_column;
using cudf::detail::gload;
using cudf::detail::gstore;
namespace prof = cudf::detail::prof;

// an integer key as a uint64 whose unsigned order is the keys' order (signed: sign-extended, sign bit flipped)
__device__ __forceinline__ uint64_t ordered_key(device_column const& c, int64_t row)
{
  uint64_t const raw = col_load_bits(c, row);
  if (c.cls == CLS_SINT) {
    int const sh = 64 - 8 * c.width;
    return static_cast<uint64_t>(static_cast<int64_t>(raw << sh) >> sh) ^ 0x8000000000000000ull;
  }
  return c.cls == CLS_BOOL ? static_cast<uint64_t>(raw != 0) : raw;
}

// out[0] = min, out[1] = max of the ordered keys of rows 0, stride, 2 * stride ... (valid ones)
__global__ void __launch_bounds__(256) k_key_range(device_column col, int64_t nrows, int64_t stride, unsigned long long* out)
{
  int64_t const r = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * stride;
  unsigned long long lo = ~0ull, hi = 0ull;
  if (r < nrows && col_is_valid(col, r)) lo = hi = ordered_key(col, r);
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long const l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0 && lo <= hi) {
    atomicMin(out, lo);
    atomicMax(out + 1, hi);
  }
}
__global__ void k_init_range(unsigned long long* out)
{
  out[0] = ~0ull;
  out[1] = 0ull;
}

__global__ void __launch_bounds__(256) k_fill_i32(int32_t* out, int64_t n, int32_t v)
{
  int64_t const i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) gstore(out + i, v);
}

// One pass per values column. Tables BY KEY SLOT (slot = ordered key - lo; slot `range` = the NULL key): best[slot * NE + e] = the raw bits
// of the group's extreme e, index[slot * NE + e] = the smallest row seen to attain it (preset INT32_MAX). One lookup per row and extreme -
// the first form went key -> group -> extreme (two dependent L2 round trips per row: 14.8 ms per extreme at 1B rows).
struct arg_pass {
  device_column keys, values;
  device_column ukeys;    // the unique keys of step 1 (G rows)
  device_column ext[2];   // the groups' MIN / MAX of `values` (a column of the values' type, one element per group); width 0: not asked for
  int32_t ne;             // extremes asked for (1 or 2): entries per slot
  uint64_t* best;
  int32_t* index;
  int32_t* out[2];        // per group: the answer (row index, or -1)
  uint64_t lo, range;
  int64_t nrows;
  int32_t G, keep_null_keys;
};

__device__ __forceinline__ bool attains(int cls, uint64_t v, uint64_t best)
{
  if (cls == CLS_F64) return __longlong_as_double(static_cast<long long>(v)) == __longlong_as_double(static_cast<long long>(best));
  if (cls == CLS_F32) return __uint_as_float(static_cast<uint32_t>(v)) == __uint_as_float(static_cast<uint32_t>(best));
  if (cls == CLS_BOOL) return (v != 0) == (best != 0);
  return v == best;
}

__device__ __forceinline__ uint64_t slot_of_group(arg_pass const& a, int32_t j)
{
  return col_is_valid(a.ukeys, j) ? ordered_key(a.ukeys, j) - a.lo : a.range;
}

__global__ void __launch_bounds__(256) k_arg_fill(arg_pass a)
{
  int32_t const j = blockIdx.x * 256 + threadIdx.x;
  if (j >= a.G) return;
  uint64_t const slot = slot_of_group(a, j);
  int q = 0;
  for (int e = 0; e < 2; ++e)
    if (a.ext[e].width != 0) gstore(a.best + slot * a.ne + q++, col_load_bits(a.ext[e], j));
}

// four rows per thread, their key / value loads issued together, then the lookups
template <int NE>
__global__ void __launch_bounds__(256) k_arg_rows(arg_pass a)
{
  constexpr int R = 4;
  int const cls        = a.values.cls;
  int64_t const stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t r0 = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; r0 < a.nrows; r0 += stride * R) {
    uint64_t key[R], val[R];
    bool live[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      int64_t const r = min(r0 + j * stride, a.nrows - 1);
      key[j]          = ordered_key(a.keys, r);
      val[j]          = col_load_bits(a.values, r);
      live[j]         = r0 + j * stride < a.nrows && col_is_valid(a.values, r);
    }
    uint64_t slot[R], b0[R], b1[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      int64_t const r = min(r0 + j * stride, a.nrows - 1);
      if (col_is_valid(a.keys, r)) {
        slot[j] = key[j] - a.lo;
        live[j] = live[j] && slot[j] < a.range;
      } else {
        slot[j] = a.range;
        live[j] = live[j] && a.keep_null_keys != 0;
      }
      slot[j] = live[j] ? slot[j] : 0;
      if constexpr (NE == 2) {
        cudf::detail::u64x2 const v = gload(reinterpret_cast<cudf::detail::u64x2 const*>(a.best) + slot[j]);
        b0[j] = v.x;
        b1[j] = v.y;
      } else {
        b0[j] = gload(a.best + slot[j]);
        b1[j] = 0;
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (!live[j]) continue;
      if (attains(cls, val[j], b0[j])) atomicMin(a.index + slot[j] * NE, static_cast<int32_t>(r0 + j * stride));
      if constexpr (NE == 2) {
        if (attains(cls, val[j], b1[j])) atomicMin(a.index + slot[j] * NE + 1, static_cast<int32_t>(r0 + j * stride));
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_arg_finish(arg_pass a)
{
  int32_t const j = blockIdx.x * 256 + threadIdx.x;
  if (j >= a.G) return;
  uint64_t const slot = slot_of_group(a, j);
  int q = 0;
  for (int e = 0; e < 2; ++e) {
    if (a.ext[e].width == 0) continue;
    int32_t const row = gload(a.index + slot * a.ne + q++);
    gstore(a.out[e] + j, row == std::numeric_limits<int32_t>::max() ? -1 : row);
  }
}

unsigned blocks_of(int64_t n) { return static_cast<unsigned>(std::max<int64_t>((n + 255) / 256, 1)); }
bool is_arg(aggregation::Kind k) { return k == aggregation::ARGMIN || k == aggregation::ARGMAX; }
}  // namespace

std::optional<std::pair<std::unique_ptr<table>, std::vector<aggregation_result>>> arg_by_lookup(table_view const& keys, null_policy include_null_keys,
                                                                                              std::span<aggregation_request const> requests,
                                                                                              stream_ref stream, rmm::device_async_resource_ref mr,
                                                                                              hash_path* path)
{
  int64_t const n = keys.num_rows();
  bool any        = false;
  for (auto const& r : requests)
    for (auto const& a : r.aggregations) any = any || is_arg(a->kind);
  if (!any || keys.num_columns() != 1 || n > std::numeric_limits<int32_t>::max()) return std::nullopt;
  auto const kcls = cudf::detail::class_of(keys.column(0).type().id());
  if (kcls != CLS_SINT && kcls != CLS_UINT && kcls != CLS_BOOL) return std::nullopt;
  // (the switches are read per call: the test suite flips them between calls of one process)
  planner_env const env = planner_env::load();
  char const* const sw  = std::getenv("CUDF_AMD_GB_ARG_LOOKUP");
  if ((sw != nullptr && std::atoi(sw) == 0) || n < env.big_min_rows) return std::nullopt;

  hipStream_t const s = stream.value();
  auto tmp            = get_current_device_resource_ref();
  auto const dkeys    = cudf::detail::make_device_column(keys.column(0));
  constexpr uint64_t MAX_RANGE = uint64_t{1} << 28;  // 1 GiB of group indices at most
  // a sample of the keys says whether the range can be small at all (the unique keys of step 1 give the exact range)
  rmm::device_buffer d_range{2 * sizeof(unsigned long long), s, tmp};
  auto* range = static_cast<unsigned long long*>(d_range.data());
  unsigned long long h_range[2];
  {
    int64_t const stride = std::max<int64_t>(1, n >> 16);
    hipLaunchKernelGGL(k_init_range, dim3(1), dim3(1), 0, s, range);
    hipLaunchKernelGGL(k_key_range, dim3(blocks_of((n + stride - 1) / stride)), dim3(256), 0, s, dkeys, n, stride, range);
    CUDF_HIP_TRY(hipMemcpyAsync(h_range, range, sizeof(h_range), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    if (h_range[0] > h_range[1] || h_range[1] - h_range[0] >= MAX_RANGE) return std::nullopt;
  }

  // ---- step 1: the same call with MIN / MAX in the place of ARGMIN / ARGMAX
  std::vector<aggregation_request> step1;
  for (auto const& r : requests) {
    aggregation_request q{r.values, {}};
    for (auto const& a : r.aggregations) {
      if (a->kind == aggregation::ARGMIN) {
        q.aggregations.push_back(make_min_aggregation<groupby_aggregation>());
      } else if (a->kind == aggregation::ARGMAX) {
        q.aggregations.push_back(make_max_aggregation<groupby_aggregation>());
      } else {
        auto clone       = a->clone();
        auto* as_groupby = dynamic_cast<groupby_aggregation*>(clone.get());
        CUDF_EXPECTS(as_groupby != nullptr, "not a groupby aggregation");
        clone.release();
        q.aggregations.emplace_back(as_groupby);
      }
    }
    step1.push_back(std::move(q));
  }
  cudf::groupby::groupby first{keys, include_null_keys};
  auto [ukeys, results] = first.aggregate(step1, stream, mr);
  if (path != nullptr) *path = first.last_path();
  int32_t const G       = ukeys->num_rows();
  if (G == 0) return std::nullopt;
  auto const dukeys = cudf::detail::make_device_column(ukeys->get_column(0).view());
  hipLaunchKernelGGL(k_init_range, dim3(1), dim3(1), 0, s, range);
  hipLaunchKernelGGL(k_key_range, dim3(blocks_of(G)), dim3(256), 0, s, dukeys, static_cast<int64_t>(G), int64_t{1}, range);
  CUDF_HIP_TRY(hipMemcpyAsync(h_range, range, sizeof(h_range), hipMemcpyDeviceToHost, s));
  CUDF_HIP_TRY(hipStreamSynchronize(s));
  bool const any_valid_key = h_range[0] <= h_range[1];
  uint64_t const lo = any_valid_key ? h_range[0] : 0, span_ = any_valid_key ? h_range[1] - h_range[0] + 1 : 1;
  if (span_ > MAX_RANGE) return std::nullopt;  // (the sample missed far-out keys: the engine's own ARGMIN / ARGMAX)

  // ---- step 2: one pass over the rows per values column
  for (std::size_t r = 0; r < requests.size(); ++r) {
    int slot[2] = {-1, -1};  // the first ARGMIN / ARGMAX of this request (repeats are copies)
    for (std::size_t j = 0; j < requests[r].aggregations.size(); ++j) {
      auto const k = requests[r].aggregations[j]->kind;
      if (k == aggregation::ARGMIN && slot[0] < 0) slot[0] = static_cast<int>(j);
      if (k == aggregation::ARGMAX && slot[1] < 0) slot[1] = static_cast<int>(j);
    }
    if (slot[0] < 0 && slot[1] < 0) continue;
    arg_pass a{};
    a.keys   = dkeys;
    a.values = cudf::detail::make_device_column(requests[r].values);
    a.ukeys  = dukeys;
    if (!requests[r].values.has_nulls()) a.values.mask = nullptr;
    if (!keys.column(0).has_nulls()) a.keys.mask = nullptr;
    if (!ukeys->get_column(0).has_nulls()) a.ukeys.mask = nullptr;
    a.ne             = (slot[0] >= 0) + (slot[1] >= 0);
    a.lo             = lo;
    a.range          = span_;
    a.nrows          = n;
    a.G              = G;
    a.keep_null_keys = include_null_keys == null_policy::INCLUDE ? 1 : 0;
    std::size_t const entries = (static_cast<std::size_t>(span_) + 1) * static_cast<std::size_t>(a.ne);
    rmm::device_buffer d_best{entries * sizeof(uint64_t), s, tmp}, d_index{entries * sizeof(int32_t), s, tmp};
    a.best  = static_cast<uint64_t*>(d_best.data());
    a.index = static_cast<int32_t*>(d_index.data());
    std::unique_ptr<column> index[2];
    for (int e = 0; e < 2; ++e) {
      if (slot[e] < 0) continue;
      auto const ext = results[r].results[static_cast<std::size_t>(slot[e])]->view();  // MIN / MAX of the values, one per group
      a.ext[e]       = cudf::detail::make_device_column(ext);
      index[e]       = std::make_unique<column>(data_type{type_id::INT32}, G, rmm::device_buffer{static_cast<std::size_t>(G) * sizeof(int32_t), s, mr},
                                          rmm::device_buffer{}, 0);
      a.out[e]       = index[e]->mutable_view().data<int32_t>();
    }
    hipLaunchKernelGGL(k_fill_i32, dim3(blocks_of(static_cast<int64_t>(entries))), dim3(256), 0, s, a.index, static_cast<int64_t>(entries),
                       std::numeric_limits<int32_t>::max());
    CUDF_HIP_TRY(hipMemsetAsync(a.best, 0, entries * sizeof(uint64_t), s));
    hipLaunchKernelGGL(k_arg_fill, dim3(blocks_of(G)), dim3(256), 0, s, a);
    {
      prof::scope p_{"arg_rows", s};
      unsigned const grid = static_cast<unsigned>(std::clamp<int64_t>((n + 1023) / 1024, 1, 256 * 32));
      if (a.ne == 2) hipLaunchKernelGGL(k_arg_rows<2>, dim3(grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL(k_arg_rows<1>, dim3(grid), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(k_arg_finish, dim3(blocks_of(G)), dim3(256), 0, s, a);
    CUDF_HIP_TRY(hipGetLastError());
    for (int e = 0; e < 2; ++e) {
      if (slot[e] < 0) continue;
      // validity: the group has a valid value at all = its MIN / MAX is valid
      auto const ext = results[r].results[static_cast<std::size_t>(slot[e])]->view();
      if (ext.nullable() && ext.null_count() > 0)
        index[e]->set_null_mask(rmm::device_buffer{ext.null_mask(), bitmask_allocation_size_bytes(G), s, mr}, ext.null_count());
      auto const kind = e == 0 ? aggregation::ARGMIN : aggregation::ARGMAX;
      for (std::size_t j = 0; j < requests[r].aggregations.size(); ++j) {
        if (requests[r].aggregations[j]->kind != kind) continue;
        if (static_cast<int>(j) == slot[e]) continue;
        results[r].results[j] = std::make_unique<column>(index[e]->view(), stream, mr);  // a repeated ARGMIN / ARGMAX: a deep copy
      }
      results[r].results[static_cast<std::size_t>(slot[e])] = std::move(index[e]);
    }
  }
  CUDF_HIP_TRY(hipStreamSynchronize(s));
  return std::make_pair(std::move(ukeys), std::move(results));
}

}  // namespace cudf::groupby::detail
