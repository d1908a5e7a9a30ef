// SPDX-License-Identifier: Apache-2.0
// Host side of cudf::groupby::groupby for the hash path: request validation, aggregation flattening, record
// planning, strategy selection and kernel orchestration.
// Reference counterparts: cpp/src/groupby/groupby.cu:39-70,186-236 (ctor, validation, dispatch),
// cpp/src/groupby/hash/groupby.cu:33-147 (hash dispatch), extract_single_pass_aggs.cpp:26-177 (flattening),
// output_utils.cu:49-224 (result columns), hash_compound_agg_finalizer.cu:92-133 (MEAN).
#include "engine.hpp"
#include "../common/wc_scatter.hpp"

#include <cudf/groupby.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/utilities/error.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
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
    case aggregation::NTH_ELEMENT:
    case aggregation::MEDIAN: return source;
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
std::unique_ptr<Base> make_nth_element_aggregation(size_type, null_policy)
{
  return std::make_unique<detail::simple_aggregation>(aggregation::NTH_ELEMENT);
}
template std::unique_ptr<aggregation> make_nth_element_aggregation<aggregation>(size_type, null_policy);
template std::unique_ptr<groupby_aggregation> make_nth_element_aggregation<groupby_aggregation>(size_type, null_policy);

namespace groupby {
namespace detail {
namespace {

using cudf::detail::CLS_BOOL;
using cudf::detail::CLS_F32;
using cudf::detail::CLS_F64;
using cudf::detail::CLS_SINT;
using cudf::detail::CLS_UINT;
using cudf::detail::class_of;

int64_t env_i64(char const* name, int64_t dflt)
{
  char const* e = std::getenv(name);
  return (e != nullptr && *e != 0) ? std::strtoll(e, nullptr, 10) : dflt;
}

// HyperLogLog estimate (Flajolet et al. 2007) from m registers of ranks, with the small-range (linear counting)
// correction; a 64-bit hash needs no large-range correction.
double hyperloglog_estimate(std::vector<uint32_t> const& regs)
{
  double const m = static_cast<double>(regs.size());
  double z       = 0;
  int64_t zeros  = 0;
  for (uint32_t r : regs) {
    z += std::ldexp(1.0, -static_cast<int>(r));
    zeros += r == 0;
  }
  double const alpha = 0.7213 / (1.0 + 1.079 / m);
  double const e     = alpha * m * m / z;
  return (e <= 2.5 * m && zeros > 0) ? m * std::log(m / static_cast<double>(zeros)) : e;
}

// Kinds the hash engine computes (reference groupby/common/utils.hpp:66-85 lists the hashable set; the
// remaining ones need the sort path, which is out of scope — SURVEY.md §8f rank 4).
bool is_engine_kind(aggregation::Kind k)
{
  switch (k) {
    case aggregation::SUM:
    case aggregation::SUM_OVERFLOW:
    case aggregation::PRODUCT:
    case aggregation::MIN:
    case aggregation::MAX:
    case aggregation::COUNT_VALID:
    case aggregation::COUNT_ALL:
    case aggregation::MEAN:
    case aggregation::SUM_OF_SQUARES:
    case aggregation::M2:
    case aggregation::VARIANCE:
    case aggregation::STD:
    case aggregation::ARGMAX:
    case aggregation::ARGMIN: return true;
    default: return false;
  }
}

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

int find_or_add_acc(plan_dev& p, acc_desc const& d)
{
  for (int i = 0; i < p.NACC; ++i) {
    auto const& e = p.acc[i];
    if (e.op == d.op && e.src == d.src && e.pay == d.pay && e.valid_bit == d.valid_bit) return i;
  }
  CUDF_EXPECTS(p.NACC < MAX_ACC, "Too many distinct accumulators for one hash groupby call (limit 12).");
  p.acc[p.NACC] = d;
  return p.NACC++;
}

host_plan build_plan(table_view const& keys, null_policy policy, std::span<aggregation_request const> requests)
{
  host_plan hp;
  auto& p = hp.dev;
  CUDF_EXPECTS(keys.num_columns() >= 1, "groupby requires at least one key column.");
  // ---- columns: keys, then distinct value columns
  for (auto const& r : requests) {
    bool found = false;
    for (auto const& v : hp.value_cols) found = found || cudf::detail::is_shallow_equivalent(v, r.values);
    if (!found) hp.value_cols.push_back(r.values);
  }
  // float key columns also travel as value columns: the key units hold NORMALISED bits (-0.0 -> +0.0, one NaN), the output
  // key must be a representative input row (reference compute_groupby.cu:104-111)
  for (int c = 0; c < keys.num_columns(); ++c) {
    hp.key_raw_vidx[c] = -1;
    hp.key_acc[c]      = -1;
    auto const cls     = class_of(keys.column(c).type().id());
    if (cls != CLS_F32 && cls != CLS_F64) continue;
    int vidx = 0;
    for (; vidx < static_cast<int>(hp.value_cols.size()); ++vidx)
      if (cudf::detail::is_shallow_equivalent(hp.value_cols[vidx], keys.column(c))) break;
    if (vidx == static_cast<int>(hp.value_cols.size())) hp.value_cols.push_back(keys.column(c));
    hp.key_raw_vidx[c] = vidx;
  }
  CUDF_EXPECTS(static_cast<int>(hp.value_cols.size()) <= MAX_PAY - 1, "Too many distinct value columns (limit 7).");
  CUDF_EXPECTS(keys.num_columns() + static_cast<int>(hp.value_cols.size()) <= MAX_COLS,
               "Too many key + value columns for the hash path (limit 16).");
  std::vector<column_view> all;
  for (auto const& k : keys) all.push_back(k);
  for (auto const& v : hp.value_cols) all.push_back(v);
  auto const dt = cudf::detail::make_device_table(table_view{all});
  for (int c = 0; c < dt.ncols; ++c) p.cols[c] = dt.col[c];
  p.ncols    = dt.ncols;
  p.nkeycols = keys.num_columns();

  bool const keys_have_nulls = cudf::has_nulls(keys);
  p.drop_null_keys           = keys_have_nulls && policy == null_policy::EXCLUDE;
  bool const need_keynulls   = keys_have_nulls && policy == null_policy::INCLUDE;
  bool need_valvalid         = false;
  for (auto const& v : hp.value_cols) need_valvalid = need_valvalid || v.has_nulls();
  bool need_rowid = false;  // ARGMIN / ARGMAX: records carry the row index
  for (auto const& r : requests)
    for (auto const& a : r.aggregations) need_rowid = need_rowid || a->kind == aggregation::ARGMIN || a->kind == aggregation::ARGMAX;

  // ---- key units: 8-byte columns take a full unit, narrower ones share units two per unit
  int u = 0;
  for (int c = 0; c < p.nkeycols; ++c) {
    if (p.cols[c].width == 8) {
      CUDF_EXPECTS(u < MAX_KU, "Key too wide for the hash path (limit 32 bytes).");
      p.unit[u]       = unit_desc{1, static_cast<int8_t>(c), H_NONE, 1};
      p.key_mask[u]   = ~uint64_t{0};
      hp.key_unit[c]  = u;
      hp.key_half[c]  = 2;
      ++u;
    }
  }
  int half = 0;  // next free half in unit u (0 = lo of a fresh unit)
  auto put_half = [&](int8_t src, bool is_key_material) {
    if (half == 0) {
      CUDF_EXPECTS(u < MAX_UNITS, "Record too wide.");
      p.unit[u] = unit_desc{0, src, H_NONE, static_cast<int8_t>(is_key_material)};
      if (u < MAX_KU) p.key_mask[u] = is_key_material ? 0xffffffffull : 0;
      half = 1;
      return std::pair<int, int>{u, 0};
    }
    p.unit[u].hi = src;
    if (is_key_material && u < MAX_KU) p.key_mask[u] |= 0xffffffff00000000ull;
    half       = 0;
    int const w = u++;
    return std::pair<int, int>{w, 1};
  };
  for (int c = 0; c < p.nkeycols; ++c) {
    if (p.cols[c].width < 8) {
      CUDF_EXPECTS(u < MAX_KU, "Key too wide for the hash path (limit 32 bytes).");
      auto const [w, h] = put_half(static_cast<int8_t>(c), true);
      hp.key_unit[c]    = w;
      hp.key_half[c]    = h;
    }
  }
  if (need_keynulls) {
    CUDF_EXPECTS(u < MAX_KU, "Key too wide for the hash path (limit 32 bytes).");
    auto const [w, h]  = put_half(H_KEYNULLS, true);
    hp.keynulls_unit   = w;
    hp.keynulls_hi     = h;
  }
  p.flags_unit = -1;
  if (need_valvalid && half == 1) {  // free high half of the last key unit: park VALVALID there (masked out of the key)
    auto const [w, h] = put_half(H_VALVALID, false);
    p.flags_unit      = w;
    p.flags_hi        = h;
  }
  if (half == 1) {
    half = 0;
    ++u;
  }
  p.KU = u;
  CUDF_EXPECTS(p.KU <= MAX_KU, "Key too wide for the hash path (limit 32 bytes).");
  // ---- payload units: one per distinct value column, then VALVALID if it still needs a home
  for (int v = 0; v < static_cast<int>(hp.value_cols.size()); ++v) {
    CUDF_EXPECTS(u < MAX_UNITS, "Record too wide.");
    p.unit[u++] = unit_desc{1, static_cast<int8_t>(p.nkeycols + v), H_NONE, 0};
  }
  p.rowid_unit = -1;
  if (need_rowid) {  // row index in the low half; VALVALID rides in the high half if it still needs a home
    CUDF_EXPECTS(u < MAX_UNITS, "Record too wide.");
    bool const with_flags = need_valvalid && p.flags_unit < 0;
    p.unit[u]    = unit_desc{0, H_ROWID, with_flags ? H_VALVALID : H_NONE, 0};
    p.rowid_unit = u;
    if (with_flags) {
      p.flags_unit = u;
      p.flags_hi   = 1;
    }
    ++u;
  }
  if (need_valvalid && p.flags_unit < 0) {
    CUDF_EXPECTS(u < MAX_UNITS, "Record too wide.");
    p.unit[u]    = unit_desc{0, H_VALVALID, H_NONE, 0};
    p.flags_unit = u;
    p.flags_hi   = 0;
    ++u;
  }
  p.NPAY = u - p.KU;

  // ---- accumulators
  p.NACC = 0;
  for (auto const& r : requests) {
    int vidx = 0;
    for (; vidx < static_cast<int>(hp.value_cols.size()); ++vidx)
      if (cudf::detail::is_shallow_equivalent(hp.value_cols[vidx], r.values)) break;
    auto const vtype     = r.values.type();
    int const cls        = class_of(vtype.id());
    bool const has_nulls = r.values.has_nulls();
    int8_t const vbit    = has_nulls ? static_cast<int8_t>(vidx) : int8_t{-1};
    bool const is_float  = cls == CLS_F32 || cls == CLS_F64;
    auto count_valid_acc = [&]() {
      return has_nulls ? find_or_add_acc(p, acc_desc{ADD_I64, SRC_ONE_IF_VALID, static_cast<int8_t>(vidx), vbit})
                       : find_or_add_acc(p, acc_desc{ADD_I64, SRC_ONE, -1, -1});
    };
    auto sum_acc = [&](acc_src src) {
      return find_or_add_acc(p, acc_desc{static_cast<int8_t>(is_float ? ADD_F64 : ADD_I64), static_cast<int8_t>(src),
                                         static_cast<int8_t>(vidx), vbit});
    };
    for (auto const& agg : r.aggregations) {
      result_spec rs{};
      rs.kind      = agg->kind;
      rs.target    = cudf::detail::target_type(vtype, agg->kind);
      rs.value_idx = vidx;
      rs.acc_cls   = is_float ? CLS_F64 : (cls == CLS_UINT ? CLS_UINT : CLS_SINT);
      rs.nullable  = has_nulls;  // reference output_utils.cu:67-68 (COUNT handled below)
      switch (agg->kind) {
        case aggregation::SUM: rs.a0 = sum_acc(SRC_VALUE); break;
        case aggregation::SUM_OF_SQUARES: rs.a0 = sum_acc(SRC_SQUARE); break;
        case aggregation::PRODUCT:
          rs.a0 = find_or_add_acc(p, acc_desc{static_cast<int8_t>(is_float ? MUL_F64 : MUL_I64), SRC_VALUE, static_cast<int8_t>(vidx), vbit});
          break;
        case aggregation::MIN:
          rs.a0 = find_or_add_acc(
            p, acc_desc{static_cast<int8_t>(is_float ? MIN_F64 : (cls == CLS_SINT ? MIN_I64 : MIN_U64)), SRC_VALUE,
                        static_cast<int8_t>(vidx), vbit});
          break;
        case aggregation::MAX:
          rs.a0 = find_or_add_acc(
            p, acc_desc{static_cast<int8_t>(is_float ? MAX_F64 : (cls == CLS_SINT ? MAX_I64 : MAX_U64)), SRC_VALUE,
                        static_cast<int8_t>(vidx), vbit});
          break;
        case aggregation::ARGMIN:
        case aggregation::ARGMAX: {
          bool const is_min = agg->kind == aggregation::ARGMIN;
          int const valacc  = find_or_add_acc(
            p, acc_desc{static_cast<int8_t>(is_min ? (is_float ? MIN_F64 : (cls == CLS_SINT ? MIN_I64 : MIN_U64))
                                                   : (is_float ? MAX_F64 : (cls == CLS_SINT ? MAX_I64 : MAX_U64))),
                        SRC_VALUE, static_cast<int8_t>(vidx), vbit});
          int const before = p.NACC;
          rs.a0 = find_or_add_acc(p, acc_desc{MIN_I64, static_cast<int8_t>(is_min ? SRC_ARG_IDX : SRC_ARG_IDX_OF_MAX),
                                              static_cast<int8_t>(vidx), vbit});
          if (p.NACC != before) {  // a new pair (the same request twice shares it)
            CUDF_EXPECTS(p.narg < MAX_ARG, "Too many ARGMIN / ARGMAX aggregations in one call (limit 4).");
            p.arg[p.narg++] = arg_desc{static_cast<int8_t>(valacc), static_cast<int8_t>(rs.a0), static_cast<int8_t>(is_float), 0};
          }
          rs.acc_cls = CLS_SINT;
          break;
        }
        case aggregation::COUNT_VALID:
          rs.a0       = count_valid_acc();
          rs.nullable = false;
          break;
        case aggregation::COUNT_ALL:
          rs.a0       = find_or_add_acc(p, acc_desc{ADD_I64, SRC_ONE, -1, -1});
          rs.nullable = false;
          break;
        case aggregation::MEAN:
          // FLOAT64 for plain numerics; duration / decimal columns keep their type: integer division of the SUM in the
          // source type by the count (reference hash_compound_agg_finalizer.cu:92-133)
          rs.a0 = sum_acc(SRC_VALUE);
          rs.a1 = count_valid_acc();
          break;
        case aggregation::SUM_OVERFLOW:
          // exact sum: one int64 accumulator for sources of at most 4 bytes; hi / lo half sums for 8-byte sources
          if (size_of(vtype) < 8) {
            rs.a0 = sum_acc(SRC_VALUE);
          } else {
            rs.a0 = sum_acc(SRC_HI32);
            rs.a2 = sum_acc(SRC_LO32);
          }
          break;
        case aggregation::M2:
        case aggregation::VARIANCE:
        case aggregation::STD: {
          // reference extract_single_pass_aggs.cpp:26-177: {SUM_OF_SQUARES, SUM, COUNT_VALID}
          rs.a0       = sum_acc(SRC_SQUARE);
          rs.a1       = sum_acc(SRC_VALUE);
          rs.a2       = count_valid_acc();
          rs.nullable = agg->kind != aggregation::M2;  // M2 is never null; VAR/STD get a mask from the counts
          if (auto const* dd = dynamic_cast<cudf::detail::ddof_aggregation const*>(agg.get())) rs.ddof = dd->_ddof;
          hp.results.push_back(rs);
          continue;
        }
        default: CUDF_FAIL("Unsupported aggregation on the hash path.");
      }
      if (rs.nullable) rs.valid_acc = count_valid_acc();
      hp.results.push_back(rs);
    }
  }
  for (int c = 0; c < p.nkeycols; ++c)
    if (hp.key_raw_vidx[c] >= 0)
      hp.key_acc[c] = find_or_add_acc(p, acc_desc{ANY_U64, SRC_VALUE, static_cast<int8_t>(hp.key_raw_vidx[c]), -1});
  // ---- fast path: all units are plain 8-byte columns (no nulls, no conversion, no normalisation)
  p.simple = 1;
  for (int w = 0; w < p.KU + p.NPAY; ++w) {
    auto const& d = p.unit[w];
    if (!d.full) { p.simple = 0; break; }
    auto const& c = p.cols[d.lo];
    bool const plain = c.width == 8 && c.mask == nullptr &&
                       (d.is_key ? (c.cls == CLS_SINT || c.cls == CLS_UINT)
                                 : (c.cls == CLS_SINT || c.cls == CLS_UINT || c.cls == CLS_F64));
    if (!plain) { p.simple = 0; break; }
    p.simple_base[w] = static_cast<uint64_t const*>(c.head) + c.offset;
  }
  if (env_i64("CUDF_AMD_GB_NO_SIMPLE", 0) || need_rowid) p.simple = 0;
  p.simple_vec16 = p.simple;
  for (int w = 0; p.simple && w < p.KU + p.NPAY; ++w)
    if (reinterpret_cast<uintptr_t>(p.simple_base[w]) % 16 != 0) p.simple_vec16 = 0;
  // measured slower than 8-byte loads in both the histogram (1.85 vs 1.6 ms) and the scatter (25.5M vs 22.2M
  // cycles per workgroup): opt-in only
  if (!env_i64("CUDF_AMD_GB_VEC16", 0)) p.simple_vec16 = 0;
  return hp;
}

// Heavy-hitter handling covers plans whose accumulators are SUMs of the single value column and row COUNTs (no nulls).
bool hot_plan_ok(plan_dev const& p)
{
  if (p.NACC < 1 || p.NACC > 2 || p.narg != 0) return false;
  for (int q = 0; q < p.NACC; ++q) {
    auto const& a = p.acc[q];
    bool const sum = (a.op == ADD_F64 || a.op == ADD_I64) && a.src == SRC_VALUE && a.pay == 0 && a.valid_bit < 0;
    bool const cnt = a.op == ADD_I64 && a.src == SRC_ONE;
    if (!sum && !cnt) return false;
  }
  return true;
}

// Page-locked host staging for the call's small read-backs (overflow flag, group counts, null counts): a hipMemcpyAsync into
// pageable memory blocks the host until the copy is done, so every read-back was a stream synchronisation of its own.
int32_t* pinned_ints(std::size_t count)
{
  thread_local int32_t* buf   = nullptr;
  thread_local std::size_t cap = 0;
  if (count > cap) {
    if (buf != nullptr) (void)hipHostFree(buf);
    cap = std::max<std::size_t>(count, 4096);
    CUDF_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&buf), cap * sizeof(int32_t), hipHostMallocDefault));
  }
  return buf;
}

// The same for the estimate pass's read-backs (bitmap population, key ranges, heavy-hitter table): its own buffer, so that the
// pointers pinned_ints() hands out stay valid next to it.
unsigned char* pinned_bytes(std::size_t count)
{
  thread_local unsigned char* buf = nullptr;
  thread_local std::size_t cap    = 0;
  if (count > cap) {
    if (buf != nullptr) (void)hipHostFree(buf);
    cap = std::max<std::size_t>(count, 65536);
    CUDF_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&buf), cap, hipHostMallocDefault));
  }
  return buf;
}

// CUDF_AMD_GB_TRACE=1: host-side timeline of a call (microseconds since entry at every mark), printed to stderr when the call returns.
struct call_trace {
  bool on;
  std::chrono::steady_clock::time_point t0;
  std::string line;
  call_trace() : on{env_i64("CUDF_AMD_GB_TRACE", 0) != 0}, t0{std::chrono::steady_clock::now()} {}
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

}  // namespace
}  // namespace detail

groupby::groupby(table_view const& keys, null_policy null_handling, sorted keys_are_sorted,
                 std::vector<order> const& column_order, std::vector<null_order> const& null_precedence)
  : _keys{keys}, _include_null_keys{null_handling}, _keys_are_sorted{keys_are_sorted}, _column_order{column_order},
    _null_precedence{null_precedence}
{
}
groupby::~groupby() = default;

std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> groupby::aggregate(
  std::span<aggregation_request const> requests, stream_ref stream, rmm::device_async_resource_ref mr)
{
  using namespace detail;
  // keys_are_sorted is a hint with which the reference picks its sort-based path (groupby.cu:64-69); group order is
  // unspecified either way, so the hash path serves both
  (void)_keys_are_sorted;
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
  for (auto const& r : requests)
    for (auto const& a : r.aggregations)
      CUDF_EXPECTS(is_engine_kind(a->kind),
                   "This aggregation needs the sort-based groupby, which this build does not provide.");

  hipStream_t const s = stream.value();
  auto tmp_mr         = cudf::get_current_device_resource_ref();
  detail::call_trace trace;

  // ---- empty input: typed empty outputs (reference groupby.cu:233, :87-182)
  if (_keys.num_rows() == 0) {
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
    _last_path = hash_path::NONE;
    return {empty_like(_keys), std::move(res)};
  }

  host_plan hp     = build_plan(_keys, _include_null_keys, requests);
  trace.mark("plan");
  plan_dev const& p = hp.dev;
  int64_t const n   = _keys.num_rows();
  int const RU = p.KU + p.NPAY, PU = p.KU + p.NACC;

  // ---- geometry
  agg_geom ag{};
  int64_t const lds_budget = env_i64("CUDF_AMD_GB_LDS_KB", 159) * 1024;  // 160 KiB minus the kernels' static words
  int const slot_bytes     = aggregate_slot_bytes(p);  // key units + accumulators (COUNTs take 4 bytes) + state word
  // a multiple of 4: the table is probed in aligned buckets of four slots (one ds_read_b128 of state words)
  ag.cap                   = static_cast<int32_t>(std::min<int64_t>(lds_budget / slot_bytes, 16384)) & ~3;
  ag.block                 = static_cast<int32_t>(env_i64("CUDF_AMD_GB_AGG_BLOCK", 1024));
  ag.fill_limit            = static_cast<int32_t>(ag.cap * 0.6);
  CUDF_EXPECTS(ag.cap >= 64, "Aggregation state per group too large for an LDS table.");

  scratch sc{s, tmp_mr, {}};
  int32_t* d_overflow = sc.alloc<int32_t>(1);

  std::vector<uint64_t> hot_keys;  // heavy hitters found in the sample (aggregated inside the scatter workgroups)
  // Dense-key candidate: one plain 8-byte integer key column, one plain 8-byte value column, no ARGMIN / ARGMAX
  bool const dense_signed    = p.cols[0].cls == cudf::detail::CLS_SINT;
  // (the dense-key, heavy-hitter and pre-aggregation paths are for big inputs; CUDF_AMD_GB_BIG_MIN_ROWS lets the fuzz tests walk
  // them at sizes a CPU checker can follow)
  int64_t const big_rows     = env_i64("CUDF_AMD_GB_BIG_MIN_ROWS", int64_t{1} << 22);
  bool const dense_candidate = p.simple && p.KU == 1 && p.NPAY == 1 && p.narg == 0 && n >= big_rows &&
                               env_i64("CUDF_AMD_GB_DENSE", 1) != 0;
  // Composite dense keys: 1-4 integer key columns of any width (rows with a NULL key are dropped: no nullable key under
  // null_policy::INCLUDE), exactly one value column, no ARGMIN / ARGMAX
  bool dense_composite = !dense_candidate && p.nkeycols <= DENSE_MAX_KEYS && hp.value_cols.size() == 1 && p.narg == 0 &&
                         hp.keynulls_unit < 0 && n >= big_rows && env_i64("CUDF_AMD_GB_DENSE", 1) != 0 &&
                         env_i64("CUDF_AMD_GB_DENSE_COMPOSITE", 1) != 0;
  for (int c = 0; c < p.nkeycols && dense_composite; ++c)
    dense_composite = p.cols[c].cls == cudf::detail::CLS_SINT || p.cols[c].cls == cudf::detail::CLS_UINT;
  bool allow_dense   = dense_candidate || dense_composite;
  uint64_t h_range[2] = {0, 0};  // sample minimum / maximum of the key column (bit patterns)
  int64_t h_ranges[2 * MAX_KU] = {0};  // composite: per key column, as int64
  int64_t final_cap  = 0;        // records per work item in `partial` (0: ag.cap)
  double adjacent_equal = 0.0;   // share of the sampled rows whose successor row carries the same key
  bool ranges_known     = false; // the estimate pass ran and left the sampled key ranges in h_range / h_ranges
  // ---- distinct-count estimate on a strided sample (skipped when n already fits one table)
  double est_groups = static_cast<double>(n);
  // Small inputs skip the estimate (three memsets, three kernels and a stream synchronisation: about a third of a 10K-row call):
  // they are planned for one table per 16K-row chunk, and a table that overflows sends the call through escalate() - which
  // counts the keys over all rows - like any other misjudged cardinality.
  bool const skip_estimate = n < env_i64("CUDF_AMD_GB_ESTIMATE_MIN_ROWS", 1 << 16);
  if (skip_estimate && n > ag.fill_limit) est_groups = static_cast<double>(ag.fill_limit) / 1.3 - 1.0;
  if (n > ag.fill_limit && !skip_estimate) {
    // 1M sampled rows for big inputs; small inputs sample 1/16 of their rows (at least 64K): the estimate only picks the
    // strategy, and a 1M-row sample costs more than the aggregation of a 1M-row input
    int64_t const sample = std::min<int64_t>(n, std::clamp<int64_t>(n / 16, int64_t{1} << 16, int64_t{1} << 20));
    int const bits_log2  = 24;
    uint32_t* bitmap     = sc.alloc<uint32_t>((size_t{1} << bits_log2) / 32);
    uint32_t* d_set      = sc.alloc<uint32_t>(1);
    plan_dev* d_plan = sc.alloc<plan_dev>(1);
    bool const hot_eligible = p.simple && RU == 2 && p.KU == 1 && hot_plan_ok(p) && n >= big_rows &&
                              env_i64("CUDF_AMD_GB_HOT", 1) != 0;
    uint32_t* hot_buckets = hot_eligible ? sc.alloc<uint32_t>(HOT_BUCKETS) : nullptr;
    // (one plain integer key column: the same pass takes the minimum and maximum of the sampled keys for the dense-key test)
    uint64_t* d_range = dense_candidate ? sc.alloc<uint64_t>(2) : nullptr;
    uint64_t* d_blk_range = dense_candidate ? sc.alloc<uint64_t>(2 * static_cast<std::size_t>((sample + 255) / 256)) : nullptr;
    // (and how often a row's successor carries the same key: sorted / clustered inputs are aggregated in row chunks first)
    bool const want_adj = n >= big_rows && p.narg == 0 && env_i64("CUDF_AMD_GB_PREAGG", 1) != 0;
    uint32_t* d_blk_adj = want_adj ? sc.alloc<uint32_t>(2 * static_cast<std::size_t>((sample + 255) / 256)) : nullptr;
    uint32_t* d_adj     = want_adj ? sc.alloc<uint32_t>(2) : nullptr;
    launch_estimate(p, d_plan, n, sample, bitmap, bits_log2, d_set, hot_buckets, s, dense_candidate ? (dense_signed ? 1 : 2) : 0, d_blk_range, d_range,
                    d_blk_adj, d_adj);
    // Dense integer keys ("Dense keys"): minimum and maximum of the key column over the same sample
    // (every read-back of this pass lands in page-locked memory: a copy into pageable memory blocks the host until it is done)
    unsigned char* const pin = pinned_bytes(64 + 16 + sizeof(h_ranges) + HOT_TABLE * (sizeof(uint64_t) + sizeof(uint32_t)));
    uint32_t* const pin_set    = reinterpret_cast<uint32_t*>(pin);
    uint64_t* const pin_range  = reinterpret_cast<uint64_t*>(pin + 64);
    int64_t* const pin_ranges  = reinterpret_cast<int64_t*>(pin + 64 + 16);
    uint64_t* const pin_tkeys  = reinterpret_cast<uint64_t*>(pin + 64 + 16 + sizeof(h_ranges));
    uint32_t* const pin_tcounts = reinterpret_cast<uint32_t*>(pin_tkeys + HOT_TABLE);
    if (dense_candidate) {
      CUDF_HIP_TRY(hipMemcpyAsync(pin_range, d_range, 16, hipMemcpyDeviceToHost, s));
    } else if (dense_composite) {
      int64_t* d_ranges = sc.alloc<int64_t>(2 * MAX_KU);
      launch_key_ranges(d_plan, p.nkeycols, n, sample, d_ranges, s);
      CUDF_HIP_TRY(hipMemcpyAsync(pin_ranges, d_ranges, sizeof(h_ranges), hipMemcpyDeviceToHost, s));
    }
    // Heavy hitters (plain int64 key + one plain value, SUM / COUNT): a key above ~0.05 % of the rows overflows its
    // regions of the optimistic partition, and a key with percents of the rows leaves one workgroup aggregating its
    // partition alone. Keys seen min_count times in the sample are aggregated inside the scatter workgroups instead.
    std::vector<uint64_t> h_tkeys;
    std::vector<uint32_t> h_tcounts;
    // (a key overflows its regions from about 0.24 / P of the rows: 0.023 % at P = 1024; the threshold is half of that)
    uint32_t const hot_min_count = static_cast<uint32_t>(std::max<int64_t>(16, sample / 4 / 8192));  // of every 4th sampled row
    if (hot_eligible) {
      uint32_t* buckets = hot_buckets;
      uint64_t* tkeys   = sc.alloc<uint64_t>(HOT_TABLE);
      uint32_t* tcounts = sc.alloc<uint32_t>(HOT_TABLE + 1);
      launch_hot_keys(d_plan, n, sample, hot_min_count, buckets, d_set, tkeys, tcounts, s);
      CUDF_HIP_TRY(hipMemcpyAsync(pin_tkeys, tkeys, HOT_TABLE * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
      CUDF_HIP_TRY(hipMemcpyAsync(pin_tcounts, tcounts, HOT_TABLE * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    }
    CUDF_HIP_TRY(hipMemcpyAsync(pin_set, d_set, 4, hipMemcpyDeviceToHost, s));
    if (want_adj) CUDF_HIP_TRY(hipMemcpyAsync(pin_set + 2, d_adj, 8, hipMemcpyDeviceToHost, s));
    trace.mark("estimate queued");
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    trace.mark("estimate back");
    uint32_t const h_set = *pin_set;
    if (want_adj && pin_set[2] >= 1024) adjacent_equal = static_cast<double>(pin_set[3]) / static_cast<double>(pin_set[2]);
    if (dense_candidate) std::memcpy(h_range, pin_range, 16);
    if (dense_composite) std::memcpy(h_ranges, pin_ranges, sizeof(h_ranges));
    ranges_known = dense_candidate || dense_composite;
    if (hot_eligible) {
      h_tkeys.assign(pin_tkeys, pin_tkeys + HOT_TABLE);
      h_tcounts.assign(pin_tcounts, pin_tcounts + HOT_TABLE);
    }
    if (hot_eligible) {  // the most frequent keys first, at most HOT_MAX_KEYS of them
      std::vector<std::pair<uint32_t, uint64_t>> cand;
      for (int i = 0; i < HOT_TABLE; ++i)
        if (h_tkeys[i] != ~uint64_t{0} && h_tcounts[i] >= hot_keys_threshold(hot_min_count, sample, h_set)) cand.emplace_back(h_tcounts[i], h_tkeys[i]);
      std::sort(cand.begin(), cand.end(), [](auto const& a, auto const& b) { return a.first > b.first; });
      if (cand.size() > HOT_MAX_KEYS) cand.resize(HOT_MAX_KEYS);
      for (auto const& c : cand) hot_keys.push_back(c.second);
      if (env_i64("CUDF_AMD_DEBUG", 0))
        fprintf(stderr, "[cudf_amd] heavy hitters: %zu keys (most frequent: %u of %ld counted rows)\n", hot_keys.size(),
                cand.empty() ? 0u : cand[0].first, (long)(sample / 4));
    }
    double const m  = std::ldexp(1.0, bits_log2);
    double const ds = h_set >= m ? m * 20 : -m * std::log(1.0 - h_set / m);  // distinct keys in the sample
    // population estimate under uniform frequencies: solve G (1 - exp(-S/G)) = ds
    double const S = static_cast<double>(sample);
    // (few duplicates in the sample still carry information: distinct ~ S - S^2 / 2G; only a sample without any
    // duplicate leaves G unbounded)
    if (ds >= S - 0.5 || sample == n) {
      est_groups = sample == n ? ds : static_cast<double>(n);
    } else {
      double lo = ds, hi = static_cast<double>(n);
      for (int it = 0; it < 60; ++it) {
        double const g = 0.5 * (lo + hi);
        if (g * (1.0 - std::exp(-S / g)) < ds) lo = g; else hi = g;
      }
      est_groups = std::min<double>(hi, static_cast<double>(n));
    }
    est_groups = std::max(est_groups, 1.0);
  }
  int64_t const forced_p = env_i64("CUDF_AMD_GB_P", 0);
  bool allow_optimistic    = env_i64("CUDF_AMD_GB_OPTIMISTIC", 1) != 0;
  bool const forced_exact  = env_i64("CUDF_AMD_GB_EXACT", 0) != 0;

  uint64_t* partial  = nullptr;  // final partial records: item i at [i*cap, i*cap + count[i])
  int32_t* d_count   = nullptr;
  int32_t nitems     = 0;
  double safety      = 1.3;
  bool counted_all   = false;  // the HyperLogLog pass over all rows has run (after a table overflow)

  // one stream synchronisation returns both the overflow flag and the per-item group counts of an attempt
  std::vector<int32_t> h_count;
  auto overflow_and_counts = [&]() -> int32_t {
    int32_t* const pin = pinned_ints(static_cast<std::size_t>(nitems) + 1);
    CUDF_HIP_TRY(hipMemcpyAsync(pin, d_overflow, 4, hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipMemcpyAsync(pin + 1, d_count, sizeof(int32_t) * static_cast<std::size_t>(nitems), hipMemcpyDeviceToHost, s));
    trace.mark("attempt queued");
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    trace.mark("attempt back");
    h_count.assign(pin + 1, pin + 1 + nitems);
    return pin[0];
  };
  // Heavy hitters stay in the (first-level) scatter workgroups: `pa` gets the key list and the per-workgroup partial
  // buffers; merge_hot() folds those partials into one more work item behind the tables' items.
  auto setup_hot = [&](part_args& pa, int64_t P) -> bool {
    // (first the cheap tests: a scatter without write-combining has granule 0, and wc_scatter_lds_bytes divides by it - 1B rows with
    // two value columns on 1M groups, 24-byte records at 1024 partitions, died of SIGFPE here)
    if (hot_keys.empty() || pa.wc_granule == 0 || !p.simple || RU != 2) return false;
    auto const wc_lds = cudf::detail::wc_scatter_lds_bytes(5 * 1024, static_cast<std::size_t>(P), pa.wc_granule, 2);
    if (wc_lds + partition_hot_lds_bytes() + 1200 > 160 * 1024) return false;  // (the LDS table needs room next to the tile)
    size_t const wgs = static_cast<size_t>(pa.geom.slices);
    uint64_t* d_hot  = sc.alloc<uint64_t>(HOT_MAX_KEYS);
    CUDF_HIP_TRY(hipMemcpyAsync(d_hot, hot_keys.data(), hot_keys.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    pa.hot_n          = static_cast<int32_t>(hot_keys.size());
    pa.hot_keys       = d_hot;
    pa.hot_lds_offset = static_cast<int32_t>(wc_lds);
    pa.hot_out        = sc.alloc<uint64_t>(wgs * HOT_SLOTS * PU);
    pa.hot_count      = sc.alloc<int32_t>(wgs);
    return true;
  };
  auto merge_hot = [&](part_args const& pa, agg_args const& aa) {  // partial / d_count hold room for item `nitems`
    agg_args hm    = aa;
    hm.input       = IN_PARTIAL_RECORDS;
    hm.seg         = SEG_STRIDED;
    hm.records     = pa.hot_out;
    hm.src_count   = pa.hot_count;
    hm.src_stride  = HOT_SLOTS;
    hm.fan         = pa.geom.slices;
    hm.nsrc        = pa.geom.slices;
    hm.out_records = partial + static_cast<size_t>(nitems) * ag.cap * PU;
    hm.out_count   = d_count + nitems;
    hm.nitems      = 1;
    launch_aggregate(hm, sc.alloc<agg_args>(1), s);
    nitems += 1;
  };
  auto escalate = [&]() {
    // A table overflowed: the estimate was too low (skewed sample). First time: count the distinct key rows over ALL rows (HyperLogLog, one
    // streaming pass over the key columns) and plan from that; after that, ask for 8x more tables and redo.
    sc.bufs.clear();
    if (!counted_all) {
      counted_all       = true;
      uint32_t* regs    = sc.alloc<uint32_t>(HLL_REGISTERS);
      plan_dev* d_plan2 = sc.alloc<plan_dev>(1);
      launch_distinct_count(p, d_plan2, n, regs, s);
      std::vector<uint32_t> h_regs(HLL_REGISTERS);
      CUDF_HIP_TRY(hipMemcpyAsync(h_regs.data(), regs, HLL_REGISTERS * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      CUDF_HIP_TRY(hipStreamSynchronize(s));
      double const counted = hyperloglog_estimate(h_regs);
      if (env_i64("CUDF_AMD_DEBUG", 0))
        fprintf(stderr, "[cudf_amd] table overflow: sample estimate %.0f groups, HyperLogLog over all rows %.0f\n", est_groups, counted);
      // (1.05: three standard errors of the 2^14-register estimate; never plan for fewer groups than the failed attempt)
      est_groups = std::min<double>(static_cast<double>(n), std::max(est_groups * 1.5, counted * 1.05));
      sc.bufs.clear();
    } else {
      // (an estimate already at the row count cannot grow: the planned load drops instead - tables of a few hundred slots,
      // CUDF_AMD_GB_LDS_KB=8 in the fuzz tests, overflowed at 4 sigma of an all-distinct key column and the same plan was redone)
      if (est_groups >= static_cast<double>(n)) safety *= 2.0;
      est_groups = std::min<double>(static_cast<double>(n), std::max(est_groups, static_cast<double>(ag.fill_limit)) * 8);
    }
    d_overflow = sc.alloc<int32_t>(1);
  };
  // Exact pipeline (histogram, scan, scatter with exact offsets, one or two levels, then one table per partition) over the rows
  // `pa` describes: the plan's columns, or `nrec` records of `units` 8-byte units each (raw records, or - in_mode ==
  // IN_PARTIAL_RECORDS - partial records of a local pre-aggregation; the partition kernels only hash the key units, `pplan` is
  // the plan as they shall see it). Leaves partial / d_count / nitems for the finalize step.
  auto exact_pipeline = [&](part_args& pa, part_args* d_pa, plan_dev const& pplan, int64_t nrec, int units, agg_input in_mode, int64_t P1,
                            int64_t P2, int log2P1, int log2P2, agg_args& aa) {
    size_t const items1 = static_cast<size_t>(pa.geom.nseg) * static_cast<size_t>(pa.geom.slices);
    pa.counts       = sc.alloc<uint32_t>(items1 * P1);
    pa.item_base    = sc.alloc<int64_t>(items1 * P1);
    pa.out_offsets  = sc.alloc<int64_t>(P1 + 1);
    uint64_t* recA  = sc.alloc<uint64_t>(static_cast<size_t>(nrec) * units);
    pa.out_records  = recA;
    store_args(pa, d_pa, s);
    launch_partition_hist(pa, d_pa, s);
    launch_partition_scan(pa, d_pa, s);
    launch_partition_scatter(pa, d_pa, s);
    int64_t const* offsets = pa.out_offsets;
    uint64_t const* recs   = recA;
    int64_t nparts         = P1;
    if (P2 > 1) {
      part_args pb{};
      pb.plan         = pplan;
      pb.geom.nseg    = static_cast<int32_t>(P1);
      pb.geom.slices  = static_cast<int32_t>(std::max<int64_t>(1, 1024 / P1));
      pb.geom.P       = static_cast<int32_t>(P2);
      pb.geom.shift   = 64 - log2P1 - log2P2;
      pb.geom.block   = 1024;
      pb.from_columns = 0;
      pb.in_records   = recA;
      pb.seg_offsets  = pa.out_offsets;
      size_t const items2 = static_cast<size_t>(pb.geom.nseg) * pb.geom.slices;
      pb.counts       = sc.alloc<uint32_t>(items2 * P2);
      pb.item_base    = sc.alloc<int64_t>(items2 * P2);
      pb.out_offsets  = sc.alloc<int64_t>(P1 * P2 + 1);
      uint64_t* recB  = sc.alloc<uint64_t>(static_cast<size_t>(nrec) * units);
      pb.out_records  = recB;
      part_args* d_pb = sc.alloc<part_args>(1);
      store_args(pb, d_pb, s);
      launch_partition_hist(pb, d_pb, s);
      launch_partition_scan(pb, d_pb, s);
      launch_partition_scatter(pb, d_pb, s);
      offsets = pb.out_offsets;
      recs    = recB;
      nparts  = P1 * P2;
    }
    nitems         = static_cast<int32_t>(nparts);
    partial        = sc.alloc<uint64_t>(static_cast<size_t>(nitems) * ag.cap * PU);
    d_count        = sc.alloc<int32_t>(nitems);
    aa.input       = in_mode;
    aa.seg         = SEG_OFFSETS;
    aa.offsets     = offsets;
    aa.records     = recs;
    aa.out_records = partial;
    aa.out_count   = d_count;
    aa.nitems      = nitems;
    launch_aggregate(aa, sc.alloc<agg_args>(1), s);
  };
  // The dense map of the key columns from the sampled ranges (h_range / h_ranges): lo, range, the mixed-radix digits of composite
  // keys. False if the keys are not dense-eligible or span too much.
  auto dense_map_from_sample = [&](dense_map& dm, bool tight = false) -> bool {
    bool dense_ok = false;
    if (!ranges_known) return false;
    if (dense_candidate) {
      // one plain 8-byte key: range from the sample, widened by a margin (the sample's extremes of a dense column miss the
      // true ones by about range / sample); a key outside [lo, lo + range) voids the attempt (overflow bit 2): redone by hash
      uint64_t const width  = h_range[1] - h_range[0];  // exact in two's complement for either ordering
      // (tight: a handful of groups, every one of them sampled hundreds of times - the one-table path wants a small table)
      uint64_t const margin = tight ? std::max<uint64_t>(width / 32, 64) : std::clamp<uint64_t>(width / 64, 4096, uint64_t{1} << 26);
      uint64_t lo, hi;
      if (dense_signed) {
        int64_t const l = static_cast<int64_t>(h_range[0]), h = static_cast<int64_t>(h_range[1]);
        lo = static_cast<uint64_t>(l < INT64_MIN + static_cast<int64_t>(margin) ? INT64_MIN : l - static_cast<int64_t>(margin));
        hi = static_cast<uint64_t>(h > INT64_MAX - static_cast<int64_t>(margin) ? INT64_MAX : h + static_cast<int64_t>(margin));
      } else {
        lo = h_range[0] < margin ? 0 : h_range[0] - margin;
        hi = h_range[1] > UINT64_MAX - margin ? UINT64_MAX : h_range[1] + margin;
      }
      dm.lo    = lo;
      dm.range = hi - lo + 1;  // (0 if the keys span the whole type: fails the test below)
      dense_ok = width <= (uint64_t{1} << 30) && dm.range != 0;
    } else {
      // composite: every key column contributes the digit (value - lo_c) of a mixed-radix index, last column fastest
      double total = 1.0;
      dense_ok     = true;
      for (int c = 0; c < p.nkeycols; ++c) {
        int64_t const l = h_ranges[2 * c], h = h_ranges[2 * c + 1];
        if (l > h || static_cast<double>(h) - static_cast<double>(l) > 1e9) { dense_ok = false; break; }  // (no valid sampled value / wide)
        int64_t const width  = h - l;
        int64_t const margin = width / 64 + (width >= 64 ? 2 : 0);
        dense_key& dk = dm.key[c];
        dk.lo        = static_cast<uint64_t>(l - margin);
        dk.range     = static_cast<uint32_t>(width + 2 * margin + 1);
        dk.col       = static_cast<int8_t>(c);
        dk.unit      = static_cast<int8_t>(hp.key_unit[c]);
        dk.half      = static_cast<int8_t>(hp.key_half[c]);
        dk.is_signed = p.cols[c].cls == cudf::detail::CLS_SINT;
        dk.width     = p.cols[c].width;
        total *= static_cast<double>(dk.range);
      }
      dense_ok = dense_ok && total <= static_cast<double>(uint64_t{1} << 30);
      if (dense_ok) {
        uint64_t stride = 1;
        for (int c = p.nkeycols - 1; c >= 0; --c) {
          dm.key[c].stride = static_cast<uint32_t>(stride);
          stride *= dm.key[c].range;
        }
        dm.range          = stride;
        dm.nkeys          = p.nkeycols;
        dm.value_col      = p.nkeycols;
        dm.value_nullable = p.cols[p.nkeycols].mask != nullptr;
      }
    }
    return dense_ok;
  };
  bool pre_failed = false;  // the local pre-aggregation of this call overflowed a chunk's table: not tried again
  for (int attempt = 0;; ++attempt) {
    CUDF_EXPECTS(attempt < 4, "hash groupby: could not fit the groups into LDS tables (pathological key distribution).");
    CUDF_HIP_TRY(hipMemsetAsync(d_overflow, 0, 4, s));
    // tables needed; the group count can never exceed the row count
    // Planned table load. Bucketed probing resolves a row in two LDS round trips up to ~0.4; a lighter table means
    // more partitions. 16-byte records (write-combining scatter): halving the fan-out saves more in the scatter
    // (C2: 8.3 -> 7.1 ms) than the fuller tables cost the aggregate (3.1 -> 3.7 ms), so plan for 0.45/safety = 0.35.
    bool const wc_eligible = (RU == 2 || RU == 3) && p.KU == 1 && env_i64("CUDF_AMD_GB_WC", 1) != 0;
    double const plan_fill = std::max(1.0, ag.cap * 0.01 * static_cast<double>(env_i64("CUDF_AMD_GB_PLAN_LOAD_PCT", wc_eligible ? 45 : 25)));
    // (groups <= rows, and the safety factor rides on top of that bound: more than `safety` x n / plan_fill tables are never needed
    // for the tables' MEAN load, but without it an all-distinct key column was planned at the full load with no slack)
    double const need = std::min(est_groups, static_cast<double>(n)) * (est_groups >= static_cast<double>(n) ? safety : std::min(safety, static_cast<double>(n) / est_groups)) / plan_fill;
    agg_args aa{};
    aa.plan     = p;
    aa.geom     = ag;
    aa.overflow = d_overflow;

    bool const fits_one_table = std::min(est_groups * safety, static_cast<double>(n)) <= ag.fill_limit;
    // ---------------- path T: a key range small enough for ONE direct-address table (up to 8192 groups for SUM + COUNT, whatever
    // the hash tables would hold): every workgroup aggregates its row tiles straight from the columns into a table of its own
    // (no hash, no probe, no key compare, no partition pass) and k_dense_merge_dump_wide folds the images.
    if (forced_p == 0 && allow_dense && env_i64("CUDF_AMD_GB_DENSE_ONE_TABLE", 1) != 0) {
      dense_map dm{};
      bool ok = dense_map_from_sample(dm, true);
      int bits = 6;
      while (bits < 30 && (uint64_t{1} << bits) < dm.range) ++bits;
      int const slots         = 1 << bits;
      std::size_t const image = ok ? dense_table_bytes(p, slots) : 0;
      ok = ok && dm.range <= static_cast<uint64_t>(slots) && image <= 96 * 1024;
      if (ok) {
        _last_path   = hash_path::DENSE_DIRECT;
        dm.mult      = 1;
        dm.mult_inv  = 1;
        dm.bits      = bits;
        dm.log2P     = 0;
        int const DPU    = (dm.nkeys > 0 ? p.KU : 1) + p.NACC;
        int const nwg    = static_cast<int>(std::clamp<int64_t>(n / 16384, 1, image <= 48 * 1024 ? 512 : 256));
        int const dsplit = slots / 64;  // (items of the image fold: 64 slots each)
        dense_agg_args da{};
        da.plan        = p;
        da.map         = dm;
        da.nsplit      = nwg;
        da.slots       = slots;
        da.image_bytes = static_cast<int32_t>(image);
        da.occ_acc     = dense_occ_acc(p);
        da.KU          = dm.nkeys > 0 ? p.KU : 1;
        da.tables      = sc.alloc<uint64_t>(static_cast<size_t>(nwg) * image / 8);
        partial        = sc.alloc<uint64_t>(static_cast<size_t>(slots) * DPU);
        d_count        = sc.alloc<int32_t>(dsplit);
        da.out_records = partial;
        da.out_count   = d_count;
        da.overflow    = d_overflow;
        da.nitems      = 1;
        da.block       = 1024;
        da.nrows       = n;
        if (dm.nkeys > 0) {
          uint32_t* ones = sc.alloc<uint32_t>(16);
          CUDF_HIP_TRY(hipMemsetAsync(ones, 0xff, 64, s));
          da.ones = ones;
        }
        dense_agg_args* d_da = sc.alloc<dense_agg_args>(1);
        store_args(da, d_da, s);
        launch_aggregate_dense_columns(da, d_da, s);
        launch_dense_merge_dump_wide(da, d_da, s);
        nitems    = dsplit;
        final_cap = 64;
        int32_t const h_ov = overflow_and_counts();
        if (env_i64("CUDF_AMD_DEBUG", 0))
          fprintf(stderr, "[cudf_amd] dense keys (one table): nkeys=%d lo=%lld range=%llu slots=%d image=%zu B workgroups=%d overflow=%d\n", dm.nkeys,
                  (long long)dm.lo, (unsigned long long)dm.range, slots, image, nwg, h_ov);
        if (h_ov == 0) break;
        // a key outside the sampled range: the hash tables
        allow_dense = false;
        final_cap   = 0;
        sc.bufs.clear();
        d_overflow = sc.alloc<int32_t>(1);
        --attempt;
        continue;
      }
    }
    if (fits_one_table && forced_p == 0) {
      // ---------------- path S: every workgroup aggregates a row chunk in LDS, then partials are merged
      _last_path          = hash_path::LDS_SINGLE_PASS;
      int64_t const items = std::clamp<int64_t>(n / 16384, 1, env_i64("CUDF_AMD_GB_S_ITEMS", 1024));
      nitems              = static_cast<int32_t>(items);
      partial             = sc.alloc<uint64_t>(static_cast<size_t>(nitems) * ag.cap * PU);
      d_count             = sc.alloc<int32_t>(nitems);
      aa.input            = IN_COLUMNS;
      aa.seg              = SEG_ROW_CHUNKS;
      aa.nrows            = n;
      aa.chunk            = (n + items - 1) / items;
      aa.out_records      = partial;
      aa.out_count        = d_count;
      aa.nitems           = nitems;
      launch_aggregate(aa, sc.alloc<agg_args>(1), s);
      int const fan = 16;
      while (nitems > 1) {
        int32_t const next = (nitems + fan - 1) / fan;
        uint64_t* out      = sc.alloc<uint64_t>(static_cast<size_t>(next) * ag.cap * PU);
        int32_t* cnt       = sc.alloc<int32_t>(next);
        agg_args m         = aa;
        m.input            = IN_PARTIAL_RECORDS;
        m.seg              = SEG_STRIDED;
        m.records          = partial;
        m.src_count        = d_count;
        m.src_stride       = ag.cap;
        m.fan              = fan;
        m.nsrc             = nitems;
        m.out_records      = out;
        m.out_count        = cnt;
        m.nitems           = next;
        launch_aggregate(m, sc.alloc<agg_args>(1), s);
        partial = out;
        d_count = cnt;
        nitems  = next;
      }
    } else {
      final_cap = 0;
      // ---------------- path A: sorted / clustered keys (most rows are followed by a row of the same key). Every partition
      // scheme here gives a workgroup whole keys instead of a share of every key - regions overflow, rings stall, a wave's 64 rows
      // meet in one table slot (1B sorted rows on 1M groups: 74 ms against 7 ms uniform). Row chunks small enough to hold few
      // distinct keys are aggregated straight from the columns into one LDS table each (the single-pass kernel, wave-combined
      // accumulate); their partial records - about one per run of equal keys - then take the exact partition pipeline and are
      // merged. Reference: none (its global hash set does not care about row order).
      if (adjacent_equal >= 0.01 * static_cast<double>(env_i64("CUDF_AMD_GB_PREAGG_MIN_PCT", 90)) && !pre_failed && forced_p == 0 && p.narg == 0) {
        double const run_starts = std::max(1.0 - adjacent_equal, 1e-6);  // distinct keys of a chunk <= its run starts
        // (and at least ~2048 chunks: long runs would otherwise leave most CUs without a chunk - 200M sorted rows on 100K groups
        // ran on 48 workgroups, 5.1 ms)
        int64_t const chunk_rows = std::clamp<int64_t>(static_cast<int64_t>(static_cast<double>(ag.fill_limit) / 1.5 / run_starts), int64_t{1} << 14,
                                                       std::max<int64_t>(int64_t{1} << 14, n / 2048));
        int64_t const items      = (n + chunk_rows - 1) / chunk_rows;
        if (static_cast<double>(items) * ag.cap * PU * 8.0 <= 16.0 * 1024 * 1024 * 1024) {
          _last_path         = hash_path::PARTITIONED_LDS;
          uint64_t* partial1 = sc.alloc<uint64_t>(static_cast<size_t>(items) * ag.cap * PU);
          int32_t* d_count1  = sc.alloc<int32_t>(static_cast<size_t>(items));
          agg_args a1        = aa;
          a1.input           = IN_COLUMNS;
          a1.seg             = SEG_ROW_CHUNKS;
          a1.nrows           = n;
          a1.chunk           = chunk_rows;
          a1.out_records     = partial1;
          a1.out_count       = d_count1;
          a1.nitems          = static_cast<int32_t>(items);
          launch_aggregate(a1, sc.alloc<agg_args>(1), s);
          nitems  = static_cast<int32_t>(items);
          d_count = d_count1;
          int32_t const ov1 = overflow_and_counts();
          int64_t n2 = 0;
          for (int32_t c : h_count) n2 += c;
          if (env_i64("CUDF_AMD_DEBUG", 0))
            fprintf(stderr, "[cudf_amd] pre-aggregation: %.1f %% of the rows repeat their predecessor's key, %ld chunks of %ld rows -> %ld partial records, overflow=%d\n",
                    100.0 * adjacent_equal, (long)items, (long)chunk_rows, (long)n2, ov1);
          if (ov1 != 0 || n2 * 2 > n) {  // a chunk held too many keys, or nothing was gained: the ordinary paths
            pre_failed = true;
            sc.bufs.clear();
            d_overflow = sc.alloc<int32_t>(1);
            --attempt;
            continue;
          }
          // chunks' partial records -> one contiguous run (the partition kernels read segments of one buffer)
          int64_t* d_prefix1 = sc.alloc<int64_t>(static_cast<size_t>(items) + 1);
          launch_count_prefix(d_count1, static_cast<int32_t>(items), d_prefix1, s);
          uint64_t* compact = sc.alloc<uint64_t>(static_cast<size_t>(std::max<int64_t>(n2, 1)) * PU);
          launch_compact_records(partial1, ag.cap, d_prefix1, static_cast<int32_t>(items), PU, compact, s);
          int64_t* d_seg = sc.alloc<int64_t>(2);
          launch_store_i64x2(0, n2, d_seg, s);
          // tables for the merged groups
          double const need2 = std::min(est_groups * safety, static_cast<double>(n2)) / std::max(1.0, ag.cap * 0.25);
          auto pow2_up = [](double x) { int64_t v = 1; while (static_cast<double>(v) < x) v <<= 1; return v; };
          int64_t Q1 = std::clamp<int64_t>(pow2_up(need2), 16, 1024), Q2 = 1;
          if (need2 > 1024.0) {
            int64_t const tot = pow2_up(need2);
            Q1 = std::min<int64_t>(pow2_up(std::sqrt(static_cast<double>(tot))), 1024);
            Q2 = std::clamp<int64_t>(tot / Q1, 2, 1024);
          }
          int lq1 = 0, lq2 = 0;
          while ((int64_t{1} << lq1) < Q1) ++lq1;
          while ((int64_t{1} << lq2) < Q2) ++lq2;
          plan_dev p2 = p;  // what the partition kernels see: records of KU key units + NACC accumulator units
          p2.NPAY     = p.NACC;
          p2.simple   = 0;
          part_args pq{};
          pq.plan         = p2;
          pq.geom.nseg    = 1;
          pq.geom.slices  = static_cast<int32_t>(std::clamp<int64_t>(n2 / 16384, 16, 512));
          pq.geom.P       = static_cast<int32_t>(Q1);
          pq.geom.shift   = 64 - lq1;
          pq.geom.block   = 1024;
          pq.geom.tile_rows = 8 * 1024;
          pq.from_columns = 0;
          pq.in_records   = compact;
          pq.seg_offsets  = d_seg;
          exact_pipeline(pq, sc.alloc<part_args>(1), p2, n2, PU, IN_PARTIAL_RECORDS, Q1, Q2, lq1, lq2, aa);
          int32_t const ov2 = overflow_and_counts();
          if (ov2 == 0) break;
          escalate();  // a merged table overflowed: more tables (the chunks are aggregated again)
          continue;
        }
      }
      // ---------------- path D: dense integer keys -> direct-address LDS tables (no hash, no probe, no key words)
      // (heavy hitters in the sample: only the single-level ring scatter of one plain key takes them out of the partition)
      bool const ring_env = env_i64("CUDF_AMD_GB_DENSE_RING", 1) != 0 && env_i64("CUDF_AMD_GB_CHUNKED", 0) == 0;
      if (allow_dense && forced_p == 0 && (hot_keys.empty() || (dense_candidate && ring_env)) && !forced_exact) {
        dense_map dm{};
        bool dense_ok = dense_map_from_sample(dm);
        int bits = 14;
        while (bits < 31 && (uint64_t{1} << bits) < dm.range) ++bits;
        // ---- ring scatter (dense_ring_kernels.hip): 12-byte records in two streams, one or two levels of fan-out 16 ... 256, the
        // largest tables that fit (one 1024-thread aggregate workgroup per CU; a partition's regions are shared out to several
        // workgroups when there are fewer partitions than CUs)
        if (dense_ok && ring_env) {
          int rlog2P = 7;
          while (rlog2P < 17 && dense_table_bytes(p, 1 << std::max(bits - rlog2P, 0)) > 150 * 1024) ++rlog2P;
          if (env_i64("CUDF_AMD_GB_DENSE_LOG2P", 0) > 0) rlog2P = static_cast<int>(env_i64("CUDF_AMD_GB_DENSE_LOG2P", 0));
          bool const two_level = rlog2P > 8;
          int const l1 = two_level ? (rlog2P + 1) / 2 : rlog2P, l2 = rlog2P - l1;
          int const slots         = 1 << std::max(bits - rlog2P, 0);
          std::size_t const image = dense_table_bytes(p, slots);
          int32_t const ntables = static_cast<int32_t>(int64_t{1} << rlog2P);
          int const nsplit      = static_cast<int>(std::clamp<int64_t>(env_i64("CUDF_AMD_GB_DENSE_NSPLIT", 256 / ntables), 1, two_level ? 1 : 16));
          // (heavy hitters: one level, and their merged item - at most HOT_MAX_KEYS groups - must fit the stride of the items)
          bool const ring_ok = (hot_keys.empty() || (!two_level && slots / nsplit >= HOT_MAX_KEYS)) &&
                               dm.range <= (uint64_t{1} << bits) && bits <= 30 && bits - rlog2P >= 6 && rlog2P <= 16 && l1 >= 4 && l1 <= 8 &&
                               (l2 == 0 || (l2 >= 4 && l2 <= 8)) && image <= 150 * 1024 &&
                               static_cast<double>(dm.range) <= 8.0 * std::max(est_groups, 4096.0);
          if (ring_ok) {
            _last_path  = hash_path::DENSE_DIRECT;
            dm.mult     = 0x9E3779B1u;  // odd: index -> (index * mult) mod 2^bits is a bijection
            uint32_t inv = dm.mult;     // Newton: inv = mult^-1 mod 2^32
            for (int it = 0; it < 5; ++it) inv *= 2u - dm.mult * inv;
            dm.mult_inv = inv;
            dm.bits     = bits;
            dm.log2P    = rlog2P;
            int const DPU      = (dm.nkeys > 0 ? p.KU : 1) + p.NACC;  // units of a dumped partial record
            int64_t const PD = int64_t{1} << l1, P2D = int64_t{1} << l2, S = 256;
            auto region_cap_for = [&](double mean, double parts) {
              double const keys_per_p = std::max(1.0, 0.5 * est_groups / parts);
              double const rel_sigma  = std::sqrt(1.0 / keys_per_p + 1.0 / std::max(1.0, mean));
              return (static_cast<int64_t>(mean * (1.0 + 6.0 * std::min(rel_sigma, 1.0)) + 64.0) + 63) / 64 * 64;
            };
            // (workgroup w takes the 4096-row tiles w, w + S, ...: the busiest workgroup has ceil(tiles / S) of them)
            int64_t const ring_tile = 4 * 1024, wg_rows = std::min<int64_t>(n, ((n + ring_tile - 1) / ring_tile + S - 1) / S * ring_tile);
            int64_t const capR = region_cap_for(static_cast<double>(wg_rows) / static_cast<double>(PD), static_cast<double>(PD));
            dense_ring_args ra{};
            ra.plan         = p;
            ra.map          = dm;
            ra.from_columns = 1;
            ra.nrows        = n;
            ra.P            = static_cast<int32_t>(PD);
            ra.capl         = 13 - l1;
            ra.shift        = bits - l1;  // level 1: the top l1 bits of the scrambled index
            ra.slices       = static_cast<int32_t>(S);
            ra.nseg         = 1;
            ra.region_cap   = capR;
            ra.region_count = sc.alloc<int32_t>(static_cast<size_t>(S * PD));
            ra.overflow     = d_overflow;
            ra.out_val      = sc.alloc<uint64_t>(static_cast<size_t>(S * PD) * static_cast<size_t>(capR));
            ra.tag16        = two_level ? 0 : 1;
            ra.out_tag      = sc.alloc<uint16_t>(static_cast<size_t>(S * PD) * static_cast<size_t>(capR) * (two_level ? 2 : 1));
            bool const ring_hot = !hot_keys.empty();
            if (ring_hot) {  // (hot_eligible: one plain key, SUM / COUNT accumulators)
              uint64_t* d_hot = sc.alloc<uint64_t>(HOT_MAX_KEYS);
              CUDF_HIP_TRY(hipMemcpyAsync(d_hot, hot_keys.data(), hot_keys.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s));
              ra.hot_n     = static_cast<int32_t>(hot_keys.size());
              ra.hot_keys  = d_hot;
              ra.hot_out   = sc.alloc<uint64_t>(static_cast<size_t>(S) * HOT_SLOTS * PU);
              ra.hot_count = sc.alloc<int32_t>(static_cast<size_t>(S));
            }
            if (dm.nkeys > 0) {
              uint32_t* ones = sc.alloc<uint32_t>(16);
              CUDF_HIP_TRY(hipMemsetAsync(ones, 0xff, 64, s));
              ra.ones = ones;
            }
            dense_ring_args* d_ra = sc.alloc<dense_ring_args>(1);
            store_args(ra, d_ra, s);
            dense_agg_args da{};
            da.plan         = p;
            da.map          = dm;
            da.rec_val      = ra.out_val;
            da.rec_tag      = static_cast<uint16_t const*>(ra.out_tag);
            da.region_count = ra.region_count;
            da.region_cap   = capR;
            da.slices       = static_cast<int32_t>(S);
            int64_t cap2    = 0;
            dense_ring_args rb{};
            dense_ring_args* d_rb = nullptr;
            if (two_level) {
              // level 2: work item (g, s) reads level-1 partition g as the strided list of its regions s, s + slices2, ... and appends
              // to the regions of the global partitions g * P2 + d (the next l2 bits); the aggregate walks those
              int64_t const slices2 = std::max<int64_t>(1, 512 / PD);
              cap2 = region_cap_for(static_cast<double>(n) / static_cast<double>(PD * slices2 * P2D), static_cast<double>(PD * P2D));
              size_t const nreg2 = static_cast<size_t>(PD * P2D * slices2);
              rb                 = ra;
              rb.from_columns    = 0;
              rb.P               = static_cast<int32_t>(P2D);
              rb.capl            = 13 - l2;
              rb.shift           = bits - rlog2P;  // the l2 bits below the level-1 digit
              rb.slices          = static_cast<int32_t>(slices2);
              rb.nseg            = static_cast<int32_t>(PD);
              rb.in_val          = ra.out_val;
              rb.in_tag          = static_cast<uint32_t const*>(ra.out_tag);
              rb.in_region_count = ra.region_count;
              rb.in_region_cap   = capR;
              rb.in_slices       = static_cast<int32_t>(S);
              rb.region_cap      = cap2;
              rb.region_count    = sc.alloc<int32_t>(nreg2);
              rb.out_val         = sc.alloc<uint64_t>(nreg2 * static_cast<size_t>(cap2));
              rb.tag16           = 1;
              rb.out_tag         = sc.alloc<uint16_t>(nreg2 * static_cast<size_t>(cap2));
              d_rb               = sc.alloc<dense_ring_args>(1);
              store_args(rb, d_rb, s);
              da.rec_val      = rb.out_val;
              da.rec_tag      = static_cast<uint16_t const*>(rb.out_tag);
              da.region_count = rb.region_count;
              da.region_cap   = cap2;
              da.slices       = static_cast<int32_t>(slices2);
            }
            da.nsplit       = nsplit;
            da.slots        = slots;
            da.image_bytes  = static_cast<int32_t>(image);
            da.occ_acc      = dense_occ_acc(p);
            da.KU           = dm.nkeys > 0 ? p.KU : 1;
            nitems          = ntables * nsplit;  // (partial records: partition d's slots in nsplit shares)
            da.tables       = sc.alloc<uint64_t>(nsplit > 1 ? static_cast<size_t>(nitems) * image / 8 : 2);
            // (+ one item of the same stride for the merged heavy hitters)
            partial         = sc.alloc<uint64_t>((static_cast<size_t>(ntables) * slots + static_cast<size_t>(slots / nsplit)) * DPU);
            d_count         = sc.alloc<int32_t>(nitems + 1);
            da.out_records  = partial;
            da.out_count    = d_count;
            da.overflow     = d_overflow;
            da.nitems       = ntables;
            da.block        = 1024;
            dense_agg_args* d_da = sc.alloc<dense_agg_args>(1);
            store_args(da, d_da, s);
            launch_dense_ring_scatter(ra, d_ra, s);
            if (two_level) launch_dense_ring_scatter(rb, d_rb, s);
            launch_aggregate_dense(da, d_da, true, true, s);
            if (nsplit > 1) launch_dense_merge_dump(da, d_da, nsplit, s);
            final_cap          = slots / nsplit;
            if (ring_hot) {  // the workgroups' heavy-hitter partials -> one more item behind the tables' (hash-table merge kernel)
              CUDF_EXPECTS(final_cap >= HOT_MAX_KEYS, "dense keys: heavy-hitter item");
              agg_args hm{};
              hm.plan        = p;
              hm.geom        = ag;
              hm.overflow    = d_overflow;
              hm.input       = IN_PARTIAL_RECORDS;
              hm.seg         = SEG_STRIDED;
              hm.records     = ra.hot_out;
              hm.src_count   = ra.hot_count;
              hm.src_stride  = HOT_SLOTS;
              hm.fan         = static_cast<int32_t>(S);
              hm.nsrc        = static_cast<int32_t>(S);
              hm.out_records = partial + static_cast<size_t>(nitems) * final_cap * DPU;
              hm.out_count   = d_count + nitems;
              hm.nitems      = 1;
              launch_aggregate(hm, sc.alloc<agg_args>(1), s);
              nitems += 1;
            }
            int32_t const h_ov = overflow_and_counts();
            if (env_i64("CUDF_AMD_DEBUG", 0))
              fprintf(stderr, "[cudf_amd] dense keys (ring): nkeys=%d lo=%lld range=%llu bits=%d P=%ld x %ld slots=%d image=%zu B nsplit=%d capR=%ld cap2=%ld overflow=%d\n",
                      dm.nkeys, (long long)dm.lo, (unsigned long long)dm.range, bits, (long)PD, (long)P2D, slots, image, nsplit, (long)capR, (long)cap2, h_ov);
            if (h_ov == 0) break;
            // a region overflowed (skewed or clustered keys) or a key lay outside the sampled range: redo by hash
            allow_dense = false;
            final_cap   = 0;
            sc.bufs.clear();
            d_overflow = sc.alloc<int32_t>(1);
            --attempt;
            continue;
          }
        }
        double const unit_bytes = static_cast<double>(dense_table_bytes(p, 4096)) / 4096.0;  // LDS bytes per key of the range
        // One level: tables of ~32 KiB (two 1024-thread workgroups per CU), 256 to 1024 of them. A range that needs more than
        // 1024 tables of 150 KiB takes two levels (P1 x P2) with the largest tables that fit.
        int log2P = 8;
        while (log2P < 10 && std::ldexp(unit_bytes, bits - log2P) > 32.0 * 1024) ++log2P;
        while (log2P < 20 && std::ldexp(unit_bytes, bits - log2P) > 150.0 * 1024) ++log2P;
        if (env_i64("CUDF_AMD_GB_DENSE_LOG2P", 0) > 0) log2P = static_cast<int>(env_i64("CUDF_AMD_GB_DENSE_LOG2P", 0));
        bool const two_level      = log2P > 10;
        int const log2P1          = two_level ? (log2P + 1) / 2 : log2P;
        int const log2P2          = log2P - log2P1;
        int const slots           = 1 << std::max(bits - log2P, 0);
        std::size_t const image   = dense_table_bytes(p, slots);
        dense_ok = dense_ok && dm.range <= (uint64_t{1} << bits) && bits <= 30 && bits - log2P >= 6 && log2P <= 20 && image <= 150 * 1024 &&
                   static_cast<double>(dm.range) <= 8.0 * std::max(est_groups, 4096.0) &&
                   // (the carried table images of a chunked single-level call travel to LDS and back once per chunk)
                   (two_level || static_cast<double>(image) * (1 << log2P) <= 32.0 * 1024 * 1024 || env_i64("CUDF_AMD_GB_CHUNKED", 0) == 0);
        int64_t const PD = int64_t{1} << log2P1, P2D = int64_t{1} << log2P2;
        // scatter workgroups: one of 1024 threads per CU (128-byte granules up to 512 partitions), or - CUDF_AMD_GB_SCATTER_BLOCK=512 -
        // two of 512 threads per CU with 64-byte granules (measured slower: profiles/r2_mall_pipeline.txt)
        int const SB     = env_i64("CUDF_AMD_GB_SCATTER_BLOCK", 1024) == 512 ? 512 : 1024;
        int const GD     = static_cast<int>(env_i64("CUDF_AMD_GB_WC_G", (PD > 512 || SB == 512) ? 4 : 8));
        if (dense_ok && partition_wc_fits(2, static_cast<int>(PD), GD, SB) && (!two_level || partition_wc_fits(2, static_cast<int>(P2D), P2D > 512 ? 4 : 8))) {
          _last_path  = hash_path::DENSE_DIRECT;
          dm.mult     = 0x9E3779B1u;  // odd: index -> (index * mult) mod 2^bits is a bijection
          uint32_t inv = dm.mult;     // Newton: inv = mult^-1 mod 2^32
          for (int it = 0; it < 5; ++it) inv *= 2u - dm.mult * inv;
          dm.mult_inv = inv;
          dm.bits     = bits;
          dm.log2P    = log2P;
          int const DPU = (dm.nkeys > 0 ? p.KU : 1) + p.NACC;  // units of a dumped partial record
          // chunks (single level, opt-in): a multiple of one tile per workgroup; the ring of a chunk's regions stays in the
          // Infinity Cache. Measured: no gain (profiles/r2_mall_pipeline.txt); CUDF_AMD_GB_CHUNKED=1 keeps it testable.
          int64_t const S        = 256 * (1024 / SB), tile_rows = 5 * SB;
          int64_t const quantum  = S * tile_rows;
          int64_t const want     = std::max<int64_t>(quantum, env_i64("CUDF_AMD_GB_CHUNK_ROWS", 8 * quantum));
          int64_t const nchunks  = (!two_level && env_i64("CUDF_AMD_GB_CHUNKED", 0) != 0) ? std::max<int64_t>(1, (n + want - 1) / want) : 1;
          int64_t const C        = ((n + nchunks - 1) / nchunks + quantum - 1) / quantum * quantum;
          double const cell_mean = static_cast<double>(std::min(C, n)) / static_cast<double>(S * PD);
          double const keys_per_p = std::max(1.0, 0.5 * est_groups / static_cast<double>(PD));
          double const rel_sigma  = std::sqrt(1.0 / keys_per_p + 1.0 / std::max(1.0, cell_mean));
          int64_t const capR      = (static_cast<int64_t>(cell_mean * (1.0 + 6.0 * std::min(rel_sigma, 1.0)) + 16.0) + 7) / 8 * 8;
          part_args pa{};
          pa.plan          = p;
          pa.geom.nseg     = 1;
          pa.geom.slices   = static_cast<int32_t>(S);
          pa.geom.P        = static_cast<int32_t>(PD);
          pa.geom.shift    = bits - log2P1;  // level 1: the top log2P1 bits of the scrambled index
          pa.geom.block    = SB;
          pa.geom.tile_rows = static_cast<int32_t>(tile_rows);
          pa.from_columns  = 1;
          pa.nrows         = n;
          pa.optimistic    = 1;
          pa.region_cap    = capR;
          pa.region_count  = sc.alloc<int32_t>(static_cast<size_t>(S * PD));
          pa.overflow      = d_overflow;
          pa.out_records   = sc.alloc<uint64_t>(static_cast<size_t>(S * PD) * static_cast<size_t>(capR) * 2);
          pa.wc_granule    = GD;
          pa.cyclic_tiles  = 1;
          pa.use_dense     = 1;
          pa.dense         = dm;
          part_args* d_pa  = sc.alloc<part_args>(1);
          store_args(pa, d_pa, s);
          dense_agg_args da{};
          da.plan         = p;
          da.map          = dm;
          da.records      = pa.out_records;
          da.region_count = pa.region_count;
          da.region_cap   = capR;
          da.slices       = static_cast<int32_t>(S);
          int64_t cap2    = 0;
          part_args pb{};
          part_args* d_pb = nullptr;
          if (two_level) {
            // level 2: work item (g, s) reads level-1 partition g as the strided list of its regions s, s + slices2, ... and appends
            // to the regions of the global partitions g * P2 + d (the next log2P2 bits); the aggregate walks those
            int64_t const slices2 = std::max<int64_t>(1, 512 / PD);
            double const mean2    = static_cast<double>(n) / static_cast<double>(PD * slices2 * P2D);
            double const sigma2   = std::sqrt(1.0 / std::max(1.0, 0.5 * est_groups / static_cast<double>(PD * P2D)) + 1.0 / std::max(1.0, mean2));
            cap2                  = (static_cast<int64_t>(mean2 * (1.0 + 6.0 * std::min(sigma2, 1.0)) + 16.0) + 7) / 8 * 8;
            pb.plan            = p;
            pb.geom.nseg       = static_cast<int32_t>(PD);
            pb.geom.slices     = static_cast<int32_t>(slices2);
            pb.geom.P          = static_cast<int32_t>(P2D);
            pb.geom.shift      = bits - log2P;  // the log2P2 bits below the level-1 digit
            pb.geom.block      = 1024;
            pb.geom.tile_rows  = 5 * 1024;
            pb.from_columns    = 0;
            pb.in_records      = pa.out_records;
            pb.from_regions    = 1;
            pb.in_region_count = pa.region_count;
            pb.in_region_cap   = capR;
            pb.in_slices       = static_cast<int32_t>(S);
            pb.optimistic      = 1;
            pb.region_cap      = cap2;
            size_t const nreg2 = static_cast<size_t>(PD * P2D * slices2);
            pb.region_count    = sc.alloc<int32_t>(nreg2);
            pb.overflow        = d_overflow;
            pb.out_records     = sc.alloc<uint64_t>(nreg2 * static_cast<size_t>(cap2) * 2);
            pb.wc_granule      = P2D > 512 ? 4 : 8;
            pb.use_dense       = 1;
            pb.dense           = dm;
            d_pb               = sc.alloc<part_args>(1);
            store_args(pb, d_pb, s);
            da.records      = pb.out_records;
            da.region_count = pb.region_count;
            da.region_cap   = cap2;
            da.slices       = static_cast<int32_t>(slices2);
          }
          da.slots        = slots;
          da.image_bytes  = static_cast<int32_t>(image);
          da.occ_acc      = dense_occ_acc(p);
          da.KU           = dm.nkeys > 0 ? p.KU : 1;
          nitems          = static_cast<int32_t>(int64_t{1} << log2P);
          da.tables       = sc.alloc<uint64_t>(nchunks > 1 ? static_cast<size_t>(nitems) * image / 8 : 2);
          partial         = sc.alloc<uint64_t>(static_cast<size_t>(nitems) * slots * DPU);
          d_count         = sc.alloc<int32_t>(nitems);
          da.out_records  = partial;
          da.out_count    = d_count;
          da.overflow     = d_overflow;
          da.nitems       = nitems;
          da.block        = 1024;
          dense_agg_args* d_da = sc.alloc<dense_agg_args>(1);
          store_args(da, d_da, s);
          for (int64_t c = 0; c < nchunks; ++c) {
            chunk_range const cr{c * C, std::min(n, (c + 1) * C)};
            launch_partition_scatter(pa, d_pa, s, nchunks > 1 ? cr : chunk_range{0, 0});
            if (two_level) launch_partition_scatter(pb, d_pb, s);
            launch_aggregate_dense(da, d_da, c == 0, c == nchunks - 1, s);
          }
          final_cap          = slots;
          int32_t const h_ov = overflow_and_counts();
          if (env_i64("CUDF_AMD_DEBUG", 0))
            fprintf(stderr, "[cudf_amd] dense keys: nkeys=%d lo=%lld range=%llu bits=%d P=%ld x %ld slots=%d image=%zu B chunks=%ld capR=%ld cap2=%ld overflow=%d\n",
                    dm.nkeys, (long long)dm.lo, (unsigned long long)dm.range, bits, (long)PD, (long)P2D, slots, image, (long)nchunks, (long)capR, (long)cap2, h_ov);
          if (h_ov == 0) break;
          // a region overflowed (skewed or clustered keys) or a key lay outside the sampled range: redo by hash
          allow_dense = false;
          final_cap   = 0;
          sc.bufs.clear();
          d_overflow = sc.alloc<int32_t>(1);
          --attempt;
          continue;
        }
      }
      // ---------------- path P: radix-partition raw records on hash bits, then one LDS table per partition
      _last_path = hash_path::PARTITIONED_LDS;
      auto pow2_at_least = [](double x) {
        int64_t v = 1;
        while (static_cast<double>(v) < x) v <<= 1;
        return v;
      };
      int64_t const maxP1 = 1024;  // LDS: 8192-row stage + P * 12 B + pid must fit 160 KiB
      int64_t P1 = forced_p ? forced_p : std::clamp<int64_t>(pow2_at_least(need), 256, maxP1);
      int64_t P2 = 1;
      if (!forced_p && need > static_cast<double>(maxP1)) {
        int64_t const tot = pow2_at_least(need);
        P1 = pow2_at_least(std::sqrt(static_cast<double>(tot)));
        P2 = tot / P1;
        P1 = std::min<int64_t>(P1, maxP1);
        P2 = std::clamp<int64_t>(P2, 2, maxP1);
      }
      int log2P1 = 0, log2P2 = 0;
      while ((int64_t{1} << log2P1) < P1) ++log2P1;
      while ((int64_t{1} << log2P2) < P2) ++log2P2;

      // granule (records) of the write-combining scatter for a fan-out, 0 = run-per-tile kernel: 64-byte granules for
      // 16-byte records at P = 1024 (a 128-byte carry area would not leave room for a tile), else 128-256 bytes
      auto wc_granule_for = [&](int64_t P) -> int32_t {
        if (!env_i64("CUDF_AMD_GB_WC", 1)) return 0;
        int const G = RU == 2 ? static_cast<int>(env_i64("CUDF_AMD_GB_WC_G", P > 512 ? 4 : 8)) : (RU == 4 ? 4 : 8);
        if (partition_wc_fits(RU, static_cast<int>(P), G)) return G;
        return (RU == 3 && partition_wc_fits(RU, static_cast<int>(P), 4)) ? 4 : 0;  // 24-byte records: 96-byte granules
      };
      part_args pa{};
      pa.plan         = p;
      pa.geom.nseg    = 1;
      // optimistic: one persistent workgroup per CU (longer regions for the aggregate); exact: 2 per CU (the
      // histogram pass wants the parallelism: 1.6 ms at 512 slices vs 2.7 ms at 256)
      bool const will_try_optimistic = allow_optimistic && P2 == 1 && !forced_exact && n >= (int64_t{1} << 22);
      // (small inputs: one slice per 16K rows - the single-workgroup scan walks slices x P counters)
      pa.geom.slices  = static_cast<int32_t>(env_i64("CUDF_AMD_GB_SLICES", will_try_optimistic ? 256 : std::clamp<int64_t>(n / 16384, 16, 512)));
      pa.geom.P       = static_cast<int32_t>(P1);
      pa.geom.shift   = 64 - log2P1;
      pa.geom.block   = 1024;
      pa.geom.tile_rows = static_cast<int32_t>(env_i64("CUDF_AMD_GB_RPT", 8)) * 1024;
      pa.from_columns = 1;
      pa.nrows        = n;
      size_t const items1 = static_cast<size_t>(pa.geom.slices);
      part_args* d_pa     = sc.alloc<part_args>(1);
      // single-level partitions of big inputs: try the optimistic single-pass partition first
      // Region sizing: rows of a (slice, partition) cell = sum over the ~G/P keys of the partition of their rows in
      // the slice; its relative spread has a key-count part 1/sqrt(G/P) (which keys hash there) and a row-sampling
      // part 1/sqrt(mean). Six sigmas of slack; if that needs more than 2x the memory, use the exact pipeline.
      double const cell_mean   = static_cast<double>(n) / static_cast<double>(items1) / static_cast<double>(P1);
      // (half the estimated key count: an over-estimate would under-size the regions)
      double const keys_per_p  = std::max(1.0, 0.5 * est_groups / static_cast<double>(P1));
      double const rel_sigma   = std::sqrt(1.0 / keys_per_p + 1.0 / std::max(1.0, cell_mean));
      bool const optimistic = allow_optimistic && P2 == 1 && !forced_exact && n >= (int64_t{1} << 22) && 6.0 * rel_sigma <= 1.0;
      uint64_t* recA = nullptr;
      if (optimistic) {
        int64_t const capR  = (static_cast<int64_t>(cell_mean * (1.0 + 6.0 * rel_sigma) + 16.0) + 7) / 8 * 8;
        pa.optimistic       = 1;
        pa.region_cap       = capR;
        pa.region_count     = sc.alloc<int32_t>(items1 * P1);
        pa.overflow         = d_overflow;
        recA                = sc.alloc<uint64_t>(items1 * static_cast<size_t>(P1) * static_cast<size_t>(capR) * RU);
        pa.out_records      = recA;
        // 16-byte records: write-combining scatter (whole aligned granules only); 64-byte granules at P = 1024
        // (the carry area of 128-byte granules would not leave room for a tile), 128-byte granules at P <= 512
        pa.wc_granule = wc_granule_for(P1);
        pa.cyclic_tiles = pa.wc_granule != 0 && env_i64("CUDF_AMD_GB_CYCLIC", 1) != 0;
        bool const hot = setup_hot(pa, P1);
        if (env_i64("CUDF_AMD_GB_STAMPS", 0)) pa.stamps = sc.alloc<unsigned long long>(items1 * 8);
        store_args(pa, d_pa, s);
        launch_partition_scatter(pa, d_pa, s);
        if (pa.stamps != nullptr) {
          std::vector<unsigned long long> h(items1 * 8);
          CUDF_HIP_TRY(hipMemcpyAsync(h.data(), pa.stamps, h.size() * 8, hipMemcpyDeviceToHost, s));
          CUDF_HIP_TRY(hipStreamSynchronize(s));
          double tot[8] = {0};
          for (size_t w = 0; w < items1; ++w) for (int i = 0; i < 8; ++i) tot[i] += static_cast<double>(h[w * 8 + i]);
          double all = 0; for (double t : tot) all += t;
          fprintf(stderr, "[cudf_amd] scatter phase shares (wave 0 of each WG, s_memtime): rank %.1f%% | barrier %.1f%% | scan %.1f%% | stage %.1f%% | prefetch-issue %.1f%% | barrier %.1f%% | write-out %.1f%% | barrier %.1f%%  (avg cycles/WG %.0f)\n",
                  100 * tot[0] / all, 100 * tot[1] / all, 100 * tot[2] / all, 100 * tot[3] / all, 100 * tot[4] / all, 100 * tot[5] / all, 100 * tot[6] / all, 100 * tot[7] / all, all / items1);
        }
        if (env_i64("CUDF_AMD_DEBUG", 0)) { CUDF_HIP_TRY(hipStreamSynchronize(s)); fprintf(stderr, "[cudf_amd] optimistic scatter done capR=%ld P=%ld slices=%zu\n", (long)capR, (long)P1, items1); }
        nitems         = static_cast<int32_t>(P1);
        partial        = sc.alloc<uint64_t>(static_cast<size_t>(nitems + 1) * ag.cap * PU);
        d_count        = sc.alloc<int32_t>(nitems + 1);
        aa.input       = IN_RAW_RECORDS;
        aa.seg         = SEG_STRIDED;
        aa.records     = recA;
        aa.src_count   = pa.region_count;
        aa.src_stride  = capR;
        aa.fan         = pa.geom.slices;
        aa.nsrc        = static_cast<int32_t>(items1 * P1);
        aa.out_records = partial;
        aa.out_count   = d_count;
        aa.nitems      = nitems;
        launch_aggregate(aa, sc.alloc<agg_args>(1), s);
        if (hot) merge_hot(pa, aa);
        if (env_i64("CUDF_AMD_DEBUG", 0)) { CUDF_HIP_TRY(hipStreamSynchronize(s)); fprintf(stderr, "[cudf_amd] optimistic aggregate done\n"); }

        int32_t const h_ov = overflow_and_counts();
        if (env_i64("CUDF_AMD_DEBUG", 0)) fprintf(stderr, "[cudf_amd] optimistic overflow flag = %d\n", h_ov);
        if (h_ov == 0) break;
        if ((h_ov & 1) == 0) {  // the regions held, a table overflowed: more tables, still without a histogram pass
          escalate();
          continue;
        }
        // a region overflowed (skewed keys): redo with exact offsets
        allow_optimistic = false;
        sc.bufs.clear();
        d_overflow = sc.alloc<int32_t>(1);
        --attempt;
        continue;
      }
      // two optimistic levels (more than 1024 partitions): level 1 as above into per-(slice, partition) regions; the
      // level-2 items read their level-1 partition as a strided list of those regions and write per-(item, final
      // partition) regions, which the aggregate walks. No histogram pass on either level.
      // (a full heavy-hitter list means many more warm keys behind it: each overflows a level-2 region, whose share of the
      // rows is 1 / (P1 * slices2 * P2) - the attempt would be wasted, go to exact offsets at once)
      bool const warm_tail = hot_keys.size() >= static_cast<std::size_t>(HOT_MAX_KEYS);
      if (allow_optimistic && P2 > 1 && !forced_exact && !warm_tail && n >= (int64_t{1} << 22) && env_i64("CUDF_AMD_GB_OPTIMISTIC2", 1)) {
        int64_t const S1      = 256;
        int64_t const slices2 = std::max<int64_t>(1, 512 / P1);
        double const mean1    = static_cast<double>(n) / static_cast<double>(S1 * P1);
        double const sigma1   = std::sqrt(1.0 / std::max(1.0, 0.5 * est_groups / static_cast<double>(P1)) + 1.0 / std::max(1.0, mean1));
        double const mean2    = static_cast<double>(n) / static_cast<double>(P1 * slices2 * P2);
        double const sigma2   = std::sqrt(1.0 / std::max(1.0, 0.5 * est_groups / static_cast<double>(P1 * P2)) + 1.0 / std::max(1.0, mean2));
        // (very many tiny tables: the per-table partial buffers dominate the memory; keep to one attempt there)
        bool const partials_fit = static_cast<double>(P1 * P2) * ag.cap * PU * 8.0 <= 32.0 * 1024 * 1024 * 1024;
        if (6.0 * sigma1 <= 1.0 && 6.0 * sigma2 <= 1.0 && S1 / slices2 <= 256 && partials_fit) {
          int64_t const cap1 = (static_cast<int64_t>(mean1 * (1.0 + 6.0 * sigma1) + 16.0) + 7) / 8 * 8;
          int64_t const cap2 = (static_cast<int64_t>(mean2 * (1.0 + 6.0 * sigma2) + 16.0) + 7) / 8 * 8;
          pa.geom.slices     = static_cast<int32_t>(S1);
          pa.optimistic      = 1;
          pa.region_cap      = cap1;
          pa.region_count    = sc.alloc<int32_t>(static_cast<size_t>(S1 * P1));
          pa.overflow        = d_overflow;
          recA               = sc.alloc<uint64_t>(static_cast<size_t>(S1 * P1) * static_cast<size_t>(cap1) * RU);
          pa.out_records     = recA;
          pa.wc_granule      = wc_granule_for(P1);
          pa.cyclic_tiles    = pa.wc_granule != 0 && env_i64("CUDF_AMD_GB_CYCLIC", 1) != 0;
          bool const hot     = setup_hot(pa, P1);  // (a key with percents of the rows would leave one table's workgroup alone with them)
          store_args(pa, d_pa, s);
          launch_partition_scatter(pa, d_pa, s);
          part_args pb{};
          pb.plan            = p;
          pb.geom.nseg       = static_cast<int32_t>(P1);
          pb.geom.slices     = static_cast<int32_t>(slices2);
          pb.geom.P          = static_cast<int32_t>(P2);
          pb.geom.shift      = 64 - log2P1 - log2P2;
          pb.geom.block      = 1024;
          pb.geom.tile_rows  = pa.geom.tile_rows;
          pb.from_columns    = 0;
          pb.in_records      = recA;
          pb.from_regions    = 1;
          pb.in_region_count = pa.region_count;
          pb.in_region_cap   = cap1;
          pb.in_slices       = static_cast<int32_t>(S1);
          pb.optimistic      = 1;
          pb.region_cap      = cap2;
          size_t const nreg2 = static_cast<size_t>(P1 * P2 * slices2);
          pb.region_count    = sc.alloc<int32_t>(nreg2);
          pb.overflow        = d_overflow;
          uint64_t* recB     = sc.alloc<uint64_t>(nreg2 * static_cast<size_t>(cap2) * RU);
          pb.out_records     = recB;
          pb.wc_granule      = wc_granule_for(P2);
          part_args* d_pb = sc.alloc<part_args>(1);
          store_args(pb, d_pb, s);
          launch_partition_scatter(pb, d_pb, s);
          nitems         = static_cast<int32_t>(P1 * P2);
          partial        = sc.alloc<uint64_t>(static_cast<size_t>(nitems + 1) * ag.cap * PU);
          d_count        = sc.alloc<int32_t>(nitems + 1);
          aa.input       = IN_RAW_RECORDS;
          aa.seg         = SEG_STRIDED;
          aa.records     = recB;
          aa.src_count   = pb.region_count;
          aa.src_stride  = cap2;
          aa.fan         = static_cast<int32_t>(slices2);
          aa.nsrc        = static_cast<int32_t>(nreg2);
          aa.out_records = partial;
          aa.out_count   = d_count;
          aa.nitems      = nitems;
          launch_aggregate(aa, sc.alloc<agg_args>(1), s);
          if (hot) merge_hot(pa, aa);
          int32_t const h_ov = overflow_and_counts();
          if (env_i64("CUDF_AMD_DEBUG", 0))
            fprintf(stderr, "[cudf_amd] two-level optimistic P1=%ld P2=%ld slices2=%ld cap1=%ld cap2=%ld RU=%d wc=%d/%d overflow=%d\n",
                    (long)P1, (long)P2, (long)slices2, (long)cap1, (long)cap2, RU, pa.wc_granule, pb.wc_granule, h_ov);
          if (h_ov == 0) break;
          if ((h_ov & 1) == 0) {  // the regions held, a table overflowed
            escalate();
            continue;
          }
          allow_optimistic = false;  // a region overflowed: redo with exact offsets
          sc.bufs.clear();
          d_overflow = sc.alloc<int32_t>(1);
          --attempt;
          continue;
        }
      }
      exact_pipeline(pa, d_pa, p, n, RU, IN_RAW_RECORDS, P1, P2, log2P1, log2P2, aa);
    }
    int32_t const h_ov_exact = overflow_and_counts();
    if (h_ov_exact == 0) break;
    if (env_i64("CUDF_AMD_DEBUG", 0))
      fprintf(stderr, "[cudf_amd] attempt %d: overflow flag %d with %d tables of %d slots (fill limit %d) for an estimate of %.0f groups in %ld rows\n",
              attempt, h_ov_exact, nitems, ag.cap, ag.fill_limit, est_groups, (long)n);
    escalate();
  }

  // ---- group counts -> prefix (on the device, from the counts the attempt left there: the host only needs the total)
  int64_t G = 0;
  for (int i = 0; i < nitems; ++i) G += h_count[i];
  CUDF_EXPECTS(G <= std::numeric_limits<size_type>::max(), "group count exceeds size_type");
  int64_t* d_prefix = sc.alloc<int64_t>(nitems + 1);
  launch_count_prefix(d_count, nitems, d_prefix, s);

  // ---- output columns
  finalize_args fa{};
  fa.plan           = p;
  finalize_dev& fin = fa.fin;
  std::vector<std::unique_ptr<column>> key_cols;
  std::vector<std::unique_ptr<column>> res_cols;
  int const nres = static_cast<int>(hp.results.size());
  CUDF_EXPECTS(p.nkeycols + nres <= MAX_OUT, "Too many output columns for one call (limit 40).");
  (void)nres;
  int32_t* d_nulls = sc.alloc<int32_t>(MAX_OUT);
  CUDF_HIP_TRY(hipMemsetAsync(d_nulls, 0, sizeof(int32_t) * MAX_OUT, s));
  auto make_out = [&](data_type t, bool nullable) {
    auto col = std::make_unique<column>(t, static_cast<size_type>(G),
                                        rmm::device_buffer{static_cast<size_t>(G) * size_of(t), s, mr},
                                        nullable ? create_null_mask(static_cast<size_type>(G), mask_state::UNINITIALIZED, stream, mr)
                                                 : rmm::device_buffer{},
                                        0);
    return col;
  };
  for (int c = 0; c < p.nkeycols; ++c) {
    auto const& kc = _keys.column(c);
    auto col       = make_out(kc.type(), kc.nullable());
    out_desc d{};
    auto mv       = col->mutable_view();
    d.data        = mv.head();
    d.mask        = kc.nullable() ? mv.null_mask() : nullptr;
    d.null_count  = d_nulls + fin.nout;
    d.kind        = OUT_KEY;
    d.a0          = static_cast<int8_t>(c);
    d.width       = static_cast<int8_t>(p.cols[c].width);
    d.key_unit    = static_cast<int8_t>(hp.key_unit[c]);
    d.key_full    = hp.key_half[c] == 2;
    d.key_hi      = hp.key_half[c] == 1;
    d.key_null_bit  = (hp.keynulls_unit >= 0 && kc.has_nulls()) ? static_cast<int8_t>(c) : int8_t{-1};
    d.keynulls_unit = static_cast<int8_t>(hp.keynulls_unit);
    d.keynulls_hi   = static_cast<int8_t>(hp.keynulls_hi);
    d.valid_acc     = -1;
    d.key_acc       = static_cast<int8_t>(hp.key_acc[c]);
    fin.out[fin.nout++] = d;
    key_cols.push_back(std::move(col));
  }
  std::vector<int> res_desc;  // result column -> the out_desc whose null count is the column's
  for (auto const& rs : hp.results) {
    res_desc.push_back(fin.nout);
    if (rs.kind == aggregation::SUM_OVERFLOW) {
      // struct {sum: source type, overflow: BOOL8}; the children carry no masks, the struct's mask is the validity of the group's
      // sum (reference groupby/hash/output_utils.cu:83-111)
      auto const vtype = hp.value_cols[rs.value_idx].type();
      auto sum_col     = make_out(vtype, false);
      auto flag_col    = make_out(data_type{type_id::BOOL8}, false);
      rmm::device_buffer smask = rs.nullable ? create_null_mask(static_cast<size_type>(G), mask_state::UNINITIALIZED, stream, mr) : rmm::device_buffer{};
      CUDF_EXPECTS(fin.nout + 2 <= MAX_OUT, "Too many output columns for one call (limit 40).");
      for (int part = 0; part < 2; ++part) {
        out_desc d{};
        auto mv       = (part == 0 ? sum_col : flag_col)->mutable_view();
        d.data        = mv.head();
        d.mask        = (part == 0 && rs.nullable) ? static_cast<bitmask_type*>(smask.data()) : nullptr;
        d.null_count  = d_nulls + fin.nout;
        d.kind        = part == 0 ? OUT_SUMOV_SUM : OUT_SUMOV_FLAG;
        d.a0          = static_cast<int8_t>(rs.a0);
        d.a1          = -1;
        d.a2          = static_cast<int8_t>(rs.a2);
        d.valid_acc   = static_cast<int8_t>(rs.valid_acc);
        d.cls         = static_cast<int8_t>(CLS_SINT);
        d.width       = static_cast<int8_t>(part == 0 ? size_of(vtype) : 1);
        d.out_cls     = static_cast<int8_t>(part == 0 ? CLS_SINT : CLS_BOOL);
        d.key_unit    = static_cast<int8_t>(size_of(vtype));  // (source width)
        d.key_null_bit = -1;
        d.key_acc      = -1;
        fin.out[fin.nout++] = d;
      }
      std::vector<std::unique_ptr<column>> children;
      children.push_back(std::move(sum_col));
      children.push_back(std::move(flag_col));
      res_cols.push_back(std::make_unique<column>(data_type{type_id::STRUCT}, static_cast<size_type>(G), rmm::device_buffer{}, std::move(smask), 0,
                                                  std::move(children)));
      continue;
    }
    auto col = make_out(rs.target, rs.nullable);
    out_desc d{};
    auto mv      = col->mutable_view();
    d.data       = mv.head();
    d.mask       = rs.nullable ? mv.null_mask() : nullptr;
    d.null_count = d_nulls + fin.nout;
    d.kind       = (rs.kind == aggregation::COUNT_VALID || rs.kind == aggregation::COUNT_ALL) ? OUT_COUNT
                   : rs.kind == aggregation::MEAN ? (rs.target.id() == type_id::FLOAT64 ? OUT_MEAN : OUT_MEAN_INT)
                   : rs.kind == aggregation::M2                                                ? OUT_M2
                   : rs.kind == aggregation::VARIANCE                                          ? OUT_VAR
                   : rs.kind == aggregation::STD                                               ? OUT_STD
                                                                                               : OUT_ACC;
    d.a0         = static_cast<int8_t>(rs.a0);
    d.a1         = static_cast<int8_t>(rs.a1);
    d.a2         = static_cast<int8_t>(rs.a2);
    d.ddof       = static_cast<int8_t>(rs.ddof);
    d.valid_acc  = static_cast<int8_t>(rs.valid_acc);
    d.cls        = static_cast<int8_t>(rs.acc_cls);
    d.width      = static_cast<int8_t>(size_of(rs.target));
    d.out_cls    = static_cast<int8_t>(class_of(rs.target.id()));
    d.key_null_bit = -1;
    d.key_acc      = -1;
    fin.out[fin.nout++] = d;
    res_cols.push_back(std::move(col));
  }
  launch_finalize(fa, sc.alloc<finalize_args>(1), partial, final_cap > 0 ? final_cap : ag.cap, d_prefix, nitems, G, s);
  // the null counts come back with one more synchronisation - only if some output column can hold a null at all (the results are
  // stream-ordered like every libcudf result: the caller synchronises before it reads them on another stream or on the host)
  bool any_nullable = false;
  for (int o = 0; o < fin.nout; ++o) any_nullable = any_nullable || fin.out[o].mask != nullptr;
  trace.mark("finalize queued");
  if (any_nullable) {
    int32_t* const h_nulls = pinned_ints(MAX_OUT);
    CUDF_HIP_TRY(hipMemcpyAsync(h_nulls, d_nulls, sizeof(int32_t) * MAX_OUT, hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    trace.mark("finalize back");
    int oc = 0;
    for (auto& k : key_cols) k->set_null_count(h_nulls[oc++]);
    for (std::size_t r = 0; r < res_cols.size(); ++r) res_cols[r]->set_null_count(h_nulls[res_desc[r]]);
  }

  // ---- hand results back in request order; every (column, aggregation) pair has its own column, so a
  // repeated pair needs no cache deep copy (reference groupby/common/utils.hpp:39-51 copies instead).
  std::vector<aggregation_result> results;
  size_t ri = 0;
  for (auto const& r : requests) {
    aggregation_result ar;
    for (size_t j = 0; j < r.aggregations.size(); ++j) ar.results.push_back(std::move(res_cols[ri++]));
    results.push_back(std::move(ar));
  }
  return {std::make_unique<table>(std::move(key_cols)), std::move(results)};
}

}  // namespace groupby
}  // namespace cudf
