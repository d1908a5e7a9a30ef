// SPDX-License-Identifier: Apache-2.0
// The plan of a hash-groupby call: record layout (key units, payload units), accumulators, result columns; the planner's
// switches. Reference counterparts: cpp/src/groupby/hash/extract_single_pass_aggs.cpp:26-177 (flattening of the requests into
// single-pass aggregations), groupby/common/utils.hpp:66-85 (the hashable set).
#include "call.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace cudf::groupby::detail {

using cudf::detail::CLS_BOOL;
using cudf::detail::CLS_F32;
using cudf::detail::CLS_F64;
using cudf::detail::CLS_SINT;
using cudf::detail::CLS_UINT;
using cudf::detail::class_of;

namespace {
int64_t env_i64(char const* name, int64_t dflt)
{
  char const* e = std::getenv(name);
  return (e != nullptr && *e != 0) ? std::strtoll(e, nullptr, 10) : dflt;
}
}  // namespace

planner_env planner_env::load()
{
  planner_env e{};
  e.lds_kb            = env_i64("CUDF_AMD_GB_LDS_KB", 159);  // 160 KiB minus the kernels' static words
  e.agg_block         = env_i64("CUDF_AMD_GB_AGG_BLOCK", 1024);
  e.big_min_rows      = env_i64("CUDF_AMD_GB_BIG_MIN_ROWS", int64_t{1} << 22);
  e.sample_div        = env_i64("CUDF_AMD_GB_SAMPLE_DIV", 16);
  // (the one-table path T needs no partition pass: it pays from far fewer rows - 1M rows on 1000 groups 260 -> ~150 us)
  e.one_table_min_rows = std::min(e.big_min_rows, env_i64("CUDF_AMD_GB_ONE_TABLE_MIN_ROWS", int64_t{1} << 17));
  e.estimate_min_rows = env_i64("CUDF_AMD_GB_ESTIMATE_MIN_ROWS", 1 << 16);
  e.forced_p          = env_i64("CUDF_AMD_GB_P", 0);
  e.s_items           = env_i64("CUDF_AMD_GB_S_ITEMS", 1024);
  e.preagg_min_pct    = env_i64("CUDF_AMD_GB_PREAGG_MIN_PCT", 90);
  e.dense_log2p       = env_i64("CUDF_AMD_GB_DENSE_LOG2P", 0);
  e.scatter_block     = env_i64("CUDF_AMD_GB_SCATTER_BLOCK", 1024);
  e.rpt               = env_i64("CUDF_AMD_GB_RPT", 8);
  e.dense_nsplit      = env_i64("CUDF_AMD_GB_DENSE_NSPLIT", -1);
  e.wc_g              = env_i64("CUDF_AMD_GB_WC_G", -1);
  e.slices            = env_i64("CUDF_AMD_GB_SLICES", -1);
  e.plan_load_pct     = env_i64("CUDF_AMD_GB_PLAN_LOAD_PCT", -1);
  e.dense             = env_i64("CUDF_AMD_GB_DENSE", 1) != 0;
  e.dense_composite   = env_i64("CUDF_AMD_GB_DENSE_COMPOSITE", 1) != 0;
  e.dense_one_table   = env_i64("CUDF_AMD_GB_DENSE_ONE_TABLE", 1) != 0;
  e.dense_ring        = env_i64("CUDF_AMD_GB_DENSE_RING", 1) != 0;
  e.dense_multi       = env_i64("CUDF_AMD_GB_DENSE_MULTI", 1) != 0;
  e.hot               = env_i64("CUDF_AMD_GB_HOT", 1) != 0;
  e.preagg            = env_i64("CUDF_AMD_GB_PREAGG", 1) != 0;
  e.optimistic        = env_i64("CUDF_AMD_GB_OPTIMISTIC", 1) != 0;
  e.optimistic2       = env_i64("CUDF_AMD_GB_OPTIMISTIC2", 1) != 0;
  e.exact             = env_i64("CUDF_AMD_GB_EXACT", 0) != 0;
  e.wc                = env_i64("CUDF_AMD_GB_WC", 1) != 0;
  e.cyclic            = env_i64("CUDF_AMD_GB_CYCLIC", 1) != 0;
  e.stamps            = env_i64("CUDF_AMD_GB_STAMPS", 0) != 0;
  e.debug             = env_i64("CUDF_AMD_DEBUG", 0) != 0;
  e.no_simple         = env_i64("CUDF_AMD_GB_NO_SIMPLE", 0) != 0;
  e.collapse_runs     = env_i64("CUDF_AMD_GB_COLLAPSE_RUNS", 1) != 0;
  e.vec16             = env_i64("CUDF_AMD_GB_VEC16", 0) != 0;
  e.trace             = env_i64("CUDF_AMD_GB_TRACE", 0) != 0;
  // (off by default: measured SLOWER than the write-combining scatter + tagged tables - profiles/r3_sparse_ring.txt)
  e.static_shapes     = env_i64("CUDF_AMD_GB_STATIC_SHAPES", 2);
  return e;
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

namespace {
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

struct record_needs {
  bool keynulls, valvalid, rowid;
};

// ---- columns: keys, then the distinct value columns (float key columns also travel as value columns)
record_needs plan_columns(host_plan& hp, table_view const& keys, null_policy policy, std::span<aggregation_request const> requests)
{
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

  return record_needs{need_keynulls, need_valvalid, need_rowid};
}

// ---- record units
void plan_units(host_plan& hp, record_needs const& needs)
{
  auto& p = hp.dev;
  bool const need_keynulls = needs.keynulls, need_valvalid = needs.valvalid, need_rowid = needs.rowid;
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

}

void plan_accumulators(host_plan& hp, std::span<aggregation_request const> requests)
{
  auto& p = hp.dev;
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
}

void plan_fast_path(host_plan& hp, planner_env const& env, bool need_rowid)
{
  auto& p = hp.dev;
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
  if (env.no_simple || need_rowid) p.simple = 0;
  p.simple_vec16 = p.simple;
  for (int w = 0; p.simple && w < p.KU + p.NPAY; ++w)
    if (reinterpret_cast<uintptr_t>(p.simple_base[w]) % 16 != 0) p.simple_vec16 = 0;
  // measured slower than 8-byte loads in both the histogram (1.85 vs 1.6 ms) and the scatter (25.5M vs 22.2M
  // cycles per workgroup): opt-in only
  if (!env.vec16) p.simple_vec16 = 0;
}
}  // namespace

host_plan build_plan(table_view const& keys, null_policy policy, std::span<aggregation_request const> requests, planner_env const& env)
{
  host_plan hp;
  auto const needs = plan_columns(hp, keys, policy, requests);
  plan_units(hp, needs);
  plan_accumulators(hp, requests);
  plan_fast_path(hp, env, needs.rowid);
  return hp;
}

}  // namespace cudf::groupby::detail
