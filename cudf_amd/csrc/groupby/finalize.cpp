// SPDX-License-Identifier: Apache-2.0
// aggregate_call::finalize: the partial records of the successful attempt -> typed key and result columns (k_finalize),
// null counts, results handed back in request order. Reference counterparts: cpp/src/groupby/hash/output_utils.cu:49-224
// (result columns and their nullability), hash_compound_agg_finalizer.cu:92-133 (MEAN / M2 / VARIANCE / STD from the
// single-pass aggregations), groupby/common/utils.hpp:39-51 (extract_results).
#include "call.hpp"

#include <algorithm>
#include <limits>

namespace cudf::groupby::detail {

using cudf::detail::CLS_BOOL;
using cudf::detail::CLS_SINT;
using cudf::detail::class_of;

std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> aggregate_call::finalize(table_view const& keys,
                                                                                            std::span<aggregation_request const> requests,
                                                                                            stream_ref stream, rmm::device_async_resource_ref mr)
{
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
    auto const& kc = keys.column(c);
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

}  // namespace cudf::groupby::detail
