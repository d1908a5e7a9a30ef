// SPDX-License-Identifier: Apache-2.0
// k_aggregate instantiations for rows read straight from the input columns (LDS_SINGLE_PASS; aggregate_kernel.inl).
#include "aggregate_kernel.inl"

namespace cudf::groupby::detail {
void launch_aggregate_columns(agg_args const& a, agg_args* d_args, hipStream_t stream)
{
  int const KU = a.plan.KU;
  bool const simple = a.plan.simple;
  if (simple && KU == 1 && a.plan.NPAY == 1) {  // one plain key column, one plain value column
    uint64_t const sig = plan_sig(a.plan);
    if (sig == SIG_SUMF_CNT) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, true, true, SIG_SUMF_CNT>(a, d_args, stream);
    if (sig == SIG_SUMI_CNT) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, true, true, SIG_SUMI_CNT>(a, d_args, stream);
    if (sig == SIG_SUMF) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, true, true, SIG_SUMF>(a, d_args, stream);
    if (sig == SIG_SUMI) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, true, true, SIG_SUMI>(a, d_args, stream);
    return launch_aggregate_t<IN_COLUMNS, 1, 1, true, true>(a, d_args, stream);
  }
  if (!simple && a.plan.NPAY >= 1 && a.plan.NPAY <= 2 && KU >= 1 && KU <= 2 && a.plan.ncols <= MAX_LOCAL_COLS) {  // generic columns, known record shape
    uint64_t const sig = plan_sig(a.plan);
    if (KU == 1 && a.plan.NPAY == 1) {
      if (sig == SIG_SUMF_CNT) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, false, true, SIG_SUMF_CNT>(a, d_args, stream);
      if (sig == SIG_SUMI_CNT) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, false, true, SIG_SUMI_CNT>(a, d_args, stream);
      if (sig == SIG_SUMF_CNT_NULLS) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, false, true, SIG_SUMF_CNT_NULLS>(a, d_args, stream);
      if (sig == SIG_SUMI_CNT_NULLS) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, false, true, SIG_SUMI_CNT_NULLS>(a, d_args, stream);
      return launch_aggregate_t<IN_COLUMNS, 1, 1, false, true>(a, d_args, stream);
    }
    if (KU == 1 && a.plan.NPAY == 2) {
      if (sig == SIG_SUMF_CNT_NULLS) return launch_aggregate_n<IN_COLUMNS, 1, 2, 2, false, true, SIG_SUMF_CNT_NULLS>(a, d_args, stream);
      if (sig == SIG_SUMI_CNT_NULLS) return launch_aggregate_n<IN_COLUMNS, 1, 2, 2, false, true, SIG_SUMI_CNT_NULLS>(a, d_args, stream);
      return launch_aggregate_t<IN_COLUMNS, 1, 2, false, true>(a, d_args, stream);
    }
    if (KU == 2 && a.plan.NPAY == 1) return launch_aggregate_t<IN_COLUMNS, 2, 1, false, true>(a, d_args, stream);
    return launch_aggregate_t<IN_COLUMNS, 2, 2, false, true>(a, d_args, stream);
  }
  if (KU <= 1) simple ? launch_aggregate_t<IN_COLUMNS, 1, 0, true, false>(a, d_args, stream) : launch_aggregate_t<IN_COLUMNS, 1, 0, false, false>(a, d_args, stream);
  else if (KU <= 2) simple ? launch_aggregate_t<IN_COLUMNS, 2, 0, true, false>(a, d_args, stream) : launch_aggregate_t<IN_COLUMNS, 2, 0, false, false>(a, d_args, stream);
  else simple ? launch_aggregate_t<IN_COLUMNS, 4, 0, true, false>(a, d_args, stream) : launch_aggregate_t<IN_COLUMNS, 4, 0, false, false>(a, d_args, stream);
}

}  // namespace cudf::groupby::detail
