// SPDX-License-Identifier: Apache-2.0
// k_aggregate instantiations for RAW partition records (aggregate_kernel.inl).
#include "aggregate_kernel.inl"

namespace cudf::groupby::detail {
void launch_aggregate_raw(agg_args const& a, agg_args* d_args, hipStream_t stream) { launch_aggregate_records<IN_RAW_RECORDS>(a, d_args, stream); }
}  // namespace cudf::groupby::detail
