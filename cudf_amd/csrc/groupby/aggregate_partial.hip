// SPDX-License-Identifier: Apache-2.0
// k_aggregate instantiations for PARTIAL records (merge rounds; aggregate_kernel.inl).
#include "aggregate_kernel.inl"

namespace cudf::groupby::detail {
void launch_aggregate_partial(agg_args const& a, agg_args* d_args, hipStream_t stream) { launch_aggregate_records<IN_PARTIAL_RECORDS>(a, d_args, stream); }
}  // namespace cudf::groupby::detail
