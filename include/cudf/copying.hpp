// SPDX-License-Identifier: Apache-2.0
// cudf::gather (reference cpp/include/cudf/copying.hpp; detail cpp/include/cudf/detail/gather.cuh:119-127,525):
// out[i] = source[gather_map[i]]; with out_of_bounds_policy::NULLIFY out-of-range indices (incl. JoinNoMatch)
// yield null rows. The step every caller runs right after inner_join.
#pragma once
#include <cudf/table/table.hpp>
#include <memory>

namespace cudf {
std::unique_ptr<table> gather(table_view const& source_table, column_view const& gather_map,
                              out_of_bounds_policy bounds_policy = out_of_bounds_policy::DONT_CHECK,
                              stream_ref stream                  = get_default_stream(),
                              rmm::device_async_resource_ref mr  = get_current_device_resource_ref());
}  // namespace cudf
