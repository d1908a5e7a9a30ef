// SPDX-License-Identifier: Apache-2.0
// Null-mask utilities used on the path (reference cpp/include/cudf/null_mask.hpp:56 (64-byte padding),
// bitmask_and -> cpp/include/cudf/detail/null_mask.cuh:67, null counting -> cpp/src/bitmask/null_mask.cu:409).
#pragma once
#include <cudf/table/table_view.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <rmm/device_buffer.hpp>
#include <utility>

namespace cudf {
std::size_t bitmask_allocation_size_bytes(size_type number_of_bits, std::size_t padding_boundary = 64);
rmm::device_buffer create_null_mask(size_type size, mask_state state, stream_ref stream = get_default_stream(),
                                    rmm::device_async_resource_ref mr = get_current_device_resource_ref());
// AND of the null masks of all columns of `view` (offset-aware); returns {mask (empty if no column is
// nullable), number of unset bits}.
std::pair<rmm::device_buffer, size_type> bitmask_and(table_view const& view,
                                                     stream_ref stream = get_default_stream(),
                                                     rmm::device_async_resource_ref mr = get_current_device_resource_ref());
// Number of unset bits in [start, stop) of `bitmask`.
size_type null_count(bitmask_type const* bitmask, size_type start, size_type stop,
                     stream_ref stream = get_default_stream());
}  // namespace cudf
