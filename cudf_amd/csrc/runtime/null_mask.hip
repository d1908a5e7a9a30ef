// SPDX-License-Identifier: Apache-2.0
// Null-mask kernels used on the path: allocation, offset-aware AND of column masks, unset-bit counting.
// Reference: cpp/include/cudf/null_mask.hpp:56 (64-byte padded allocation), detail/null_mask.cuh:67
// (offset_bitmask_binop), src/bitmask/null_mask.cu:409 (count_set_bits_kernel) — one wave-wide word per lane,
// popcounts reduced with 64-lane shuffles and one atomic per wave.
#include "../common/device_table.hpp"

#include <cudf/null_mask.hpp>
#include <cudf/utilities/bit.hpp>
#include <cudf/utilities/error.hpp>

namespace cudf {
namespace {

struct mask_list {
  bitmask_type const* mask[detail::MAX_COLS];
  int32_t offset[detail::MAX_COLS];
  int32_t n;
};

// 32 mask bits starting at absolute bit `b` of `m`, never reading words at or beyond word index `last`.
__device__ __forceinline__ uint32_t load_bits32(bitmask_type const* m, int64_t b, int64_t last_word)
{
  int64_t const w = b >> 5;
  int const sh    = static_cast<int>(b & 31);
  uint32_t lo     = m[w];
  if (sh == 0) return lo;
  uint32_t const hi = (w + 1 <= last_word) ? m[w + 1] : 0u;
  return (lo >> sh) | (hi << (32 - sh));
}

__global__ void __launch_bounds__(256) k_bitmask_and(mask_list ml, bitmask_type* out, int32_t nbits, int32_t* unset)
{
  int64_t const w       = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  int64_t const nwords  = (static_cast<int64_t>(nbits) + 31) / 32;
  int nulls             = 0;
  if (w < nwords) {
    uint32_t acc = 0xffffffffu;
    for (int c = 0; c < ml.n; ++c) {
      int64_t const first = static_cast<int64_t>(ml.offset[c]) + w * 32;
      int64_t const last  = (static_cast<int64_t>(ml.offset[c]) + nbits - 1) >> 5;
      acc &= load_bits32(ml.mask[c], first, last);
    }
    int const live = static_cast<int>(min<int64_t>(32, nbits - w * 32));
    if (live < 32) acc &= (1u << live) - 1u;
    out[w] = acc;
    nulls  = live - __popc(acc);
  }
  for (int o = 32; o > 0; o >>= 1) nulls += __shfl_down(nulls, o);
  if ((threadIdx.x & 63) == 0 && nulls) atomicAdd(unset, nulls);
}

__global__ void __launch_bounds__(256) k_count_unset(bitmask_type const* m, int64_t start, int64_t stop, int32_t* unset)
{
  int64_t const nbits  = stop - start;
  int64_t const nwords = (nbits + 31) / 32;
  int64_t w            = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  int nulls            = 0;
  for (; w < nwords; w += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    uint32_t bits  = load_bits32(m, start + w * 32, (stop - 1) >> 5);
    int const live = static_cast<int>(min<int64_t>(32, nbits - w * 32));
    if (live < 32) bits &= (1u << live) - 1u;
    nulls += live - __popc(bits);
  }
  for (int o = 32; o > 0; o >>= 1) nulls += __shfl_down(nulls, o);
  if ((threadIdx.x & 63) == 0 && nulls) atomicAdd(unset, nulls);
}
}  // namespace

std::size_t bitmask_allocation_size_bytes(size_type number_of_bits, std::size_t padding_boundary)
{
  CUDF_EXPECTS(padding_boundary > 0, "Invalid padding boundary");
  auto const bytes = (static_cast<std::size_t>(number_of_bits) + 7) / 8;
  return padding_boundary * ((bytes + padding_boundary - 1) / padding_boundary);
}

rmm::device_buffer create_null_mask(size_type size, mask_state state, stream_ref stream, rmm::device_async_resource_ref mr)
{
  if (state == mask_state::UNALLOCATED) return rmm::device_buffer{};
  rmm::device_buffer mask{bitmask_allocation_size_bytes(size), stream.value(), mr};
  if (state != mask_state::UNINITIALIZED && mask.size() > 0) {
    CUDF_HIP_TRY(hipMemsetAsync(mask.data(), state == mask_state::ALL_VALID ? 0xff : 0x00, mask.size(), stream.value()));
  }
  return mask;
}

std::pair<rmm::device_buffer, size_type> bitmask_and(table_view const& view, stream_ref stream,
                                                     rmm::device_async_resource_ref mr)
{
  mask_list ml{};
  for (auto const& c : view) {
    if (c.nullable()) {
      CUDF_EXPECTS(ml.n < detail::MAX_COLS, "Too many nullable columns.");
      ml.mask[ml.n]   = c.null_mask();
      ml.offset[ml.n] = c.offset();
      ++ml.n;
    }
  }
  size_type const n = view.num_rows();
  if (ml.n == 0 || n == 0) return {rmm::device_buffer{}, 0};
  auto out = create_null_mask(n, mask_state::UNINITIALIZED, stream, mr);
  rmm::device_buffer counter{sizeof(int32_t), stream.value(), cudf::get_current_device_resource_ref()};
  CUDF_HIP_TRY(hipMemsetAsync(counter.data(), 0, sizeof(int32_t), stream.value()));
  int64_t const nwords = (static_cast<int64_t>(n) + 31) / 32;
  hipLaunchKernelGGL(k_bitmask_and, dim3(static_cast<unsigned>((nwords + 255) / 256)), dim3(256), 0, stream.value(), ml,
                     static_cast<bitmask_type*>(out.data()), n, static_cast<int32_t*>(counter.data()));
  CUDF_HIP_TRY(hipGetLastError());
  int32_t h = 0;
  CUDF_HIP_TRY(hipMemcpyAsync(&h, counter.data(), sizeof(int32_t), hipMemcpyDeviceToHost, stream.value()));
  CUDF_HIP_TRY(hipStreamSynchronize(stream.value()));
  return {std::move(out), h};
}

size_type null_count(bitmask_type const* bitmask, size_type start, size_type stop, stream_ref stream)
{
  CUDF_EXPECTS(start >= 0 && start <= stop, "Invalid bit range.");
  if (bitmask == nullptr || start == stop) return 0;
  rmm::device_buffer counter{sizeof(int32_t), stream.value(), cudf::get_current_device_resource_ref()};
  CUDF_HIP_TRY(hipMemsetAsync(counter.data(), 0, sizeof(int32_t), stream.value()));
  int64_t const nwords = (static_cast<int64_t>(stop - start) + 31) / 32;
  unsigned const grid  = static_cast<unsigned>(std::min<int64_t>((nwords + 255) / 256, 2048));
  hipLaunchKernelGGL(k_count_unset, dim3(grid), dim3(256), 0, stream.value(), bitmask, static_cast<int64_t>(start),
                     static_cast<int64_t>(stop), static_cast<int32_t*>(counter.data()));
  CUDF_HIP_TRY(hipGetLastError());
  int32_t h = 0;
  CUDF_HIP_TRY(hipMemcpyAsync(&h, counter.data(), sizeof(int32_t), hipMemcpyDeviceToHost, stream.value()));
  CUDF_HIP_TRY(hipStreamSynchronize(stream.value()));
  return h;
}
}  // namespace cudf
