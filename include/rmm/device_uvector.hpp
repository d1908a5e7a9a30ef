// SPDX-License-Identifier: Apache-2.0
// Typed uninitialised device vector; the join API returns unique_ptr<device_uvector<size_type>>
// (reference cpp/include/cudf/join/join.hpp:160-166).
#pragma once
#include <rmm/device_buffer.hpp>

namespace rmm {
template <typename T>
class device_uvector {
 public:
  using value_type = T;
  device_uvector(std::size_t n, hipStream_t stream,
                 device_async_resource_ref mr = mr::get_current_device_resource())
    : _buf{n * sizeof(T), stream, mr}
  {
  }
  device_uvector(device_uvector&&) noexcept            = default;
  device_uvector& operator=(device_uvector&&) noexcept = default;
  [[nodiscard]] T* data() noexcept { return static_cast<T*>(_buf.data()); }
  [[nodiscard]] T const* data() const noexcept { return static_cast<T const*>(_buf.data()); }
  [[nodiscard]] T* begin() noexcept { return data(); }
  [[nodiscard]] T* end() noexcept { return data() + size(); }
  [[nodiscard]] std::size_t size() const noexcept { return _buf.size() / sizeof(T); }
  [[nodiscard]] bool is_empty() const noexcept { return size() == 0; }
  [[nodiscard]] hipStream_t stream() const noexcept { return _buf.stream(); }
  // Hands the storage over (pylibcudf moves a join index vector into a column this way,
  // reference python/pylibcudf/pylibcudf/join.pyx:51-64).
  device_buffer release() noexcept { return std::move(_buf); }

 private:
  device_buffer _buf;
};
}  // namespace rmm
