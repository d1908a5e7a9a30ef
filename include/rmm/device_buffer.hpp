// SPDX-License-Identifier: Apache-2.0
// Owning, stream-ordered, untyped device allocation (what cudf::column is built from:
// reference cpp/include/cudf/column/column.hpp:107 takes rmm::device_buffer&&).
#pragma once
#include <rmm/resource_ref.hpp>
#include <cstddef>
#include <utility>

namespace rmm {

class device_buffer {
 public:
  device_buffer() : _mr{mr::get_current_device_resource()} {}
  device_buffer(std::size_t size, hipStream_t stream,
                device_async_resource_ref mr = mr::get_current_device_resource())
    : _size{size}, _stream{stream}, _mr{mr}
  {
    _data = _mr.allocate_async(size, stream);
  }
  // Deep copy of `size` bytes of device memory at `src`.
  device_buffer(void const* src, std::size_t size, hipStream_t stream,
                device_async_resource_ref mr = mr::get_current_device_resource());
  device_buffer(device_buffer const&)            = delete;
  device_buffer& operator=(device_buffer const&) = delete;
  device_buffer(device_buffer&& o) noexcept
    : _data{o._data}, _size{o._size}, _stream{o._stream}, _mr{o._mr}
  {
    o._data = nullptr;
    o._size = 0;
  }
  device_buffer& operator=(device_buffer&& o) noexcept
  {
    if (this != &o) {
      release_();
      _data   = o._data;
      _size   = o._size;
      _stream = o._stream;
      _mr     = o._mr;
      o._data = nullptr;
      o._size = 0;
    }
    return *this;
  }
  ~device_buffer() { release_(); }

  [[nodiscard]] void* data() noexcept { return _data; }
  [[nodiscard]] void const* data() const noexcept { return _data; }
  [[nodiscard]] std::size_t size() const noexcept { return _size; }
  [[nodiscard]] bool is_empty() const noexcept { return _size == 0; }
  [[nodiscard]] hipStream_t stream() const noexcept { return _stream; }
  void set_stream(hipStream_t s) noexcept { _stream = s; }

 private:
  void release_() noexcept
  {
    if (_data) _mr.deallocate_async(_data, _size, _stream);
    _data = nullptr;
  }
  void* _data{nullptr};
  std::size_t _size{0};
  hipStream_t _stream{nullptr};
  device_async_resource_ref _mr;
};
}  // namespace rmm
