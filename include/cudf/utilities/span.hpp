// SPDX-License-Identifier: Apache-2.0
// cudf::host_span / cudf::device_span: non-owning views over contiguous host- / device-accessible elements
// (reference cpp/include/cudf/utilities/span.hpp:65-300, :302-440). The reference backs them with cuda::std::span; here
// they are a pointer and a size with the members the path's signatures use.
#pragma once
#include <cudf/types.hpp>
#include <rmm/device_uvector.hpp>

#include <cstddef>
#include <type_traits>
#include <vector>

namespace cudf {
namespace detail {
template <typename T>
class span_base {
 public:
  using element_type = T;
  using value_type   = std::remove_cv_t<T>;
  using size_type    = std::size_t;
  using pointer      = T*;
  using iterator     = T*;
  using reference    = T&;

  constexpr span_base() noexcept = default;
  constexpr span_base(T* data, std::size_t size) noexcept : _data{data}, _size{size} {}

  [[nodiscard]] constexpr T* data() const noexcept { return _data; }
  [[nodiscard]] constexpr std::size_t size() const noexcept { return _size; }
  [[nodiscard]] constexpr std::size_t size_bytes() const noexcept { return _size * sizeof(T); }
  [[nodiscard]] constexpr bool empty() const noexcept { return _size == 0; }
  [[nodiscard]] constexpr T* begin() const noexcept { return _data; }
  [[nodiscard]] constexpr T* end() const noexcept { return _data + _size; }
  [[nodiscard]] constexpr T& operator[](std::size_t i) const { return _data[i]; }
  [[nodiscard]] constexpr T& front() const { return _data[0]; }
  [[nodiscard]] constexpr T& back() const { return _data[_size - 1]; }

 private:
  T* _data{nullptr};
  std::size_t _size{0};
};
}  // namespace detail

template <typename T>
struct host_span : public detail::span_base<T> {
  using base = detail::span_base<T>;
  using base::base;
  constexpr host_span() noexcept = default;
  // from a std::vector (reference span.hpp:117-131)
  template <typename U, typename A, std::enable_if_t<std::is_convertible_v<U (*)[], T (*)[]>, int> = 0>
  constexpr host_span(std::vector<U, A>& in) : base{in.data(), in.size()}
  {
  }
  template <typename U, typename A, std::enable_if_t<std::is_convertible_v<U const (*)[], T (*)[]>, int> = 0>
  constexpr host_span(std::vector<U, A> const& in) : base{in.data(), in.size()}
  {
  }
  template <typename U, std::enable_if_t<!std::is_same_v<U, T> && std::is_convertible_v<U (*)[], T (*)[]>, int> = 0>
  constexpr host_span(host_span<U> const& other) noexcept : base{other.data(), other.size()}
  {
  }
  [[nodiscard]] constexpr host_span subspan(std::size_t offset, std::size_t count) const noexcept
  {
    return host_span{this->data() + offset, count};
  }
};

template <typename T>
struct device_span : public detail::span_base<T> {
  using base = detail::span_base<T>;
  using base::base;
  constexpr device_span() noexcept = default;
  // from an rmm::device_uvector (reference span.hpp:355-369)
  template <typename U, std::enable_if_t<std::is_convertible_v<U (*)[], T (*)[]>, int> = 0>
  device_span(rmm::device_uvector<U>& in) : base{in.data(), in.size()}
  {
  }
  template <typename U, std::enable_if_t<std::is_convertible_v<U const (*)[], T (*)[]>, int> = 0>
  device_span(rmm::device_uvector<U> const& in) : base{in.data(), in.size()}
  {
  }
  template <typename U, std::enable_if_t<!std::is_same_v<U, T> && std::is_convertible_v<U (*)[], T (*)[]>, int> = 0>
  constexpr device_span(device_span<U> const& other) noexcept : base{other.data(), other.size()}
  {
  }
  [[nodiscard]] constexpr device_span subspan(std::size_t offset, std::size_t count) const noexcept
  {
    return device_span{this->data() + offset, count};
  }
};
}  // namespace cudf
