// SPDX-License-Identifier: Apache-2.0
// Owning column: data buffer + optional null mask buffer (reference cpp/include/cudf/column/column.hpp:107).
#pragma once
#include <cudf/column/column_view.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <rmm/device_buffer.hpp>
#include <rmm/device_uvector.hpp>
#include <memory>
#include <vector>

namespace cudf {
class column {
 public:
  column() = default;
  column(column&&) noexcept = default;
  column(column const& other, stream_ref stream = get_default_stream(),
         rmm::device_async_resource_ref mr = get_current_device_resource_ref());
  column(data_type dtype, size_type size, rmm::device_buffer&& data, rmm::device_buffer&& null_mask,
         size_type null_count, std::vector<std::unique_ptr<column>>&& children = {})
    : _type{dtype}, _size{size}, _data{std::move(data)}, _null_mask{std::move(null_mask)},
      _null_count{null_count}, _children{std::move(children)}
  {
  }
  // Takes over a device_uvector's storage (join index vector -> INT32 column).
  template <typename T>
  column(rmm::device_uvector<T>&& other, rmm::device_buffer&& null_mask, size_type null_count)
    : _type{data_type{type_to_id<T>()}}, _size{static_cast<size_type>(other.size())},
      _data{other.release()}, _null_mask{std::move(null_mask)}, _null_count{null_count}
  {
  }
  // Deep copy of a view.
  explicit column(column_view view, stream_ref stream = get_default_stream(),
                  rmm::device_async_resource_ref mr = get_current_device_resource_ref());

  [[nodiscard]] data_type type() const noexcept { return _type; }
  [[nodiscard]] size_type size() const noexcept { return _size; }
  [[nodiscard]] size_type null_count() const { return _null_count; }
  [[nodiscard]] bool nullable() const noexcept { return _null_mask.size() > 0; }
  [[nodiscard]] bool has_nulls() const noexcept { return _null_count > 0; }
  void set_null_mask(rmm::device_buffer&& new_null_mask, size_type new_null_count);
  void set_null_count(size_type new_null_count);
  [[nodiscard]] size_type num_children() const noexcept { return static_cast<size_type>(_children.size()); }
  [[nodiscard]] column_view view() const;
  operator column_view() const { return this->view(); }
  [[nodiscard]] mutable_column_view mutable_view();

  struct contents {
    std::unique_ptr<rmm::device_buffer> data;
    std::unique_ptr<rmm::device_buffer> null_mask;
    std::vector<std::unique_ptr<column>> children;
  };
  contents release() noexcept;

 private:
  data_type _type{type_id::EMPTY};
  size_type _size{};
  rmm::device_buffer _data{};
  rmm::device_buffer _null_mask{};
  mutable size_type _null_count{};
  std::vector<std::unique_ptr<column>> _children{};
};

// Typed empty column / uninitialised fixed-width column factories
// (reference cpp/include/cudf/column/column_factories.hpp).
std::unique_ptr<column> make_empty_column(data_type type);
std::unique_ptr<column> make_fixed_width_column(data_type type, size_type size, mask_state state,
                                                stream_ref stream = get_default_stream(),
                                                rmm::device_async_resource_ref mr = get_current_device_resource_ref());
}  // namespace cudf
