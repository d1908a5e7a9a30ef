// SPDX-License-Identifier: Apache-2.0
// Owning table (reference cpp/include/cudf/table/table.hpp:59).
#pragma once
#include <cudf/column/column.hpp>
#include <cudf/table/table_view.hpp>
#include <memory>
#include <vector>

namespace cudf {
class table {
 public:
  table()                 = default;
  table(table&&) noexcept = default;
  explicit table(std::vector<std::unique_ptr<column>>&& columns);
  explicit table(table_view view, stream_ref stream = get_default_stream(),
                 rmm::device_async_resource_ref mr = get_current_device_resource_ref());
  [[nodiscard]] size_type num_columns() const noexcept { return static_cast<size_type>(_columns.size()); }
  [[nodiscard]] size_type num_rows() const noexcept { return _num_rows; }
  [[nodiscard]] table_view view() const;
  operator table_view() const { return this->view(); }
  std::vector<std::unique_ptr<column>> release() noexcept;
  [[nodiscard]] column& get_column(size_type i) { return *(_columns.at(i)); }
  [[nodiscard]] column const& get_column(size_type i) const { return *(_columns.at(i)); }

 private:
  std::vector<std::unique_ptr<column>> _columns{};
  size_type _num_rows{};
};
// Typed zero-row copy of a view's schema (reference cudf::empty_like, used by groupby.cu:233).
std::unique_ptr<table> empty_like(table_view const& input);
}  // namespace cudf
