// SPDX-License-Identifier: Apache-2.0
// Non-owning set of equally sized column views (reference cpp/include/cudf/table/table_view.hpp).
#pragma once
#include <cudf/column/column_view.hpp>
#include <vector>

namespace cudf {
class table_view {
 public:
  using iterator       = std::vector<column_view>::iterator;
  using const_iterator = std::vector<column_view>::const_iterator;
  table_view() = default;
  explicit table_view(std::vector<column_view> const& cols);
  explicit table_view(std::vector<table_view> const& views);
  [[nodiscard]] const_iterator begin() const noexcept { return _columns.begin(); }
  [[nodiscard]] const_iterator end() const noexcept { return _columns.end(); }
  [[nodiscard]] column_view const& column(size_type i) const { return _columns.at(i); }
  [[nodiscard]] size_type num_columns() const noexcept { return static_cast<size_type>(_columns.size()); }
  [[nodiscard]] size_type num_rows() const noexcept { return _num_rows; }
  [[nodiscard]] bool is_empty() const noexcept { return num_columns() == 0; }
  [[nodiscard]] table_view select(std::vector<size_type> const& column_indices) const;

 private:
  std::vector<column_view> _columns{};
  size_type _num_rows{};
};
// True if any column has nulls (reference cudf::has_nulls, table_view.hpp).
bool has_nulls(table_view const& view);
bool nullable(table_view const& view);
}  // namespace cudf
