// SPDX-License-Identifier: Apache-2.0
// Non-owning Arrow-layout column views: {type, size, data*, null_mask*, null_count, offset, children}.
// Same members, accessors and slicing rule as reference cpp/include/cudf/column/column_view.hpp:236-244,469:
// element i lives at head<T>()[offset+i]; its validity is bit (offset+i) of null_mask (LSB-first, 1=valid);
// null_mask==nullptr means all valid and null_count==0 (:149,:276).
#pragma once
#include <cudf/types.hpp>
#include <cudf/utilities/error.hpp>
#include <vector>

namespace cudf {
namespace detail {
class column_view_base {
 public:
  template <typename T = void> T const* head() const noexcept { return static_cast<T const*>(_data); }
  template <typename T> T const* data() const noexcept { return head<T>() + _offset; }
  template <typename T> T const* begin() const noexcept { return data<T>(); }
  template <typename T> T const* end() const noexcept { return begin<T>() + size(); }
  [[nodiscard]] size_type size() const noexcept { return _size; }
  [[nodiscard]] bool is_empty() const noexcept { return size() == 0; }
  [[nodiscard]] data_type type() const noexcept { return _type; }
  [[nodiscard]] bool nullable() const noexcept { return nullptr != _null_mask; }
  [[nodiscard]] size_type null_count() const { return _null_count; }
  [[nodiscard]] bool has_nulls() const { return null_count() > 0; }
  [[nodiscard]] bitmask_type const* null_mask() const noexcept { return _null_mask; }
  [[nodiscard]] size_type offset() const noexcept { return _offset; }

 protected:
  data_type _type{type_id::EMPTY};
  size_type _size{};
  void const* _data{};
  bitmask_type const* _null_mask{};
  mutable size_type _null_count{};
  size_type _offset{};

  column_view_base()                                   = default;
  ~column_view_base()                                  = default;
  column_view_base(column_view_base const&)            = default;
  column_view_base(column_view_base&&)                 = default;
  column_view_base& operator=(column_view_base const&) = default;
  column_view_base& operator=(column_view_base&&)      = default;
  column_view_base(data_type type, size_type size, void const* data, bitmask_type const* null_mask,
                   size_type null_count, size_type offset = 0);
};
}  // namespace detail

class column_view : public detail::column_view_base {
 public:
  column_view() = default;
  column_view(data_type type, size_type size, void const* data, bitmask_type const* null_mask,
              size_type null_count, size_type offset = 0, std::vector<column_view> const& children = {});
  [[nodiscard]] column_view child(size_type i) const noexcept { return _children[i]; }
  [[nodiscard]] size_type num_children() const noexcept { return static_cast<size_type>(_children.size()); }

 private:
  std::vector<column_view> _children{};
};

class mutable_column_view : public detail::column_view_base {
 public:
  mutable_column_view() = default;
  mutable_column_view(data_type type, size_type size, void* data, bitmask_type* null_mask,
                      size_type null_count, size_type offset = 0);
  template <typename T = void> T* head() const noexcept
  {
    return const_cast<T*>(detail::column_view_base::head<T>());
  }
  template <typename T> T* data() const noexcept { return const_cast<T*>(detail::column_view_base::data<T>()); }
  [[nodiscard]] bitmask_type* null_mask() const noexcept
  {
    return const_cast<bitmask_type*>(detail::column_view_base::null_mask());
  }
  void set_null_count(size_type new_null_count);
  operator column_view() const;
};

namespace detail {
// Shallow identity used by the groupby result cache (reference column_view.hpp: shallow_hash /
// is_shallow_equivalent; cache key in cpp/include/cudf/detail/aggregation/result_cache.hpp:16-29).
std::size_t shallow_hash(column_view const& input);
bool is_shallow_equivalent(column_view const& lhs, column_view const& rhs);
}  // namespace detail
}  // namespace cudf
