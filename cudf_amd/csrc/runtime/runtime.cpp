// SPDX-License-Identifier: Apache-2.0
// Host runtime behind the boundary types: memory resources, streams, errors, views and owning
// column/table. Reference counterparts: cpp/src/column/column_view.cpp, column.cu, table/table.cpp,
// table_view.cpp, utilities/default_stream.cpp:39 — restated from their documented behaviour for HIP.
#include <cudf/column/column.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/table/table.hpp>
#include <cudf/utilities/bit.hpp>
#include <cudf/utilities/default_stream.hpp>
#include <cudf/utilities/error.hpp>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <new>
#include <string>

namespace cudf {

std::size_t size_of(data_type t)
{
  auto const s = size_of_id(t.id());
  CUDF_EXPECTS(s != 0, "Invalid, non fixed-width element type.");
  return s;
}

namespace detail {
[[noreturn]] void throw_hip_error(hipError_t error, char const* file, unsigned line)
{
  // Clear the sticky-less last error so later calls are not poisoned by a recoverable failure.
  (void)hipGetLastError();
  std::string msg = std::string{"HIP error encountered at: "} + file + ":" + std::to_string(line) + ": " +
                    std::to_string(static_cast<int>(error)) + " " + hipGetErrorName(error) + " " +
                    hipGetErrorString(error);
  // A fault that kills the context stays: a second query still fails (reference error.hpp:63-86).
  if (error == hipErrorIllegalAddress || error == hipErrorLaunchFailure || error == hipErrorAssert) {
    throw fatal_hip_error{"Fatal " + msg, error};
  }
  if (error == hipErrorOutOfMemory) { throw std::bad_alloc{}; }
  throw hip_error{msg, error};
}
}  // namespace detail

// ---------------------------------------------------------------- streams
void stream_ref::synchronize() const { CUDF_HIP_TRY(hipStreamSynchronize(_stream)); }

bool is_ptds_enabled()
{
  // Reference reads CUDF_PER_THREAD_STREAM (cpp/src/utilities/default_stream.cpp:39).
  static bool const v = [] {
    char const* e = std::getenv("CUDF_PER_THREAD_STREAM");
    return e != nullptr && std::string{e} == "1";
  }();
  return v;
}

stream_ref const get_default_stream()
{
  return is_ptds_enabled() ? stream_ref{hipStreamPerThread} : stream_ref{nullptr};
}

// ---------------------------------------------------------------- views
namespace detail {
column_view_base::column_view_base(data_type type, size_type size, void const* data,
                                   bitmask_type const* null_mask, size_type null_count, size_type offset)
  : _type{type}, _size{size}, _data{data}, _null_mask{null_mask}, _null_count{null_count}, _offset{offset}
{
  CUDF_EXPECTS(size >= 0, "Column size cannot be negative.");
  if (type.id() == type_id::EMPTY) {
    _null_count = size;
    CUDF_EXPECTS(nullptr == data, "EMPTY column should have no data.");
    CUDF_EXPECTS(nullptr == null_mask, "EMPTY column should have no null mask.");
  } else if (is_fixed_width(type) && size > 0) {
    CUDF_EXPECTS(nullptr != data, "Null data pointer.");
  }
  CUDF_EXPECTS(offset >= 0, "Invalid offset.");
  if ((null_count > 0) && (type.id() != type_id::EMPTY)) {
    CUDF_EXPECTS(nullptr != null_mask, "Invalid null mask for non-zero null count.");
  }
}

std::size_t shallow_hash(column_view const& c)
{
  auto mix = [](std::size_t h, std::size_t v) { return h ^ (v + 0x9e3779b97f4a7c15ULL + (h << 6) + (h >> 2)); };
  std::size_t h = std::hash<int32_t>{}(static_cast<int32_t>(c.type().id()));
  h             = mix(h, std::hash<size_type>{}(c.size()));
  h             = mix(h, std::hash<void const*>{}(c.head()));
  h             = mix(h, std::hash<void const*>{}(c.null_mask()));
  h             = mix(h, std::hash<size_type>{}(c.offset()));
  return h;
}

bool is_shallow_equivalent(column_view const& a, column_view const& b)
{
  return a.type() == b.type() && a.size() == b.size() && a.head() == b.head() && a.null_mask() == b.null_mask() &&
         a.offset() == b.offset();
}
}  // namespace detail

column_view::column_view(data_type type, size_type size, void const* data, bitmask_type const* null_mask,
                         size_type null_count, size_type offset, std::vector<column_view> const& children)
  : detail::column_view_base{type, size, data, null_mask, null_count, offset}, _children{children}
{
  if (type.id() == type_id::EMPTY) { CUDF_EXPECTS(num_children() == 0, "EMPTY column cannot have children."); }
}

mutable_column_view::mutable_column_view(data_type type, size_type size, void* data, bitmask_type* null_mask,
                                         size_type null_count, size_type offset)
  : detail::column_view_base{type, size, data, null_mask, null_count, offset}
{
}
void mutable_column_view::set_null_count(size_type new_null_count)
{
  if (new_null_count > 0) { CUDF_EXPECTS(nullable(), "Invalid null count."); }
  _null_count = new_null_count;
}
mutable_column_view::operator column_view() const
{
  return column_view{_type, _size, _data, _null_mask, _null_count, _offset};
}

table_view::table_view(std::vector<column_view> const& cols) : _columns{cols}
{
  if (!_columns.empty()) {
    _num_rows = _columns.front().size();
    for (auto const& c : _columns) { CUDF_EXPECTS(c.size() == _num_rows, "Column size mismatch."); }
  }
}
table_view::table_view(std::vector<table_view> const& views)
{
  for (auto const& v : views) {
    for (auto const& c : v) { _columns.push_back(c); }
  }
  if (!_columns.empty()) {
    _num_rows = _columns.front().size();
    for (auto const& c : _columns) { CUDF_EXPECTS(c.size() == _num_rows, "Column size mismatch."); }
  }
}
table_view table_view::select(std::vector<size_type> const& column_indices) const
{
  std::vector<column_view> cols;
  cols.reserve(column_indices.size());
  for (auto i : column_indices) { cols.push_back(column(i)); }
  return table_view{cols};
}
bool has_nulls(table_view const& view)
{
  for (auto const& c : view) {
    if (c.has_nulls()) return true;
  }
  return false;
}
bool nullable(table_view const& view)
{
  for (auto const& c : view) {
    if (c.nullable()) return true;
  }
  return false;
}

// ---------------------------------------------------------------- owning column / table
column::column(column_view view, stream_ref stream, rmm::device_async_resource_ref mr)
  : _type{view.type()}, _size{view.size()}, _null_count{view.null_count()}
{
  // a STRUCT column (SUM_OVERFLOW's {sum, overflow}): the children are copied, the parent holds only the mask
  CUDF_EXPECTS(view.num_children() == 0 || _type.id() == type_id::STRUCT, "Only fixed-width and STRUCT columns are supported on this path.");
  for (size_type i = 0; i < view.num_children(); ++i) _children.push_back(std::make_unique<column>(view.child(i), stream, mr));
  if (_size > 0 && is_fixed_width(_type)) {
    auto const w = size_of(_type);
    _data = rmm::device_buffer{static_cast<char const*>(view.head()) + std::size_t(view.offset()) * w,
                               std::size_t(_size) * w, stream, mr};
  }
  if (view.nullable() && _size > 0) {
    // Re-base the mask at bit 0 (slice-aware copy).
    auto [mask, nulls] = bitmask_and(table_view{{view}}, stream, mr);
    _null_mask         = std::move(mask);
    _null_count        = nulls;
  }
}
column::column(column const& other, stream_ref stream, rmm::device_async_resource_ref mr)
  : column{other.view(), stream, mr}
{
}
void column::set_null_mask(rmm::device_buffer&& new_null_mask, size_type new_null_count)
{
  if (new_null_count > 0) {
    CUDF_EXPECTS(new_null_mask.size() >= bitmask_allocation_size_bytes(_size, 4),
                 "Column with null values must be nullable and the null mask buffer size should match the size "
                 "of the column.");
  }
  _null_mask  = std::move(new_null_mask);
  _null_count = new_null_count;
}
void column::set_null_count(size_type new_null_count)
{
  if (new_null_count > 0) { CUDF_EXPECTS(nullable(), "Invalid null count."); }
  _null_count = new_null_count;
}
column_view column::view() const
{
  // (children: reference column.cu - a STRUCT column's view carries the views of its children)
  std::vector<column_view> kids;
  kids.reserve(_children.size());
  for (auto const& c : _children) kids.push_back(c->view());
  return column_view{_type, _size, _data.data(), static_cast<bitmask_type const*>(_null_mask.data()), _null_count, 0, kids};
}
mutable_column_view column::mutable_view()
{
  return mutable_column_view{_type, _size, _data.data(), static_cast<bitmask_type*>(_null_mask.data()), _null_count, 0};
}
column::contents column::release() noexcept
{
  _size       = 0;
  _null_count = 0;
  _type       = data_type{type_id::EMPTY};
  return contents{std::make_unique<rmm::device_buffer>(std::move(_data)),
                  std::make_unique<rmm::device_buffer>(std::move(_null_mask)), std::move(_children)};
}

std::unique_ptr<column> make_empty_column(data_type type)
{
  return std::make_unique<column>(type, 0, rmm::device_buffer{}, rmm::device_buffer{}, 0);
}
std::unique_ptr<column> make_fixed_width_column(data_type type, size_type size, mask_state state, stream_ref stream,
                                                rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(is_fixed_width(type), "Invalid, non-fixed-width type.");
  CUDF_EXPECTS(size >= 0, "Column size cannot be negative.");
  auto const nulls = state == mask_state::ALL_NULL ? size : 0;
  return std::make_unique<column>(type, size, rmm::device_buffer{std::size_t(size) * size_of(type), stream, mr},
                                  create_null_mask(size, state, stream, mr), nulls);
}

table::table(std::vector<std::unique_ptr<column>>&& columns) : _columns{std::move(columns)}
{
  if (!_columns.empty()) {
    for (auto const& c : _columns) {
      CUDF_EXPECTS(c != nullptr, "Unexpected null column");
      CUDF_EXPECTS(c->size() == _columns.front()->size(), "Column size mismatch.");
    }
    _num_rows = _columns.front()->size();
  }
}
table::table(table_view view, stream_ref stream, rmm::device_async_resource_ref mr) : _num_rows{view.num_rows()}
{
  for (auto const& c : view) { _columns.emplace_back(std::make_unique<column>(c, stream, mr)); }
}
table_view table::view() const
{
  std::vector<column_view> v;
  v.reserve(_columns.size());
  for (auto const& c : _columns) { v.push_back(c->view()); }
  return table_view{v};
}
std::vector<std::unique_ptr<column>> table::release() noexcept
{
  _num_rows = 0;
  return std::move(_columns);
}
std::unique_ptr<table> empty_like(table_view const& input)
{
  std::vector<std::unique_ptr<column>> cols;
  for (auto const& c : input) { cols.push_back(make_empty_column(c.type())); }
  return std::make_unique<table>(std::move(cols));
}
}  // namespace cudf

// ---------------------------------------------------------------- memory resources
namespace rmm {
namespace mr {

hip_async_memory_resource::hip_async_memory_resource()
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
  hipMemPool_t pool{};
  if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
    uint64_t threshold = UINT64_MAX;  // keep freed blocks in the pool: 288 GB HBM, no trimming on sync
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &threshold);
  }
  (void)hipGetLastError();
}
void* hip_async_memory_resource::do_allocate(std::size_t bytes, hipStream_t stream)
{
  void* p      = nullptr;
  auto const e = hipMallocAsync(&p, bytes, stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (e == hipErrorOutOfMemory) throw std::bad_alloc{};
    cudf::detail::throw_hip_error(e, __FILE__, __LINE__);
  }
  return p;
}
void hip_async_memory_resource::do_deallocate(void* p, std::size_t, hipStream_t stream) noexcept
{
  (void)hipFreeAsync(p, stream);
}

// ---- caching pool
struct pool_memory_resource::impl {
  struct block {
    void* ptr;
    std::size_t size;
    hipEvent_t freed;      // recorded on `stream` only when ANOTHER stream takes the block (see do_allocate)
    hipStream_t stream;    // the stream the block was last used on
    uint64_t tick;         // when it was returned (eviction order)
  };
  uint64_t clock{0};
  int device{-1};  // the HIP device of the first allocation: cached blocks are only valid there
  std::mutex mu;
  std::map<std::size_t, std::vector<block>> free_lists;  // rounded size -> blocks
  std::unordered_map<void*, block> live;
  std::size_t cached{0};
};

static std::size_t round_pool_size(std::size_t bytes)
{
  constexpr std::size_t small = 256, big = std::size_t{2} << 20;
  if (bytes < (std::size_t{1} << 20)) return (bytes + small - 1) / small * small;
  return (bytes + big - 1) / big * big;
}

pool_memory_resource::pool_memory_resource() : _impl{new impl} {}
pool_memory_resource::~pool_memory_resource()
{
  // Process teardown: the HIP runtime may already be gone; leak the device memory deliberately.
  delete _impl;
}
std::size_t pool_memory_resource::cached_bytes() const { return _impl->cached; }
void pool_memory_resource::trim()
{
  std::lock_guard<std::mutex> g{_impl->mu};
  for (auto& [sz, v] : _impl->free_lists) {
    for (auto& b : v) {
      if (hipStreamSynchronize(b.stream) != hipSuccess) {  // (work queued on the block's stream may still use it)
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
      }
      (void)hipEventDestroy(b.freed);
      (void)hipFree(b.ptr);
    }
  }
  _impl->free_lists.clear();
  _impl->cached = 0;
}
void* pool_memory_resource::do_allocate(std::size_t bytes, hipStream_t stream)
{
  std::size_t const sz = round_pool_size(bytes);
  {
    // One pool serves one device (one process per GPU): a block cached on another device must never be handed out.
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
      std::lock_guard<std::mutex> g{_impl->mu};
      if (_impl->device < 0) _impl->device = dev;
      if (_impl->device != dev)
        throw cudf::logic_error("rmm::mr::pool_memory_resource: the pool belongs to HIP device " + std::to_string(_impl->device) +
                                ", the calling thread's current device is " + std::to_string(dev) + " (one process per GPU)");
    }
  }
  {
    std::lock_guard<std::mutex> g{_impl->mu};
    auto it = _impl->free_lists.find(sz);
    if (it != _impl->free_lists.end() && !it->second.empty()) {
      auto b = it->second.back();
      it->second.pop_back();
      _impl->cached -= sz;
      if (b.stream != stream) {
        // The block was returned on another stream: everything queued there up to NOW covers its last use (an event recorded
        // at the time of the return would do too, but a record per returned block cost 2-5 us on every deallocation of every
        // call, and blocks almost always come back to the stream that returned them). If that stream is gone, wait for the device.
        if (hipEventRecord(b.freed, b.stream) == hipSuccess) {
          auto const e = hipStreamWaitEvent(stream, b.freed, 0);
          if (e != hipSuccess) cudf::detail::throw_hip_error(e, __FILE__, __LINE__);
        } else {
          (void)hipGetLastError();
          auto const e = hipDeviceSynchronize();
          if (e != hipSuccess) cudf::detail::throw_hip_error(e, __FILE__, __LINE__);
        }
      }
      b.stream           = stream;
      _impl->live[b.ptr] = b;
      return b.ptr;
    }
  }
  void* p = nullptr;
  auto e  = hipMalloc(&p, sz);
  if (e == hipErrorOutOfMemory) {  // give cached blocks back and retry once
    (void)hipGetLastError();
    trim();
    e = hipMalloc(&p, sz);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (e == hipErrorOutOfMemory) throw std::bad_alloc{};
    cudf::detail::throw_hip_error(e, __FILE__, __LINE__);
  }
  impl::block b{p, sz, nullptr, stream, 0};
  e = hipEventCreateWithFlags(&b.freed, hipEventDisableTiming);
  if (e != hipSuccess) cudf::detail::throw_hip_error(e, __FILE__, __LINE__);
  std::lock_guard<std::mutex> g{_impl->mu};
  _impl->live[p] = b;
  return p;
}
void pool_memory_resource::do_deallocate(void* p, std::size_t, hipStream_t stream) noexcept
{
  std::lock_guard<std::mutex> g{_impl->mu};
  auto it = _impl->live.find(p);
  if (it == _impl->live.end()) return;  // not ours (should not happen)
  auto b   = it->second;
  b.stream = stream;
  _impl->live.erase(it);
  b.tick = ++_impl->clock;
  _impl->free_lists[b.size].push_back(b);
  _impl->cached += b.size;
  // Bound what the cache may hold (every distinct rounded size keeps its own list): beyond the limit
  // (CUDF_AMD_POOL_MAX_CACHED_GB, default 128 of the 288 GB) the least recently returned blocks go back to the
  // driver first, so a caller that changes its shapes does not lose the blocks of its current working set.
  static std::size_t const limit = [] {
    char const* e = std::getenv("CUDF_AMD_POOL_MAX_CACHED_GB");
    std::size_t const gb = (e != nullptr && *e != 0) ? std::strtoull(e, nullptr, 10) : 128;
    return gb << 30;
  }();
  while (_impl->cached > limit) {
    std::vector<impl::block>* oldest_list = nullptr;
    std::size_t oldest_idx = 0;
    uint64_t oldest_tick   = ~uint64_t{0};
    for (auto& [sz, v] : _impl->free_lists)
      for (std::size_t i = 0; i < v.size(); ++i)
        if (v[i].tick < oldest_tick) {
          oldest_tick = v[i].tick;
          oldest_list = &v;
          oldest_idx  = i;
        }
    if (oldest_list == nullptr) break;
    auto const blk = (*oldest_list)[oldest_idx];
    oldest_list->erase(oldest_list->begin() + static_cast<std::ptrdiff_t>(oldest_idx));
    if (hipStreamSynchronize(blk.stream) != hipSuccess) {  // (work queued on the block's stream may still use it)
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
    }
    (void)hipEventDestroy(blk.freed);
    (void)hipFree(blk.ptr);
    _impl->cached -= blk.size;
  }
}

namespace {
device_memory_resource*& current_slot()
{
  static device_memory_resource* cur = nullptr;
  return cur;
}
std::mutex& slot_mutex()
{
  static std::mutex m;
  return m;
}
}  // namespace

device_memory_resource* get_current_device_resource()
{
  std::lock_guard<std::mutex> g{slot_mutex()};
  auto*& cur = current_slot();
  if (cur == nullptr) {
    static pool_memory_resource default_mr{};
    cur = &default_mr;
  }
  return cur;
}
device_memory_resource* set_current_device_resource(device_memory_resource* mr)
{
  auto* old = get_current_device_resource();
  std::lock_guard<std::mutex> g{slot_mutex()};
  current_slot() = mr;
  return old;
}
}  // namespace mr

device_buffer::device_buffer(void const* src, std::size_t size, hipStream_t stream, device_async_resource_ref mr)
  : _size{size}, _stream{stream}, _mr{mr}
{
  _data = _mr.allocate_async(size, stream);
  if (size > 0) { CUDF_HIP_TRY(hipMemcpyAsync(_data, src, size, hipMemcpyDefault, stream)); }
}
}  // namespace rmm
