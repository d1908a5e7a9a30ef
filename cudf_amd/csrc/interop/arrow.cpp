// SPDX-License-Identifier: Apache-2.0
// Arrow C Data Interface import / export (see include/cudf/interop.hpp). Reference counterparts:
// cpp/src/interop/from_arrow_host.cu, from_arrow_device.cu, to_arrow_host.cu, to_arrow_schema.cpp — restated for the
// fixed-width types this path handles; format strings are those of the Arrow specification.
#include "../common/device_table.hpp"

#include <cudf/interop.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/utilities/error.hpp>

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>

namespace cudf {
namespace {



type_id type_of_format(char const* f)
{
  CUDF_EXPECTS(f != nullptr, "ArrowSchema without a format string", std::invalid_argument);
  std::string const s{f};
  if (s == "b") return type_id::BOOL8;
  if (s == "c") return type_id::INT8;
  if (s == "C") return type_id::UINT8;
  if (s == "s") return type_id::INT16;
  if (s == "S") return type_id::UINT16;
  if (s == "i") return type_id::INT32;
  if (s == "I") return type_id::UINT32;
  if (s == "l") return type_id::INT64;
  if (s == "L") return type_id::UINT64;
  if (s == "f") return type_id::FLOAT32;
  if (s == "g") return type_id::FLOAT64;
  if (s == "tdD") return type_id::TIMESTAMP_DAYS;
  if (s.rfind("tss:", 0) == 0) return type_id::TIMESTAMP_SECONDS;
  if (s.rfind("tsm:", 0) == 0) return type_id::TIMESTAMP_MILLISECONDS;
  if (s.rfind("tsu:", 0) == 0) return type_id::TIMESTAMP_MICROSECONDS;
  if (s.rfind("tsn:", 0) == 0) return type_id::TIMESTAMP_NANOSECONDS;
  if (s == "tDs") return type_id::DURATION_SECONDS;
  if (s == "tDm") return type_id::DURATION_MILLISECONDS;
  if (s == "tDu") return type_id::DURATION_MICROSECONDS;
  if (s == "tDn") return type_id::DURATION_NANOSECONDS;
  if (s == "+s") return type_id::STRUCT;
  CUDF_FAIL("Arrow format '" + s + "' is not a fixed-width type of this path", cudf::data_type_error);
}

char const* format_of_type(type_id t)
{
  switch (t) {
    case type_id::BOOL8: return "b";
    case type_id::INT8: return "c";
    case type_id::UINT8: return "C";
    case type_id::INT16: return "s";
    case type_id::UINT16: return "S";
    case type_id::INT32: return "i";
    case type_id::UINT32: return "I";
    case type_id::INT64: return "l";
    case type_id::UINT64: return "L";
    case type_id::FLOAT32: return "f";
    case type_id::FLOAT64: return "g";
    case type_id::TIMESTAMP_DAYS: return "tdD";
    case type_id::TIMESTAMP_SECONDS: return "tss:";
    case type_id::TIMESTAMP_MILLISECONDS: return "tsm:";
    case type_id::TIMESTAMP_MICROSECONDS: return "tsu:";
    case type_id::TIMESTAMP_NANOSECONDS: return "tsn:";
    case type_id::DURATION_SECONDS: return "tDs";
    case type_id::DURATION_MILLISECONDS: return "tDm";
    case type_id::DURATION_MICROSECONDS: return "tDu";
    case type_id::DURATION_NANOSECONDS: return "tDn";
    default: CUDF_FAIL("column type has no Arrow export on this path", cudf::data_type_error);
  }
}

// bits [bit_offset, bit_offset + n) of an LSB-first bitmap, repacked to start at bit 0 (host)
std::vector<uint8_t> repack_bits(uint8_t const* src, int64_t bit_offset, int64_t n)
{
  std::vector<uint8_t> out(static_cast<std::size_t>((n + 7) / 8 + 8), 0);
  if ((bit_offset & 7) == 0) {
    std::memcpy(out.data(), src + bit_offset / 8, static_cast<std::size_t>((n + 7) / 8));
  } else {
    for (int64_t i = 0; i < n; ++i) {
      int64_t const b = bit_offset + i;
      if ((src[b >> 3] >> (b & 7)) & 1u) out[static_cast<std::size_t>(i >> 3)] |= static_cast<uint8_t>(1u << (i & 7));
    }
  }
  if (n & 7) out[static_cast<std::size_t>(n >> 3)] &= static_cast<uint8_t>((1u << (n & 7)) - 1);  // clear the tail bits
  return out;
}

__global__ void __launch_bounds__(256) k_bits_to_bytes(uint8_t const* bits, int64_t n, uint8_t* bytes)
{
  int64_t const i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) bytes[i] = (bits[i >> 3] >> (i & 7)) & 1u;
}

std::unique_ptr<column> column_from_host(ArrowSchema const* schema, ArrowArray const* a, stream_ref stream,
                                         rmm::device_async_resource_ref mr)
{
  type_id const tid = type_of_format(schema->format);
  CUDF_EXPECTS(tid != type_id::STRUCT, "nested Arrow arrays are not supported on this path", cudf::data_type_error);
  CUDF_EXPECTS(a->dictionary == nullptr && schema->dictionary == nullptr, "dictionary-encoded Arrow arrays are not supported",
               cudf::data_type_error);
  CUDF_EXPECTS(a->length <= std::numeric_limits<size_type>::max(), "Arrow array exceeds the column size limit", std::overflow_error);
  CUDF_EXPECTS(a->n_buffers == 2, "fixed-width Arrow arrays have two buffers", std::invalid_argument);
  hipStream_t const s = stream.value();
  auto const n        = static_cast<size_type>(a->length);
  auto const* valid   = static_cast<uint8_t const*>(a->buffers[0]);
  auto const* data    = static_cast<uint8_t const*>(a->buffers[1]);
  // ---- validity
  rmm::device_buffer mask{};
  size_type nulls = 0;
  if (valid != nullptr && a->null_count != 0 && n > 0) {
    auto const packed = repack_bits(valid, a->offset, n);
    mask              = rmm::device_buffer{bitmask_allocation_size_bytes(n), s, mr};
    CUDF_HIP_TRY(hipMemsetAsync(mask.data(), 0, mask.size(), s));
    CUDF_HIP_TRY(hipMemcpyAsync(mask.data(), packed.data(), static_cast<std::size_t>((n + 7) / 8), hipMemcpyHostToDevice, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));  // `packed` goes out of scope
    nulls = cudf::null_count(static_cast<bitmask_type const*>(mask.data()), 0, n, stream);
    if (nulls == 0) mask = rmm::device_buffer{};
  }
  // ---- data
  std::size_t const width = size_of_id(tid);
  rmm::device_buffer dev{static_cast<std::size_t>(n) * width, s, mr};
  if (n > 0) {
    CUDF_EXPECTS(data != nullptr, "Arrow array without a data buffer", std::invalid_argument);
    if (tid == type_id::BOOL8) {
      auto const packed = repack_bits(data, a->offset, n);
      rmm::device_buffer bits{static_cast<std::size_t>((n + 7) / 8), s, cudf::get_current_device_resource_ref()};
      CUDF_HIP_TRY(hipMemcpyAsync(bits.data(), packed.data(), bits.size(), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(k_bits_to_bytes, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s,
                         static_cast<uint8_t const*>(bits.data()), static_cast<int64_t>(n), static_cast<uint8_t*>(dev.data()));
      CUDF_HIP_TRY(hipGetLastError());
      CUDF_HIP_TRY(hipStreamSynchronize(s));
    } else {
      CUDF_HIP_TRY(hipMemcpyAsync(dev.data(), data + static_cast<std::size_t>(a->offset) * width, dev.size(), hipMemcpyHostToDevice, s));
      CUDF_HIP_TRY(hipStreamSynchronize(s));  // the caller may release the Arrow data right after the call
    }
  }
  return std::make_unique<column>(data_type{tid}, n, std::move(dev), std::move(mask), nulls);
}

// ---- export bookkeeping: everything an exported ArrowArray / ArrowSchema points at lives in one of these
struct array_private {
  std::vector<void*> host_buffers;          // malloc'ed, freed on release
  std::vector<const void*> buffer_ptrs;     // what ArrowArray::buffers points at
  std::vector<ArrowArray> child_storage;
  std::vector<ArrowArray*> child_ptrs;
  // device export (to_arrow_device): buffers copied for the export, a column whose buffers the array owns, the sync event
  std::vector<rmm::device_buffer> device_buffers;
  std::unique_ptr<column> owned_column;
  std::vector<const void*> const_buffers;
  hipEvent_t event{};
  bool has_event{false};
};
void release_array(ArrowArray* a)
{
  if (a == nullptr || a->release == nullptr) return;
  for (int64_t i = 0; i < a->n_children; ++i)
    if (a->children[i]->release != nullptr) a->children[i]->release(a->children[i]);
  auto* p = static_cast<array_private*>(a->private_data);
  for (void* b : p->host_buffers) std::free(b);
  if (p->has_event) (void)hipEventDestroy(p->event);
  delete p;
  a->release = nullptr;
}
struct schema_private {
  std::string format, name;
  std::vector<ArrowSchema> child_storage;
  std::vector<ArrowSchema*> child_ptrs;
};
void release_schema(ArrowSchema* s)
{
  if (s == nullptr || s->release == nullptr) return;
  for (int64_t i = 0; i < s->n_children; ++i)
    if (s->children[i]->release != nullptr) s->children[i]->release(s->children[i]);
  delete static_cast<schema_private*>(s->private_data);
  s->release = nullptr;
}

__global__ void __launch_bounds__(256) k_bytes_to_bits(uint8_t const* bytes, int64_t n, uint32_t* words)
{
  int64_t const i        = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  bool const set         = i < n && bytes[i] != 0;
  unsigned long long const b = __ballot(set);
  int const lane         = threadIdx.x & 63;
  if (lane == 0 && i < n) words[i >> 5] = static_cast<uint32_t>(b);
  if (lane == 32 && i < n) words[i >> 5] = static_cast<uint32_t>(b >> 32);
}

void export_column(column_view const& c, ArrowArray* out, hipStream_t s)
{
  auto* p          = new array_private{};
  size_type const n = c.size();
  std::size_t const width = size_of_id(c.type().id());
  (void)format_of_type(c.type().id());  // throws for unsupported types before anything is allocated
  // validity bitmap: bits [offset, offset + n) repacked to bit 0 on the host
  void* valid = nullptr;
  if (c.nullable() && c.null_count() > 0 && n > 0) {
    std::size_t const first_word = static_cast<std::size_t>(c.offset()) / 32;
    std::size_t const nwords     = (static_cast<std::size_t>(c.offset()) + n + 31) / 32 - first_word;
    std::vector<uint32_t> words(nwords);
    CUDF_HIP_TRY(hipMemcpyAsync(words.data(), c.null_mask() + first_word, nwords * 4, hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    auto const packed = repack_bits(reinterpret_cast<uint8_t const*>(words.data()), c.offset() % 32, n);
    valid             = std::malloc(packed.size());
    std::memcpy(valid, packed.data(), packed.size());
    p->host_buffers.push_back(valid);
  }
  void* data = std::malloc(std::max<std::size_t>(c.type().id() == type_id::BOOL8 ? (static_cast<std::size_t>(n) + 7) / 8 + 8
                                                                                  : static_cast<std::size_t>(n) * width, 8));
  p->host_buffers.push_back(data);
  if (n > 0) {
    uint8_t const* src = static_cast<uint8_t const*>(c.head()) + static_cast<std::size_t>(c.offset()) * width;
    if (c.type().id() == type_id::BOOL8) {  // BOOL8 bytes -> Arrow bit-packed booleans
      std::size_t const nwords = (static_cast<std::size_t>(n) + 31) / 32 + 2;
      rmm::device_buffer bits{nwords * 4, s, cudf::get_current_device_resource_ref()};
      CUDF_HIP_TRY(hipMemsetAsync(bits.data(), 0, bits.size(), s));
      hipLaunchKernelGGL(k_bytes_to_bits, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, src,
                         static_cast<int64_t>(n), static_cast<uint32_t*>(bits.data()));
      CUDF_HIP_TRY(hipGetLastError());
      CUDF_HIP_TRY(hipMemcpyAsync(data, bits.data(), (static_cast<std::size_t>(n) + 7) / 8, hipMemcpyDeviceToHost, s));
    } else {
      CUDF_HIP_TRY(hipMemcpyAsync(data, src, static_cast<std::size_t>(n) * width, hipMemcpyDeviceToHost, s));
    }
    CUDF_HIP_TRY(hipStreamSynchronize(s));
  }
  p->buffer_ptrs   = {valid, data};
  out->length      = n;
  out->null_count  = valid != nullptr ? c.null_count() : 0;
  out->offset      = 0;
  out->n_buffers   = 2;
  out->n_children  = 0;
  out->buffers     = p->buffer_ptrs.data();
  out->children    = nullptr;
  out->dictionary  = nullptr;
  out->release     = release_array;
  out->private_data = p;
}


// ---- device export: the Arrow buffers ARE the column's device buffers wherever the layouts agree
__global__ void __launch_bounds__(256) k_shift_bits(uint32_t const* words, int64_t bit_offset, int64_t n, uint32_t* out)
{
  int64_t const w = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;  // output word
  if (w * 32 >= n) return;
  int64_t const b   = bit_offset + w * 32, idx = b >> 5;
  uint64_t const lo = words[idx], hi = (idx + 1) * 32 < bit_offset + n ? words[idx + 1] : 0;
  uint32_t v        = static_cast<uint32_t>(((hi << 32) | lo) >> (b & 31));
  if ((w + 1) * 32 > n) v &= (1u << (n & 31)) - 1u;  // clear the bits past the last row
  out[w] = v;
}

// One column as a device ArrowArray: buffers {validity bitmap, data} in device memory. `owned` (may be null) is a column
// whose buffers the array takes over; otherwise the array only views `c` (plus whatever had to be copied).
void export_column_device(column_view const& c, std::unique_ptr<column> owned, ArrowArray* out, hipStream_t s,
                          rmm::device_async_resource_ref mr)
{
  auto* p = new array_private{};
  try {
    size_type const n       = c.size();
    std::size_t const width = size_of_id(c.type().id());
    (void)format_of_type(c.type().id());
    void const* valid = nullptr;
    if (c.nullable() && c.null_count() > 0 && n > 0) {
      if (c.offset() == 0) {
        valid = c.null_mask();
      } else {  // a sliced column: Arrow wants the bits of [offset, offset + n) - repacked to bit 0 on the device, offset = 0
        std::size_t const nwords = (static_cast<std::size_t>(n) + 31) / 32;
        p->device_buffers.emplace_back(nwords * 4 + 64, s, mr);
        hipLaunchKernelGGL(k_shift_bits, dim3(static_cast<unsigned>((nwords + 255) / 256)), dim3(256), 0, s, c.null_mask(),
                           static_cast<int64_t>(c.offset()), static_cast<int64_t>(n), static_cast<uint32_t*>(p->device_buffers.back().data()));
        CUDF_HIP_TRY(hipGetLastError());
        valid = p->device_buffers.back().data();
      }
    }
    void const* data = nullptr;
    uint8_t const* src = static_cast<uint8_t const*>(c.head()) + static_cast<std::size_t>(c.offset()) * width;
    if (c.type().id() == type_id::BOOL8) {  // BOOL8 bytes -> Arrow bit-packed booleans (a copy, as in the reference)
      std::size_t const nwords = (static_cast<std::size_t>(n) + 31) / 32 + 2;
      p->device_buffers.emplace_back(nwords * 4, s, mr);
      CUDF_HIP_TRY(hipMemsetAsync(p->device_buffers.back().data(), 0, nwords * 4, s));
      if (n > 0) {
        hipLaunchKernelGGL(k_bytes_to_bits, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, src, static_cast<int64_t>(n),
                           static_cast<uint32_t*>(p->device_buffers.back().data()));
        CUDF_HIP_TRY(hipGetLastError());
      }
      data = p->device_buffers.back().data();
    } else {
      data = n > 0 ? src : nullptr;
    }
    p->owned_column   = std::move(owned);
    p->const_buffers  = {valid, data};
    out->length       = n;
    out->null_count   = valid != nullptr ? c.null_count() : 0;
    out->offset       = 0;
    out->n_buffers    = 2;
    out->n_children   = 0;
    out->buffers      = p->const_buffers.data();
    out->children     = nullptr;
    out->dictionary   = nullptr;
    out->release      = release_array;
    out->private_data = p;
  } catch (...) {
    delete p;
    throw;
  }
}

unique_device_array_t finish_device_array(array_private* p, int64_t length, hipStream_t s)
{
  unique_device_array_t out{new ArrowDeviceArray{}};
  // the consumer orders its work behind the exporting stream through the event (Arrow C Device Data Interface: sync_event
  // is a hipEvent_t* for ARROW_DEVICE_ROCM)
  CUDF_HIP_TRY(hipEventCreateWithFlags(&p->event, hipEventDisableTiming));
  p->has_event = true;
  CUDF_HIP_TRY(hipEventRecord(p->event, s));
  int dev = 0;
  CUDF_HIP_TRY(hipGetDevice(&dev));
  out->array.length       = length;
  out->array.null_count   = 0;
  out->array.offset       = 0;
  out->array.n_buffers    = p->child_ptrs.empty() && !p->const_buffers.empty() ? 2 : 1;
  out->array.n_children   = static_cast<int64_t>(p->child_ptrs.size());
  out->array.buffers      = p->const_buffers.data();
  out->array.children     = p->child_ptrs.empty() ? nullptr : p->child_ptrs.data();
  out->array.dictionary   = nullptr;
  out->array.release      = release_array;
  out->array.private_data = p;
  out->device_id          = dev;
  out->device_type        = ARROW_DEVICE_ROCM;
  out->sync_event         = &p->event;
  return out;
}

unique_device_array_t table_to_device_array(table_view const& table, std::vector<std::unique_ptr<column>> owned, stream_ref stream,
                                            rmm::device_async_resource_ref mr)
{
  auto* p = new array_private{};
  p->child_storage.resize(static_cast<std::size_t>(table.num_columns()));
  try {
    for (size_type i = 0; i < table.num_columns(); ++i) {
      auto& slot   = p->child_storage[static_cast<std::size_t>(i)];
      slot.release = nullptr;
      export_column_device(table.column(i), owned.empty() ? nullptr : std::move(owned[static_cast<std::size_t>(i)]), &slot, stream.value(), mr);
      p->child_ptrs.push_back(&slot);
    }
    p->const_buffers = {nullptr};
    return finish_device_array(p, table.num_rows(), stream.value());
  } catch (...) {
    for (auto& c : p->child_storage)
      if (c.release != nullptr) c.release(&c);
    delete p;
    throw;
  }
}

void export_schema(type_id t, std::string const& name, bool nullable, ArrowSchema* out)
{
  auto* p         = new schema_private{};
  p->format       = format_of_type(t);
  p->name         = name;
  out->format     = p->format.c_str();
  out->name       = p->name.c_str();
  out->metadata   = nullptr;
  out->flags      = nullable ? ARROW_FLAG_NULLABLE : 0;
  out->n_children = 0;
  out->children   = nullptr;
  out->dictionary = nullptr;
  out->release    = release_schema;
  out->private_data = p;
}

}  // namespace

void arrow_schema_deleter::operator()(ArrowSchema* s) const
{
  if (s == nullptr) return;
  if (s->release != nullptr) s->release(s);
  delete s;
}
void arrow_device_array_deleter::operator()(ArrowDeviceArray* a) const
{
  if (a == nullptr) return;
  if (a->array.release != nullptr) a->array.release(&a->array);
  delete a;
}

std::unique_ptr<column> from_arrow_column(ArrowSchema const* schema, ArrowArray const* input, stream_ref stream,
                                          rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(schema != nullptr && input != nullptr, "input ArrowSchema and ArrowArray must not be NULL", std::invalid_argument);
  return column_from_host(schema, input, stream, mr);
}

std::unique_ptr<table> from_arrow(ArrowSchema const* schema, ArrowArray const* input, stream_ref stream,
                                  rmm::device_async_resource_ref mr)
{
  CUDF_EXPECTS(schema != nullptr && input != nullptr, "input ArrowSchema and ArrowArray must not be NULL", std::invalid_argument);
  CUDF_EXPECTS(type_of_format(schema->format) == type_id::STRUCT,
               "from_arrow needs a struct array (one child per column); use from_arrow_column", cudf::data_type_error);
  CUDF_EXPECTS(schema->n_children == input->n_children, "ArrowSchema and ArrowArray disagree on the number of columns",
               std::invalid_argument);
  CUDF_EXPECTS(input->offset == 0 || input->n_children == 0, "a struct array with an offset is not supported", std::invalid_argument);
  std::vector<std::unique_ptr<column>> cols;
  for (int64_t i = 0; i < input->n_children; ++i) cols.push_back(column_from_host(schema->children[i], input->children[i], stream, mr));
  return std::make_unique<table>(std::move(cols));
}

std::unique_ptr<arrow_table_view> from_arrow_device(ArrowSchema const* schema, ArrowDeviceArray const* input, stream_ref stream)
{
  CUDF_EXPECTS(schema != nullptr && input != nullptr, "input ArrowSchema and ArrowDeviceArray must not be NULL", std::invalid_argument);
  CUDF_EXPECTS(input->device_type == ARROW_DEVICE_ROCM || input->device_type == ARROW_DEVICE_ROCM_HOST,
               "ArrowDeviceArray memory must be accessible to the ROCm device", std::invalid_argument);
  if (input->sync_event != nullptr)  // the producer's hipEvent_t*: order the consumer's stream after it
    CUDF_HIP_TRY(hipStreamWaitEvent(stream.value(), *static_cast<hipEvent_t*>(input->sync_event), 0));
  CUDF_EXPECTS(type_of_format(schema->format) == type_id::STRUCT, "from_arrow_device needs a struct array", cudf::data_type_error);
  CUDF_EXPECTS(schema->n_children == input->array.n_children, "ArrowSchema and ArrowArray disagree on the number of columns",
               std::invalid_argument);
  std::vector<column_view> cols;
  for (int64_t i = 0; i < input->array.n_children; ++i) {
    ArrowArray const* a = input->array.children[i];
    type_id const tid   = type_of_format(schema->children[i]->format);
    CUDF_EXPECTS(tid != type_id::STRUCT && tid != type_id::BOOL8,
                 "nested and bit-packed boolean Arrow arrays cannot be viewed without a copy", cudf::data_type_error);
    CUDF_EXPECTS(a->length + a->offset <= std::numeric_limits<size_type>::max(), "Arrow array exceeds the column size limit",
                 std::overflow_error);
    auto const* mask = static_cast<bitmask_type const*>(a->buffers[0]);
    size_type nulls  = 0;
    if (mask != nullptr && a->null_count != 0)
      nulls = a->null_count > 0 ? static_cast<size_type>(a->null_count)
                                : cudf::null_count(mask, static_cast<size_type>(a->offset),
                                                   static_cast<size_type>(a->offset + a->length), stream);
    cols.emplace_back(data_type{tid}, static_cast<size_type>(a->length), a->buffers[1], nulls > 0 ? mask : nullptr, nulls,
                      static_cast<size_type>(a->offset));
  }
  auto out  = std::make_unique<arrow_table_view>();
  out->view = table_view{cols};
  return out;
}

unique_schema_t to_arrow_schema(table_view const& input, std::vector<column_metadata> const& metadata)
{
  CUDF_EXPECTS(metadata.size() == static_cast<std::size_t>(input.num_columns()), "one column_metadata per column is required",
               std::invalid_argument);
  auto* p = new schema_private{};
  p->format = "+s";
  p->child_storage.resize(static_cast<std::size_t>(input.num_columns()));
  for (size_type i = 0; i < input.num_columns(); ++i) {
    export_schema(input.column(i).type().id(), metadata[static_cast<std::size_t>(i)].name, true, &p->child_storage[static_cast<std::size_t>(i)]);
    p->child_ptrs.push_back(&p->child_storage[static_cast<std::size_t>(i)]);
  }
  unique_schema_t out{new ArrowSchema{}};
  out->format       = p->format.c_str();
  out->name         = "";
  out->metadata     = nullptr;
  out->flags        = 0;
  out->n_children   = input.num_columns();
  out->children     = p->child_ptrs.data();
  out->dictionary   = nullptr;
  out->release      = release_schema;
  out->private_data = p;
  return out;
}

unique_device_array_t to_arrow_host(table_view const& table, stream_ref stream, rmm::device_async_resource_ref)
{
  auto* p = new array_private{};
  p->child_storage.resize(static_cast<std::size_t>(table.num_columns()));
  unique_device_array_t out{new ArrowDeviceArray{}};
  out->array.release = nullptr;
  try {
    for (size_type i = 0; i < table.num_columns(); ++i) {
      p->child_storage[static_cast<std::size_t>(i)].release = nullptr;
      export_column(table.column(i), &p->child_storage[static_cast<std::size_t>(i)], stream.value());
      p->child_ptrs.push_back(&p->child_storage[static_cast<std::size_t>(i)]);
    }
  } catch (...) {
    for (auto& c : p->child_storage)
      if (c.release != nullptr) c.release(&c);
    delete p;
    throw;
  }
  p->buffer_ptrs          = {nullptr};
  out->array.length       = table.num_rows();
  out->array.null_count   = 0;
  out->array.offset       = 0;
  out->array.n_buffers    = 1;
  out->array.n_children   = table.num_columns();
  out->array.buffers      = p->buffer_ptrs.data();
  out->array.children     = p->child_ptrs.data();
  out->array.dictionary   = nullptr;
  out->array.release      = release_array;
  out->array.private_data = p;
  out->device_id          = -1;
  out->device_type        = ARROW_DEVICE_CPU;
  out->sync_event         = nullptr;
  return out;
}

// ---- to_arrow_device (reference interop.hpp:500-610): the table's data stays on the device. The view forms wrap the
// caller's buffers (which must outlive the array); the rvalue forms hand the buffers' ownership to the array.
unique_device_array_t to_arrow_device(table_view const& table, stream_ref stream, rmm::device_async_resource_ref mr)
{
  return table_to_device_array(table, {}, stream, mr);
}
unique_device_array_t to_arrow_device(column_view const& col, stream_ref stream, rmm::device_async_resource_ref mr)
{
  ArrowArray tmp{};
  export_column_device(col, nullptr, &tmp, stream.value(), mr);
  auto* p = static_cast<array_private*>(tmp.private_data);
  try {
    auto out              = finish_device_array(p, col.size(), stream.value());
    out->array.null_count = tmp.null_count;
    out->array.n_buffers  = 2;
    return out;
  } catch (...) {
    delete p;
    throw;
  }
}
unique_device_array_t to_arrow_device(table&& tbl, stream_ref stream, rmm::device_async_resource_ref mr)
{
  auto const view = tbl.view();
  return table_to_device_array(view, tbl.release(), stream, mr);
}
unique_device_array_t to_arrow_device(column&& col, stream_ref stream, rmm::device_async_resource_ref mr)
{
  auto owned      = std::make_unique<column>(std::move(col));
  auto const view = owned->view();
  ArrowArray tmp{};
  export_column_device(view, std::move(owned), &tmp, stream.value(), mr);
  auto* p = static_cast<array_private*>(tmp.private_data);
  try {
    auto out              = finish_device_array(p, view.size(), stream.value());
    out->array.null_count = tmp.null_count;
    out->array.n_buffers  = 2;
    return out;
  } catch (...) {
    delete p;
    throw;
  }
}

}  // namespace cudf
