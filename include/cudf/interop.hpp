// SPDX-License-Identifier: Apache-2.0
// Arrow C Data Interface import / export for the fixed-width types of the hash-groupby / hash-join path.
// Mirrors the reference entry points of cpp/include/cudf/interop.hpp: from_arrow (:685-689), from_arrow_column
// (:705-709), to_arrow_schema (:473-475), to_arrow_host (:618-621, :643-646), from_arrow_device (:834-838),
// to_arrow_device (:500-610).
// The ABI structs are the ones the Arrow specification publishes (ArrowSchema / ArrowArray / ArrowDeviceArray);
// an including translation unit that already has them (arrow/c/abi.h, nanoarrow) keeps its own definitions.
#pragma once
#include <cudf/column/column.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf/utilities/default_stream.hpp>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
extern "C" {
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
}
#endif
#ifndef ARROW_C_DEVICE_DATA_INTERFACE
#define ARROW_C_DEVICE_DATA_INTERFACE
extern "C" {
typedef int32_t ArrowDeviceType;
#define ARROW_DEVICE_CPU 1
#define ARROW_DEVICE_CUDA 2
#define ARROW_DEVICE_CUDA_HOST 3
#define ARROW_DEVICE_ROCM 10
#define ARROW_DEVICE_ROCM_HOST 11
struct ArrowDeviceArray {
  struct ArrowArray array;
  int64_t device_id;
  ArrowDeviceType device_type;
  void* sync_event;
  int64_t reserved[3];
};
}
#endif

namespace cudf {

struct arrow_schema_deleter {
  void operator()(ArrowSchema* s) const;
};
struct arrow_device_array_deleter {
  void operator()(ArrowDeviceArray* a) const;
};
using unique_schema_t       = std::unique_ptr<ArrowSchema, arrow_schema_deleter>;
using unique_device_array_t = std::unique_ptr<ArrowDeviceArray, arrow_device_array_deleter>;

// Column names for the exported schema (the reference's column_metadata, interop.hpp:44-60, without children).
struct column_metadata {
  std::string name;
  column_metadata() = default;
  column_metadata(std::string n) : name{std::move(n)} {}
};

// Host Arrow data (struct array = table) -> owning device table. Does not release the input. Bit-packed Arrow
// booleans become BOOL8 bytes; a non-zero array offset is honoured.
std::unique_ptr<table> from_arrow(ArrowSchema const* schema, ArrowArray const* input,
                                  stream_ref stream                 = get_default_stream(),
                                  rmm::device_async_resource_ref mr = get_current_device_resource_ref());
std::unique_ptr<column> from_arrow_column(ArrowSchema const* schema, ArrowArray const* input,
                                          stream_ref stream                 = get_default_stream(),
                                          rmm::device_async_resource_ref mr = get_current_device_resource_ref());

// Device Arrow data (device_type ARROW_DEVICE_ROCM / ROCM_HOST) -> non-owning table_view over the producer's memory.
// The ArrowDeviceArray must outlive the view. Boolean columns (bit-packed in Arrow) are not viewable: data_type_error.
struct arrow_table_view {
  table_view view;
};
std::unique_ptr<arrow_table_view> from_arrow_device(ArrowSchema const* schema, ArrowDeviceArray const* input,
                                                    stream_ref stream = get_default_stream());

unique_schema_t to_arrow_schema(table_view const& input, std::vector<column_metadata> const& metadata);
// Copies the table to host memory owned by the returned array (device_type ARROW_DEVICE_CPU, struct array).
unique_device_array_t to_arrow_host(table_view const& table, stream_ref stream = get_default_stream(),
                                    rmm::device_async_resource_ref mr = get_current_device_resource_ref());

// Device export (reference interop.hpp:500-610): an ArrowDeviceArray (device_type ARROW_DEVICE_ROCM, sync_event = a
// hipEvent_t* recorded on `stream`) whose buffers are the table's device buffers - no host bounce. The view forms only wrap the
// caller's memory, which must outlive the array; the rvalue forms move the buffers' ownership into the array. Copies happen
// only where the layouts differ: BOOL8 (Arrow booleans are bit-packed) and the validity bits of a sliced column.
unique_device_array_t to_arrow_device(table_view const& table, stream_ref stream = get_default_stream(),
                                      rmm::device_async_resource_ref mr = get_current_device_resource_ref());
unique_device_array_t to_arrow_device(column_view const& col, stream_ref stream = get_default_stream(),
                                      rmm::device_async_resource_ref mr = get_current_device_resource_ref());
unique_device_array_t to_arrow_device(table&& table, stream_ref stream = get_default_stream(),
                                      rmm::device_async_resource_ref mr = get_current_device_resource_ref());
unique_device_array_t to_arrow_device(column&& col, stream_ref stream = get_default_stream(),
                                      rmm::device_async_resource_ref mr = get_current_device_resource_ref());

}  // namespace cudf
