// SPDX-License-Identifier: Apache-2.0
// Stream vocabulary: every API takes (stream, mr) last, as in the reference (`cuda::stream_ref`,
// cpp/include/cudf/utilities/default_stream.hpp; default chosen in cpp/src/utilities/default_stream.cpp:39).
// On MI355X the stream is a hipStream_t.
#pragma once
#include <hip/hip_runtime_api.h>

namespace cudf {

class stream_ref {
 public:
  constexpr stream_ref() = default;
  constexpr stream_ref(hipStream_t s) : _stream{s} {}
  [[nodiscard]] constexpr hipStream_t value() const noexcept { return _stream; }
  constexpr operator hipStream_t() const noexcept { return _stream; }
  void synchronize() const;

 private:
  hipStream_t _stream{nullptr};
};

stream_ref const get_default_stream();
bool is_ptds_enabled();

}  // namespace cudf
