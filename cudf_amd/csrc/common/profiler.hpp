// SPDX-License-Identifier: Apache-2.0
// Optional per-kernel timing with HIP events recorded on the stream each kernel is launched on (what
// bench.py's `roofline` object reports). The reference instruments with NVTX ranges
// (cpp/include/cudf/detail/nvtx/ranges.hpp:50); here the measurement is taken in-process so it can be
// reported next to the throughput number. Disabled by default: no events, no overhead.
#pragma once
#include <hip/hip_runtime_api.h>
#include <string>
#include <vector>

namespace cudf::detail::prof {

struct kernel_stat {
  std::string name;
  int64_t launches;
  double total_ms;
};

void enable(bool on);
bool enabled();
void reset();
// Synchronises the recorded events and returns per-kernel totals accumulated since reset().
std::vector<kernel_stat> collect();

class scope {
 public:
  scope(char const* name, hipStream_t stream);
  ~scope();
  scope(scope const&)            = delete;
  scope& operator=(scope const&) = delete;

 private:
  int _slot{-1};
  hipStream_t _stream{};
};

// roctx range over a public entry point - the counterpart of the reference's CUDF_FUNC_RANGE() NVTX ranges
// (cpp/include/cudf/detail/nvtx/ranges.hpp:50; first line of groupby.cu:224, join.cu:116). The roctx library
// (librocprofiler-sdk-roctx / libroctx64) is resolved at run time on first use; without it a range costs one relaxed load.
// `rocprofv3 --marker-trace` shows the ranges.
class func_range {
 public:
  explicit func_range(char const* name);
  ~func_range();
  func_range(func_range const&)            = delete;
  func_range& operator=(func_range const&) = delete;

 private:
  bool _pushed{false};
};
}  // namespace cudf::detail::prof

#define CUDF_FUNC_RANGE() ::cudf::detail::prof::func_range const cudf_func_range_{__func__}
