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
}  // namespace cudf::detail::prof
