// SPDX-License-Identifier: Apache-2.0
// Error convention of the drop-in boundary. Mirrors reference cpp/include/cudf/utilities/error.hpp:35
// (logic_error), :63/:86 (cuda_error / fatal_cuda_error), :97 (data_type_error), :182-199 (CUDF_EXPECTS),
// :280 (CUDF_CUDA_TRY). Device errors come from HIP here; `cuda_error` is kept as an alias so callers'
// catch clauses keep compiling.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdexcept>
#include <string>

namespace cudf {

struct logic_error : public std::logic_error {
  using std::logic_error::logic_error;
};
struct data_type_error : public std::invalid_argument {
  using std::invalid_argument::invalid_argument;
};
struct hip_error : public std::runtime_error {
  hip_error(std::string const& message, hipError_t error) : std::runtime_error(message), _code{error} {}
  [[nodiscard]] hipError_t error_code() const { return _code; }

 protected:
  hipError_t _code;
};
struct fatal_hip_error : public hip_error {
  using hip_error::hip_error;
};
using cuda_error       = hip_error;
using fatal_cuda_error = fatal_hip_error;

namespace detail {
[[noreturn]] void throw_hip_error(hipError_t error, char const* file, unsigned line);
}  // namespace detail
}  // namespace cudf

#define CUDF_STRINGIFY_DETAIL(x) #x
#define CUDF_STRINGIFY(x)        CUDF_STRINGIFY_DETAIL(x)

#define GET_CUDF_EXPECTS_MACRO(_1, _2, _3, NAME, ...) NAME
#define CUDF_EXPECTS_3(_cond, _reason, _etype)                                          \
  do {                                                                                   \
    if (!(_cond)) {                                                                      \
      throw _etype{std::string("CUDF failure at: " __FILE__ ":" CUDF_STRINGIFY(__LINE__) \
                               ": ") + (_reason)};                                       \
    }                                                                                    \
  } while (0)
#define CUDF_EXPECTS_2(_cond, _reason) CUDF_EXPECTS_3(_cond, _reason, cudf::logic_error)
#define CUDF_EXPECTS(...) \
  GET_CUDF_EXPECTS_MACRO(__VA_ARGS__, CUDF_EXPECTS_3, CUDF_EXPECTS_2)(__VA_ARGS__)

#define GET_CUDF_FAIL_MACRO(_1, _2, NAME, ...) NAME
#define CUDF_FAIL_2(_what, _etype) \
  throw _etype { std::string("CUDF failure at:" __FILE__ ":" CUDF_STRINGIFY(__LINE__) ": ") + (_what) }
#define CUDF_FAIL_1(_what) CUDF_FAIL_2(_what, cudf::logic_error)
#define CUDF_FAIL(...)     GET_CUDF_FAIL_MACRO(__VA_ARGS__, CUDF_FAIL_2, CUDF_FAIL_1)(__VA_ARGS__)

#define CUDF_HIP_TRY(call)                                                          \
  do {                                                                              \
    hipError_t const status_ = (call);                                              \
    if (hipSuccess != status_) { cudf::detail::throw_hip_error(status_, __FILE__, __LINE__); } \
  } while (0)
#define CUDF_CUDA_TRY(call) CUDF_HIP_TRY(call)
