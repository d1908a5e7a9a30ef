// SPDX-License-Identifier: Apache-2.0
// Stream-ordered device memory resources for MI355X. The reference allocates every output through the
// caller's `rmm::device_async_resource_ref` (e.g. cpp/include/cudf/groupby.hpp:181-184) and temporaries
// through cudf::get_current_device_resource_ref(); this is the HIP-native equivalent with the same shape:
// a non-owning ref onto a polymorphic resource with allocate/deallocate(bytes, stream).
#pragma once
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <cstddef>

namespace rmm {
namespace mr {

class device_memory_resource {
 public:
  virtual ~device_memory_resource() = default;
  void* allocate(std::size_t bytes, hipStream_t stream) { return bytes ? do_allocate(bytes, stream) : nullptr; }
  void deallocate(void* p, std::size_t bytes, hipStream_t stream) noexcept
  {
    if (p) do_deallocate(p, bytes, stream);
  }

 private:
  virtual void* do_allocate(std::size_t bytes, hipStream_t stream)                    = 0;
  virtual void do_deallocate(void* p, std::size_t bytes, hipStream_t stream) noexcept = 0;
};

// hipMallocAsync / hipFreeAsync on the device's default pool (release threshold raised so that 288 GB of
// HBM3E is recycled instead of trimmed between calls). Throws std::bad_alloc on OOM.
class hip_async_memory_resource final : public device_memory_resource {
 public:
  hip_async_memory_resource();

 private:
  void* do_allocate(std::size_t bytes, hipStream_t stream) override;
  void do_deallocate(void* p, std::size_t bytes, hipStream_t stream) noexcept override;
};

// Caching pool over hipMalloc: freed blocks are kept in exact-size free lists (sizes rounded to 256 B classes
// below 1 MiB and to 2 MiB multiples above) and handed back stream-ordered — a block freed on stream A and
// reused on stream B makes B wait for the event recorded at the free. This is the default resource: on
// ROCm 7.2 the hipMallocAsync default pool re-maps multi-GB blocks (0.1-0.9 s per 16 GB allocation) once small
// allocations have fragmented it, which dwarfs the 30 ms groupby it serves. 288 GB of HBM3E make holding the
// working set resident the right trade; trim() returns everything to the driver.
class pool_memory_resource final : public device_memory_resource {
 public:
  pool_memory_resource();
  ~pool_memory_resource() override;
  void trim();
  [[nodiscard]] std::size_t cached_bytes() const;

 private:
  void* do_allocate(std::size_t bytes, hipStream_t stream) override;
  void do_deallocate(void* p, std::size_t bytes, hipStream_t stream) noexcept override;
  struct impl;
  impl* _impl;
};

// Counts live/peak bytes on top of another resource (benchmarks and leak tests).
class statistics_resource_adaptor final : public device_memory_resource {
 public:
  explicit statistics_resource_adaptor(device_memory_resource* upstream) : _up{upstream} {}
  [[nodiscard]] std::size_t current_bytes() const { return _cur; }
  [[nodiscard]] std::size_t peak_bytes() const { return _peak; }
  [[nodiscard]] std::size_t allocation_count() const { return _count; }
  void reset_peak() { _peak = _cur.load(); }

 private:
  void* do_allocate(std::size_t bytes, hipStream_t stream) override
  {
    void* p = _up->allocate(bytes, stream);
    std::size_t const now = _cur.fetch_add(bytes) + bytes;
    ++_count;
    std::size_t pk = _peak.load();
    while (now > pk && !_peak.compare_exchange_weak(pk, now)) {}
    return p;
  }
  void do_deallocate(void* p, std::size_t bytes, hipStream_t stream) noexcept override
  {
    _up->deallocate(p, bytes, stream);
    _cur -= bytes;
  }
  device_memory_resource* _up;
  std::atomic<std::size_t> _cur{0}, _peak{0}, _count{0};  // (allocations come from any thread)
};

device_memory_resource* get_current_device_resource();
device_memory_resource* set_current_device_resource(device_memory_resource* mr);
}  // namespace mr

class device_async_resource_ref {
 public:
  device_async_resource_ref(mr::device_memory_resource* r) : _r{r} {}
  device_async_resource_ref(mr::device_memory_resource& r) : _r{&r} {}
  void* allocate_async(std::size_t bytes, hipStream_t s) { return _r->allocate(bytes, s); }
  void deallocate_async(void* p, std::size_t bytes, hipStream_t s) noexcept { _r->deallocate(p, bytes, s); }
  [[nodiscard]] mr::device_memory_resource* resource() const { return _r; }

 private:
  mr::device_memory_resource* _r;
};
}  // namespace rmm

namespace cudf {
inline rmm::device_async_resource_ref get_current_device_resource_ref()
{
  return rmm::device_async_resource_ref{rmm::mr::get_current_device_resource()};
}
}  // namespace cudf
