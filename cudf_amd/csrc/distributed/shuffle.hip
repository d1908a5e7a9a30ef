// SPDX-License-Identifier: Apache-2.0
// Multi-GPU exchange of table rows by key ownership (include/cudf/distributed.hpp): hash-range partition kernels and the
// RCCL point-to-point exchange. Role of the reference's cpp/libcudf_streaming/src/partition_utils.cpp:72-185
// (hash_partition -> pack -> shuffle -> unpack); the split itself restates partitioning.cu:187-337 (histogram, scan,
// LDS-staged scatter) for a handful of destinations and 64-lane waves.
#include "../common/device_table.hpp"
#include "../common/profiler.hpp"

#include <cudf/distributed.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/utilities/error.hpp>

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cudf/join/join.hpp>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <numeric>
#include <string>

namespace cudf {
namespace distributed {
namespace {
using cudf::detail::device_column;
using cudf::detail::device_table;
using cudf::detail::gload;
using cudf::detail::gstore;
using cudf::detail::MAX_COLS;

// ------------------------------------------------------------------ RCCL, resolved at run time
struct rccl_api {
  decltype(&ncclGetUniqueId) GetUniqueId{};
  decltype(&ncclCommInitRank) CommInitRank{};
  decltype(&ncclCommDestroy) CommDestroy{};
  decltype(&ncclAllGather) AllGather{};
  decltype(&ncclSend) Send{};
  decltype(&ncclRecv) Recv{};
  decltype(&ncclGroupStart) GroupStart{};
  decltype(&ncclGroupEnd) GroupEnd{};
  decltype(&ncclGetErrorString) GetErrorString{};
};
rccl_api const& rccl()
{
  static rccl_api api = [] {
    rccl_api a{};
    // a copy already in the process (torch.distributed's) is reused; otherwise ROCm's
    void* h = nullptr;
    for (char const* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h != nullptr) break;
    }
    CUDF_EXPECTS(h != nullptr, "librccl.so could not be loaded: the multi-GPU exchange needs RCCL");
    auto sym = [&](char const* n) {
      void* p = dlsym(h, n);
      CUDF_EXPECTS(p != nullptr, std::string{"librccl.so lacks "} + n);
      return p;
    };
    a.GetUniqueId    = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank   = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy    = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.AllGather      = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
    a.Send           = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv           = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.GroupStart     = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd       = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    return a;
  }();
  return api;
}
#define CUDF_RCCL_TRY(call)                                                                              \
  do {                                                                                                   \
    ncclResult_t const r_ = (call);                                                                      \
    if (r_ != ncclSuccess) CUDF_FAIL(std::string{"RCCL error: "} + rccl().GetErrorString(r_) + " in " #call); \
  } while (0)

// ------------------------------------------------------------------ transports
class rccl_transport final : public transport {
 public:
  rccl_transport(unique_id const& id, int world_size, int rank)
  {
    ncclUniqueId nid;
    std::memcpy(&nid, id.data(), sizeof(nid));
    CUDF_RCCL_TRY(rccl().CommInitRank(&_comm, world_size, nid, rank));
  }
  ~rccl_transport() override
  {
    if (_comm != nullptr) (void)rccl().CommDestroy(_comm);
  }
  void all_gather(void const* send, void* recv, std::size_t count_int64, hipStream_t stream) override
  {
    CUDF_RCCL_TRY(rccl().AllGather(send, recv, count_int64, ncclInt64, _comm, stream));
  }
  void group_start() override { CUDF_RCCL_TRY(rccl().GroupStart()); }
  void send(void const* buf, std::size_t bytes, int peer, hipStream_t stream) override
  {
    CUDF_RCCL_TRY(rccl().Send(buf, bytes, ncclInt8, peer, _comm, stream));
  }
  void recv(void* buf, std::size_t bytes, int peer, hipStream_t stream) override
  {
    CUDF_RCCL_TRY(rccl().Recv(buf, bytes, ncclInt8, peer, _comm, stream));
  }
  void group_end() override { CUDF_RCCL_TRY(rccl().GroupEnd()); }
  [[nodiscard]] void* native_handle() const noexcept override { return _comm; }

 private:
  ncclComm_t _comm{};
};

// Loopback: V virtual ranks in one process on one device, one host thread per rank. A collective is a rendezvous of the V
// threads: every rank drains its stream (its buffers are complete), publishes its pointers, and after the barrier copies what it
// RECEIVES with device copies on its own stream; a second barrier releases the senders' buffers. Sends and receives between two
// ranks match in the order they were posted, as RCCL's do.
struct loopback_hub {
  explicit loopback_hub(int v) : world{v}, ag_send(v, nullptr), sends(v) {}
  struct posted {
    int peer;
    void const* buf;
    std::size_t bytes;
  };
  int const world;
  std::mutex mu;
  std::condition_variable cv;
  int arrived{0};
  std::uint64_t generation{0};
  bool failed{false};
  std::vector<void const*> ag_send;         // all_gather: every rank's send buffer
  std::vector<std::vector<posted>> sends;   // group: what every rank sends, in posting order
  void barrier()
  {
    std::unique_lock<std::mutex> lk{mu};
    CUDF_EXPECTS(!failed, "loopback transport: another rank failed");
    std::uint64_t const gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return;
    }
    // (a deadline as well: a rank that died without telling anybody must not park its peers - and the device buffers they hold - for ever)
    bool const released = cv.wait_for(lk, std::chrono::seconds(barrier_timeout_s()), [&] { return generation != gen || failed; });
    if (!released) {
      failed = true;
      cv.notify_all();
    }
    CUDF_EXPECTS(released, "loopback transport: a rank did not reach the collective within CUDF_AMD_LOOPBACK_TIMEOUT_S");
    CUDF_EXPECTS(!failed || generation != gen, "loopback transport: another rank failed");
  }
  void fail()
  {
    std::lock_guard<std::mutex> lk{mu};
    failed = true;
    cv.notify_all();
  }
  static long barrier_timeout_s()
  {
    static long const v = [] {
      char const* e = std::getenv("CUDF_AMD_LOOPBACK_TIMEOUT_S");
      long const x  = e != nullptr ? std::atol(e) : 0;
      return x > 0 ? x : 120L;
    }();
    return v;
  }
};
class loopback_transport final : public transport {
 public:
  loopback_transport(std::shared_ptr<loopback_hub> hub, int rank) : _hub{std::move(hub)}, _rank{rank} {}
  void all_gather(void const* send, void* recv, std::size_t count_int64, hipStream_t stream) override
  {
    guarded([&] {
      CUDF_HIP_TRY(hipStreamSynchronize(stream));
      _hub->ag_send[_rank] = send;
      _hub->barrier();
      std::size_t const bytes = count_int64 * sizeof(int64_t);
      for (int p = 0; p < _hub->world; ++p)
        CUDF_HIP_TRY(hipMemcpyAsync(static_cast<char*>(recv) + p * bytes, _hub->ag_send[p], bytes, hipMemcpyDeviceToDevice, stream));
      CUDF_HIP_TRY(hipStreamSynchronize(stream));
      _hub->barrier();
    });
  }
  void group_start() override
  {
    CUDF_EXPECTS(!_open, "loopback transport: group already open");
    _open = true;
    _sends.clear();
    _recvs.clear();
    _stream = nullptr;
  }
  void send(void const* buf, std::size_t bytes, int peer, hipStream_t stream) override
  {
    CUDF_EXPECTS(_open && peer >= 0 && peer < _hub->world, "loopback transport: send outside a group or to a rank outside the world");
    _sends.push_back({peer, buf, bytes});
    _stream = stream;
  }
  void recv(void* buf, std::size_t bytes, int peer, hipStream_t stream) override
  {
    CUDF_EXPECTS(_open && peer >= 0 && peer < _hub->world, "loopback transport: recv outside a group or from a rank outside the world");
    _recvs.push_back({peer, buf, bytes});
    _stream = stream;
  }
  void group_end() override
  {
    CUDF_EXPECTS(_open, "loopback transport: no open group");
    _open = false;
    guarded([&] {
      CUDF_HIP_TRY(hipStreamSynchronize(_stream));  // what this rank sends is complete
      _hub->sends[_rank] = _sends;
      _hub->barrier();
      std::vector<std::size_t> next(static_cast<std::size_t>(_hub->world), 0);  // per source: its next unmatched send to this rank
      for (auto const& r : _recvs) {
        auto const& from = _hub->sends[r.peer];
        std::size_t& i   = next[r.peer];
        while (i < from.size() && from[i].peer != _rank) ++i;
        CUDF_EXPECTS(i < from.size(), "loopback transport: a receive without a matching send");
        CUDF_EXPECTS(from[i].bytes == r.bytes, "loopback transport: send and receive sizes differ");
        CUDF_HIP_TRY(hipMemcpyAsync(const_cast<void*>(r.buf), from[i].buf, r.bytes, hipMemcpyDeviceToDevice, _stream));
        ++i;
      }
      for (int p = 0; p < _hub->world; ++p) {  // every send to this rank must have been received
        auto const& from = _hub->sends[p];
        std::size_t i    = next[p];
        while (i < from.size() && from[i].peer != _rank) ++i;
        CUDF_EXPECTS(i == from.size(), "loopback transport: a send without a matching receive");
      }
      CUDF_HIP_TRY(hipStreamSynchronize(_stream));
      _hub->barrier();  // the senders' buffers are free again
    });
  }

 private:
  template <typename F>
  void guarded(F&& f)
  {
    try {
      f();
    } catch (...) {
      _hub->fail();  // the other ranks leave their barriers with an error instead of waiting for ever
      throw;
    }
  }

 public:
  void abort() noexcept override { _hub->fail(); }

 private:
  std::shared_ptr<loopback_hub> _hub;
  int _rank;
  bool _open{false};
  std::vector<loopback_hub::posted> _sends, _recvs;
  hipStream_t _stream{nullptr};
};

// Closes an open group when the scope is left by an exception (a failed Send between GroupStart and GroupEnd used to leave the
// group open for ever).
class group_scope {
 public:
  explicit group_scope(transport& t) : _t{t} { _t.group_start(); }
  void end()
  {
    _open = false;
    _t.group_end();
  }
  ~group_scope()
  {
    if (_open) {
      try {
        _t.group_end();
      } catch (...) {
      }
    }
  }
  group_scope(group_scope const&)            = delete;
  group_scope& operator=(group_scope const&) = delete;

 private:
  transport& _t;
  bool _open{true};
};

// ------------------------------------------------------------------ hash-range partition kernels
constexpr int RP_BLOCK = 1024, RP_R = 4, RP_TILE = RP_BLOCK * RP_R, RP_MAX_PARTS = 64;
struct rp_args {
  device_table hashed;
  device_table all;
  int32_t nparts;
  int32_t wgs;
  int64_t nrows;
  int64_t chunk;       // rows per workgroup (multiple of RP_TILE)
  uint32_t* counts;    // [wgs][nparts]
  int64_t* cell_base;  // [wgs][nparts]
  int64_t* offsets;    // [nparts + 1]
  void* out[MAX_COLS];
};
template <typename T>
__global__ void k_store(T v, T* dst)
{
  *dst = v;
}
__device__ __forceinline__ uint32_t dest_of(rp_args const& a, int64_t row)
{
  uint32_t const h = cudf::detail::row_hash(a.hashed, row, 0u);
  return static_cast<uint32_t>((static_cast<uint64_t>(h) * static_cast<uint32_t>(a.nparts)) >> 32);
}
__global__ void __launch_bounds__(RP_BLOCK) k_rp_hist(rp_args const* __restrict__ ap)
{
  __shared__ uint32_t whist[16 * RP_MAX_PARTS];  // one histogram per wave: 64 lanes on <= 64 addresses
  rp_args const& a = *ap;
  int const N = a.nparts, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * N; i += RP_BLOCK) whist[i] = 0;
  __syncthreads();
  int64_t const begin = static_cast<int64_t>(blockIdx.x) * a.chunk, end = min(a.nrows, begin + a.chunk);
  for (int64_t r = begin + threadIdx.x; r < end; r += RP_BLOCK) atomicAdd(&whist[wave * N + dest_of(a, r)], 1u);
  __syncthreads();
  for (int d = threadIdx.x; d < N; d += RP_BLOCK) {
    uint32_t t = 0;
    for (int w = 0; w < 16; ++w) t += whist[w * N + d];
    a.counts[static_cast<int64_t>(blockIdx.x) * N + d] = t;
  }
}
__global__ void __launch_bounds__(64) k_rp_scan(rp_args const* __restrict__ ap)
{
  // one lane per destination (nparts <= 64): its rows over all workgroups, a wave-wide exclusive scan over the destinations,
  // then the cell bases (one lane walking all wgs * nparts cells was a millisecond of dependent loads on every shuffle)
  rp_args const& a = *ap;
  int const N = a.nparts, d = threadIdx.x;
  int64_t tot = 0;
  if (d < N)
    for (int w = 0; w < a.wgs; ++w) tot += a.counts[static_cast<int64_t>(w) * N + d];
  int64_t inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int64_t const t = __shfl_up(inc, o);
    if (d >= o) inc += t;
  }
  if (d < N) {
    int64_t run  = inc - tot;
    a.offsets[d] = run;
    for (int w = 0; w < a.wgs; ++w) {
      a.cell_base[static_cast<int64_t>(w) * N + d] = run;
      run += a.counts[static_cast<int64_t>(w) * N + d];
    }
    if (d == N - 1) a.offsets[N] = run;
  }
}
template <typename T>
__device__ __forceinline__ void move_column(device_column const& col, void* out, int64_t tile, int64_t end, uint32_t const (&lpos)[RP_R],
                                            T* stage, uint8_t const* sdest, uint32_t const* doff, int64_t const* gcur, uint32_t tile_rows)
{
  T const* src = static_cast<T const*>(col.head) + col.offset;
  T v[RP_R];
#pragma unroll
  for (int k = 0; k < RP_R; ++k) {
    int64_t const r = tile + static_cast<int64_t>(k) * RP_BLOCK + threadIdx.x;
    if (r < end) v[k] = gload(src + r);
  }
#pragma unroll
  for (int k = 0; k < RP_R; ++k) {
    int64_t const r = tile + static_cast<int64_t>(k) * RP_BLOCK + threadIdx.x;
    if (r < end) stage[lpos[k]] = v[k];
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < tile_rows; j += RP_BLOCK) {
    uint32_t const d = sdest[j];
    gstore(static_cast<T*>(out) + gcur[d] + (j - doff[d]), stage[j]);
  }
  __syncthreads();
}
// Columns [c0, c1) of the flat table (every launch ranks the rows again: a pipelined shuffle moves one column per launch so that the
// exchange of column c runs beside the scatter of column c + 1).
__global__ void __launch_bounds__(RP_BLOCK) k_rp_scatter(rp_args const* __restrict__ ap, int c0, int c1)
{
  __shared__ uint64_t stage64[RP_TILE];        // one column's values of a tile, in destination order
  __shared__ uint8_t sdest[RP_TILE];           // destination of every staged position
  __shared__ uint32_t whist[16 * RP_MAX_PARTS];
  __shared__ uint32_t doff[RP_MAX_PARTS + 1];  // first staged position of a destination
  __shared__ int64_t gcur[RP_MAX_PARTS];       // this workgroup's write cursor per destination
  rp_args const& a = *ap;
  int const N = a.nparts, wave = threadIdx.x >> 6;
  int64_t const begin = static_cast<int64_t>(blockIdx.x) * a.chunk, end = min(a.nrows, begin + a.chunk);
  for (int d = threadIdx.x; d < N; d += RP_BLOCK) gcur[d] = a.cell_base[static_cast<int64_t>(blockIdx.x) * N + d];
  for (int64_t tile = begin; tile < end; tile += RP_TILE) {
    for (int i = threadIdx.x; i < 16 * N; i += RP_BLOCK) whist[i] = 0;
    __syncthreads();
    uint32_t dst[RP_R], rank[RP_R], lpos[RP_R];
#pragma unroll
    for (int k = 0; k < RP_R; ++k) {
      int64_t const r = tile + static_cast<int64_t>(k) * RP_BLOCK + threadIdx.x;
      dst[k] = 0; rank[k] = 0;
      if (r < end) {
        dst[k]  = dest_of(a, r);
        rank[k] = atomicAdd(&whist[wave * N + dst[k]], 1u);  // rank inside (wave, destination); waves rank in row order
      }
    }
    __syncthreads();
    // per destination: exclusive prefix over the waves (in place), tile total; then the staged offset of every destination
    if (static_cast<int>(threadIdx.x) < N) {
      uint32_t run = 0;
      for (int w = 0; w < 16; ++w) {
        uint32_t const c = whist[w * N + threadIdx.x];
        whist[w * N + threadIdx.x] = run;
        run += c;
      }
      doff[threadIdx.x + 1] = run;  // (totals for now)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t run = 0;
      for (int d = 0; d < N; ++d) {
        uint32_t const c = doff[d + 1];
        doff[d] = run;
        run += c;
      }
      doff[N] = run;
    }
    __syncthreads();
    uint32_t const tile_rows = doff[N];
#pragma unroll
    for (int k = 0; k < RP_R; ++k) {
      int64_t const r = tile + static_cast<int64_t>(k) * RP_BLOCK + threadIdx.x;
      lpos[k] = 0;
      if (r < end) {
        // rows of one (wave, destination) keep their order: row sets k are ranked one after the other by the same wave
        lpos[k]        = doff[dst[k]] + whist[wave * N + dst[k]] + rank[k];
        sdest[lpos[k]] = static_cast<uint8_t>(dst[k]);
      }
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
      device_column const col = a.all.col[c];
      switch (col.width) {
        case 1: move_column<uint8_t>(col, a.out[c], tile, end, lpos, reinterpret_cast<uint8_t*>(stage64), sdest, doff, gcur, tile_rows); break;
        case 2: move_column<uint16_t>(col, a.out[c], tile, end, lpos, reinterpret_cast<uint16_t*>(stage64), sdest, doff, gcur, tile_rows); break;
        case 4: move_column<uint32_t>(col, a.out[c], tile, end, lpos, reinterpret_cast<uint32_t*>(stage64), sdest, doff, gcur, tile_rows); break;
        default: move_column<uint64_t>(col, a.out[c], tile, end, lpos, stage64, sdest, doff, gcur, tile_rows);
      }
    }
    if (static_cast<int>(threadIdx.x) < N) gcur[threadIdx.x] += doff[threadIdx.x + 1] - doff[threadIdx.x];
    __syncthreads();
  }
}

// validity bits [offset, offset + n) -> one byte per row (1 = valid), and back
__global__ void __launch_bounds__(256) k_mask_to_bytes(bitmask_type const* mask, int64_t offset, int64_t n, uint8_t* out)
{
  int64_t const i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) out[i] = mask == nullptr ? 1 : ((mask[(offset + i) >> 5] >> ((offset + i) & 31)) & 1u);
}
__global__ void __launch_bounds__(256) k_bytes_to_mask(uint8_t const* bytes, int64_t n, bitmask_type* mask, int32_t* null_count)
{
  int64_t const i               = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  bool const valid              = i < n && bytes[i] != 0;
  unsigned long long const b    = __ballot(valid);
  unsigned long long const live = __ballot(i < n);
  int const lane                = threadIdx.x & 63;
  if (i < n && (lane & 31) == 0) mask[i >> 5] = static_cast<uint32_t>(b >> (lane & 32));
  int const nulls = __popcll(live & ~b);
  if (lane == 0 && nulls) atomicAdd(null_count, nulls);
}

// The columns that travel: the input columns, then one validity-byte column per column listed in `with_validity`.
struct flat_table {
  std::vector<column_view> cols;            // what is partitioned / exchanged
  std::vector<int> validity_of;             // cols[ninput + j] holds the validity bytes of input column validity_of[j]
  std::vector<rmm::device_buffer> temps;    // the byte columns' storage
  int ninput{0};
};
flat_table flatten(table_view const& input, uint64_t with_validity, hipStream_t s)
{
  flat_table f;
  f.ninput = input.num_columns();
  auto tmp = cudf::get_current_device_resource_ref();
  for (auto const& c : input) {
    CUDF_EXPECTS(c.num_children() == 0 && size_of_id(c.type().id()) >= 1 && size_of_id(c.type().id()) <= 8,
                 "distributed shuffle: fixed-width columns of at most 8 bytes only");
    f.cols.push_back(column_view{c.type(), c.size(), c.head(), nullptr, 0, c.offset()});
  }
  for (int c = 0; c < f.ninput; ++c) {
    if (!((with_validity >> c) & 1u)) continue;
    auto const& col = input.column(c);
    f.temps.emplace_back(static_cast<std::size_t>(std::max<size_type>(col.size(), 1)), s, tmp);
    if (col.size() > 0) {
      hipLaunchKernelGGL(k_mask_to_bytes, dim3(static_cast<unsigned>((col.size() + 255) / 256)), dim3(256), 0, s,
                         col.has_nulls() ? col.null_mask() : nullptr, static_cast<int64_t>(col.offset()), static_cast<int64_t>(col.size()),
                         static_cast<uint8_t*>(f.temps.back().data()));
    }
    f.cols.push_back(column_view{data_type{type_id::UINT8}, col.size(), f.temps.back().data(), nullptr, 0, 0});
    f.validity_of.push_back(c);
  }
  CUDF_EXPECTS(static_cast<int>(f.cols.size()) <= MAX_COLS, "distributed shuffle: too many columns (limit 16 including validity columns)");
  return f;
}
// buffers of the flat columns (nrows rows each) -> owning table with the input's types and re-packed validity
std::unique_ptr<table> assemble(table_view const& like, flat_table const& f, std::vector<rmm::device_buffer>&& bufs, int64_t nrows,
                                stream_ref stream, rmm::device_async_resource_ref mr)
{
  hipStream_t const s = stream.value();
  std::vector<std::unique_ptr<column>> cols;
  for (int c = 0; c < f.ninput; ++c)
    cols.push_back(std::make_unique<column>(like.column(c).type(), static_cast<size_type>(nrows), std::move(bufs[c]), rmm::device_buffer{}, 0));
  if (!f.validity_of.empty() && nrows > 0) {
    rmm::device_buffer counters{sizeof(int32_t) * f.validity_of.size(), s, cudf::get_current_device_resource_ref()};
    CUDF_HIP_TRY(hipMemsetAsync(counters.data(), 0, counters.size(), s));
    std::vector<rmm::device_buffer> masks;
    for (std::size_t j = 0; j < f.validity_of.size(); ++j) {
      masks.push_back(create_null_mask(static_cast<size_type>(nrows), mask_state::UNINITIALIZED, stream, mr));
      hipLaunchKernelGGL(k_bytes_to_mask, dim3(static_cast<unsigned>((nrows + 255) / 256)), dim3(256), 0, s,
                         static_cast<uint8_t const*>(bufs[f.ninput + j].data()), nrows, static_cast<bitmask_type*>(masks.back().data()),
                         static_cast<int32_t*>(counters.data()) + j);
    }
    CUDF_HIP_TRY(hipGetLastError());
    std::vector<int32_t> h(f.validity_of.size());
    CUDF_HIP_TRY(hipMemcpyAsync(h.data(), counters.data(), counters.size(), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    for (std::size_t j = 0; j < f.validity_of.size(); ++j)
      if (h[j] > 0) cols[f.validity_of[j]]->set_null_mask(std::move(masks[j]), h[j]);
  }
  return std::make_unique<table>(std::move(cols));
}

// The side stream of a pipelined shuffle: column c of the local split is scattered there while the exchange of the columns before it
// runs on the caller's stream; column_ready[c] is recorded behind column c's scatter (reference: the chunked pipeline of
// cpp/libcudf_streaming/src/partition.cpp:20-88 overlaps partition and exchange chunk by chunk; here the unit is a column).
struct scatter_pipeline {
  hipStream_t side{nullptr};
  hipEvent_t scan_done{nullptr};
  std::vector<hipEvent_t> column_ready;
  scatter_pipeline() = default;
  scatter_pipeline(scatter_pipeline const&)            = delete;
  scatter_pipeline& operator=(scatter_pipeline const&) = delete;
  void open(std::size_t ncols)
  {
    CUDF_HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    CUDF_HIP_TRY(hipEventCreateWithFlags(&scan_done, hipEventDisableTiming));
    column_ready.assign(ncols, nullptr);
    for (auto& e : column_ready) CUDF_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  ~scatter_pipeline()
  {
    if (side != nullptr) (void)hipStreamSynchronize(side);  // (nothing of this call may outlive its buffers, whatever path left the scope)
    for (auto e : column_ready)
      if (e != nullptr) (void)hipEventDestroy(e);
    if (scan_done != nullptr) (void)hipEventDestroy(scan_done);
    if (side != nullptr) (void)hipStreamDestroy(side);
  }
};

// Partitions the flat columns by the hash range of `hashed`'s rows; returns one buffer per flat column and N + 1 offsets.
// pipe != nullptr: the offsets are final on return but the columns are not - column c's buffer is complete once pipe->column_ready[c]
// has happened (the caller makes its stream wait for that event before it reads the buffer).
std::pair<std::vector<rmm::device_buffer>, std::vector<int64_t>> partition_flat(flat_table const& f, table_view const& hashed, int N,
                                                                                stream_ref stream, rmm::device_async_resource_ref mr,
                                                                                scatter_pipeline* pipe = nullptr)
{
  CUDF_EXPECTS(N >= 1 && N <= RP_MAX_PARTS, "distributed shuffle: 1 to 64 destinations", std::invalid_argument);
  hipStream_t const s = stream.value();
  int64_t const n     = f.cols.empty() ? 0 : f.cols.front().size();
  std::vector<rmm::device_buffer> out;
  for (auto const& c : f.cols) out.emplace_back(static_cast<std::size_t>(std::max<int64_t>(n, 1)) * size_of_id(c.type().id()), s, mr);
  std::vector<int64_t> h_off(static_cast<std::size_t>(N) + 1, 0);
  auto all_ready_on_main = [&] {  // (no side work: every column is ready where the caller's stream stands)
    if (pipe != nullptr)
      for (auto e : pipe->column_ready) CUDF_HIP_TRY(hipEventRecord(e, s));
  };
  if (n == 0) {
    all_ready_on_main();
    return {std::move(out), std::move(h_off)};
  }
  if (N == 1) {  // one destination: the rows stay as they are
    for (std::size_t c = 0; c < f.cols.size(); ++c) {
      auto const w = size_of_id(f.cols[c].type().id());
      CUDF_HIP_TRY(hipMemcpyAsync(out[c].data(), static_cast<char const*>(f.cols[c].head()) + static_cast<std::size_t>(f.cols[c].offset()) * w,
                                  static_cast<std::size_t>(n) * w, hipMemcpyDeviceToDevice, s));
    }
    h_off[1] = n;
    all_ready_on_main();
    return {std::move(out), std::move(h_off)};
  }
  auto tmp = cudf::get_current_device_resource_ref();
  rp_args a{};
  a.hashed = cudf::detail::make_device_table(hashed);
  a.all    = cudf::detail::make_device_table(table_view{f.cols});
  a.nparts = N;
  a.nrows  = n;
  a.wgs    = static_cast<int32_t>(std::clamp<int64_t>((n + RP_TILE - 1) / RP_TILE, 1, 512));
  a.chunk  = ((n + a.wgs - 1) / a.wgs + RP_TILE - 1) / RP_TILE * RP_TILE;
  rmm::device_buffer counts{sizeof(uint32_t) * a.wgs * N, s, tmp}, cell{sizeof(int64_t) * a.wgs * N, s, tmp},
    offs{sizeof(int64_t) * (N + 1), s, tmp}, d_args{sizeof(rp_args), s, tmp};
  a.counts    = static_cast<uint32_t*>(counts.data());
  a.cell_base = static_cast<int64_t*>(cell.data());
  a.offsets   = static_cast<int64_t*>(offs.data());
  for (std::size_t c = 0; c < f.cols.size(); ++c) a.out[c] = out[c].data();
  auto* da = static_cast<rp_args*>(d_args.data());
  hipLaunchKernelGGL(k_store<rp_args>, dim3(1), dim3(1), 0, s, a, da);
  {
    cudf::detail::prof::scope p_{"range_partition_hist", s};
    hipLaunchKernelGGL(k_rp_hist, dim3(a.wgs), dim3(RP_BLOCK), 0, s, da);
  }
  hipLaunchKernelGGL(k_rp_scan, dim3(1), dim3(64), 0, s, da);
  if (pipe == nullptr) {
    cudf::detail::prof::scope p_{"range_partition_scatter", s};
    hipLaunchKernelGGL(k_rp_scatter, dim3(a.wgs), dim3(RP_BLOCK), 0, s, da, 0, a.all.ncols);
  } else {
    // one launch per column on the side stream, behind the scan; the caller's stream goes on to the count exchange
    CUDF_HIP_TRY(hipEventRecord(pipe->scan_done, s));
    CUDF_HIP_TRY(hipStreamWaitEvent(pipe->side, pipe->scan_done, 0));
    for (int c = 0; c < a.all.ncols; ++c) {
      hipLaunchKernelGGL(k_rp_scatter, dim3(a.wgs), dim3(RP_BLOCK), 0, pipe->side, da, c, c + 1);
      CUDF_HIP_TRY(hipEventRecord(pipe->column_ready[static_cast<std::size_t>(c)], pipe->side));
    }
  }
  CUDF_HIP_TRY(hipGetLastError());
  CUDF_HIP_TRY(hipMemcpyAsync(h_off.data(), a.offsets, sizeof(int64_t) * (N + 1), hipMemcpyDeviceToHost, s));
  CUDF_HIP_TRY(hipStreamSynchronize(s));
  if (pipe != nullptr) {
    // the scratch of this function (histogram cells, argument block) is read by the side stream's launches: its stream-ordered release
    // on `s` waits for the last of them
    CUDF_HIP_TRY(hipStreamWaitEvent(s, pipe->column_ready.back(), 0));
  }
  return {std::move(out), std::move(h_off)};
}

uint64_t nullable_bits(table_view const& t)
{
  uint64_t b = 0;
  for (int c = 0; c < t.num_columns(); ++c)
    if (t.column(c).has_nulls()) b |= uint64_t{1} << c;
  return b;
}
}  // namespace

// ------------------------------------------------------------------ communicator
unique_id communicator::make_unique_id()
{
  ncclUniqueId id;
  CUDF_RCCL_TRY(rccl().GetUniqueId(&id));
  unique_id out{};
  static_assert(sizeof(id) == UNIQUE_ID_BYTES);
  std::memcpy(out.data(), &id, sizeof(id));
  return out;
}
communicator::communicator(std::unique_ptr<transport> t, int world_size, int rank) : _transport{std::move(t)}, _world{world_size}, _rank{rank} {}
communicator::communicator(unique_id const& id, int world_size, int rank) : _world{world_size}, _rank{rank}
{
  CUDF_EXPECTS(world_size >= 1 && rank >= 0 && rank < world_size, "communicator: rank outside the world", std::invalid_argument);
  _transport = std::make_unique<rccl_transport>(id, world_size, rank);
}
std::vector<std::unique_ptr<communicator>> communicator::make_loopback(int world_size)
{
  CUDF_EXPECTS(world_size >= 1 && world_size <= RP_MAX_PARTS, "communicator: a loopback world of 1 to 64 ranks", std::invalid_argument);
  auto hub = std::make_shared<loopback_hub>(world_size);
  std::vector<std::unique_ptr<communicator>> out;
  for (int r = 0; r < world_size; ++r)
    out.push_back(std::unique_ptr<communicator>{new communicator{std::make_unique<loopback_transport>(hub, r), world_size, r}});
  return out;
}
communicator::~communicator() = default;
void communicator::set_max_message_bytes(std::int64_t bytes)
{
  CUDF_EXPECTS(bytes >= 8, "communicator: a message holds at least one 8-byte element", std::invalid_argument);
  _max_message_bytes = bytes;
}

exchange_plan plan_exchange(std::vector<std::int64_t> const& counts, int world, int me)
{
  CUDF_EXPECTS(world >= 1 && me >= 0 && me < world && counts.size() == static_cast<std::size_t>(world) * world,
               "plan_exchange: a world x world count matrix and a rank inside the world", std::invalid_argument);
  exchange_plan ep;
  ep.recv_count.resize(world);
  ep.recv_offset.assign(static_cast<std::size_t>(world) + 1, 0);
  for (int p = 0; p < world; ++p) {
    ep.recv_count[p]      = counts[static_cast<std::size_t>(p) * world + me];
    ep.recv_offset[p + 1] = ep.recv_offset[p] + ep.recv_count[p];
    for (int q = 0; q < world; ++q)
      if (p != q) ep.biggest = std::max(ep.biggest, counts[static_cast<std::size_t>(p) * world + q]);  // (the own slice is a device copy)
  }
  return ep;
}

// ------------------------------------------------------------------ range_partition
std::pair<std::unique_ptr<table>, std::vector<size_type>> range_partition(table_view const& input, std::vector<size_type> const& key_columns,
                                                                          int num_destinations, stream_ref stream,
                                                                          rmm::device_async_resource_ref mr)
{
  auto const hashed = input.select(key_columns);
  CUDF_EXPECTS(hashed.num_columns() >= 1, "range_partition: at least one key column", std::invalid_argument);
  auto f            = flatten(input, nullable_bits(input), stream.value());
  auto [bufs, off]  = partition_flat(f, hashed, num_destinations, stream, mr);
  std::vector<size_type> offsets(off.begin(), off.end());
  return {assemble(input, f, std::move(bufs), input.num_rows(), stream, mr), std::move(offsets)};
}

// ------------------------------------------------------------------ shuffle
namespace {
std::unique_ptr<table> shuffle_body(table_view const& input, std::vector<size_type> const& key_columns, communicator& comm, stream_ref stream,
                                    rmm::device_async_resource_ref mr);
}
std::unique_ptr<table> shuffle(table_view const& input, std::vector<size_type> const& key_columns, communicator& comm, stream_ref stream,
                               rmm::device_async_resource_ref mr)
{
  CUDF_FUNC_RANGE();
  try {
    return shuffle_body(input, key_columns, comm, stream, mr);
  } catch (...) {
    comm.link().abort();  // (validation, an allocation, a too-long receive: the peers are told instead of waiting in the next collective)
    throw;
  }
}
namespace {
std::unique_ptr<table> shuffle_body(table_view const& input, std::vector<size_type> const& key_columns, communicator& comm, stream_ref stream,
                                    rmm::device_async_resource_ref mr)
{
  hipStream_t const s = stream.value();
  int const N = comm.size(), me = comm.rank();
  transport& link = comm.link();
  auto tmp        = cudf::get_current_device_resource_ref();
  CUDF_EXPECTS(input.num_columns() >= 1 && input.num_columns() < 60, "shuffle: 1 to 59 columns", std::invalid_argument);
  auto const hashed = input.select(key_columns);
  CUDF_EXPECTS(hashed.num_columns() >= 1, "shuffle: at least one key column", std::invalid_argument);

  // ---- round 0 of the control traffic: which columns carry nulls on ANY rank (they travel with validity bytes everywhere)
  rmm::device_buffer ctl_send{sizeof(int64_t) * (N + 1), s, tmp}, ctl_recv{sizeof(int64_t) * (N + 1) * N, s, tmp};
  std::vector<int64_t> h_send(static_cast<std::size_t>(N) + 1, 0), h_all(static_cast<std::size_t>(N + 1) * N, 0);
  h_send[N] = static_cast<int64_t>(nullable_bits(input));
  CUDF_HIP_TRY(hipMemcpyAsync(ctl_send.data(), h_send.data(), ctl_send.size(), hipMemcpyHostToDevice, s));
  link.all_gather(ctl_send.data(), ctl_recv.data(), static_cast<std::size_t>(N) + 1, s);
  CUDF_HIP_TRY(hipMemcpyAsync(h_all.data(), ctl_recv.data(), ctl_recv.size(), hipMemcpyDeviceToHost, s));
  CUDF_HIP_TRY(hipStreamSynchronize(s));
  uint64_t with_validity = 0;
  for (int p = 0; p < N; ++p) with_validity |= static_cast<uint64_t>(h_all[static_cast<std::size_t>(p) * (N + 1) + N]);

  // ---- local split by owner rank
  auto f = flatten(input, with_validity, s);
  // (the columns are scattered one per launch on a side stream: the count exchange below and the payload rounds of column c run
  // beside the scatter of the columns after c. CUDF_AMD_SHUFFLE_PIPELINE=0: one fused scatter on the caller's stream.)
  static bool const pipelined = [] {
    char const* e = std::getenv("CUDF_AMD_SHUFFLE_PIPELINE");
    return e == nullptr || std::atoi(e) != 0;
  }();
  scatter_pipeline pipe;  // (partition_flat returns with `s` waiting for the last column's event: buffers released on `s` are safe)
  if (pipelined && f.cols.size() > 1) pipe.open(f.cols.size());
  auto [bufs, off] = partition_flat(f, hashed, N, stream, tmp, pipe.side != nullptr ? &pipe : nullptr);

  // ---- counts: every rank learns what every rank sends to every rank
  for (int p = 0; p < N; ++p) h_send[p] = off[p + 1] - off[p];
  CUDF_HIP_TRY(hipMemcpyAsync(ctl_send.data(), h_send.data(), ctl_send.size(), hipMemcpyHostToDevice, s));
  link.all_gather(ctl_send.data(), ctl_recv.data(), static_cast<std::size_t>(N) + 1, s);
  CUDF_HIP_TRY(hipMemcpyAsync(h_all.data(), ctl_recv.data(), ctl_recv.size(), hipMemcpyDeviceToHost, s));
  CUDF_HIP_TRY(hipStreamSynchronize(s));
  std::vector<int64_t> matrix(static_cast<std::size_t>(N) * N);
  for (int p = 0; p < N; ++p)
    for (int q = 0; q < N; ++q) matrix[static_cast<std::size_t>(p) * N + q] = h_all[static_cast<std::size_t>(p) * (N + 1) + q];
  exchange_plan const ep    = plan_exchange(matrix, N, me);
  auto const& rcnt          = ep.recv_count;
  auto const& roff          = ep.recv_offset;
  int64_t const biggest     = ep.biggest;  // the largest (sender, receiver) message in rows: every rank runs the same number of rounds
  int64_t const total       = roff[N];
  CUDF_EXPECTS(total <= std::numeric_limits<size_type>::max(), "shuffle: a rank would receive more rows than a column holds", std::overflow_error);

  // ---- payload: per column, group_start { send / recv per peer } group_end in rounds of bounded messages (RCCL: ncclGroupStart /
  // ncclSend / ncclRecv / ncclGroupEnd); the own slice is a device copy
  int64_t const max_message_bytes = comm.max_message_bytes();
  std::vector<rmm::device_buffer> recv;
  cudf::detail::prof::scope p_{"shuffle_exchange", s};
  for (std::size_t c = 0; c < f.cols.size(); ++c) {
    int64_t const w = static_cast<int64_t>(size_of_id(f.cols[c].type().id()));
    if (pipe.side != nullptr) CUDF_HIP_TRY(hipStreamWaitEvent(s, pipe.column_ready[c], 0));  // column c has been scattered
    recv.emplace_back(static_cast<std::size_t>(std::max<int64_t>(total, 1) * w), s, c < static_cast<std::size_t>(f.ninput) ? mr : tmp);
    char const* src = static_cast<char const*>(bufs[c].data());
    char* dst       = static_cast<char*>(recv.back().data());
    if (h_send[me] > 0)
      CUDF_HIP_TRY(hipMemcpyAsync(dst + roff[me] * w, src + off[me] * w, static_cast<std::size_t>(h_send[me] * w), hipMemcpyDeviceToDevice, s));
    int64_t const chunk = std::max<int64_t>(1, max_message_bytes / w);
    for (int64_t done = 0; done < biggest; done += chunk) {
      group_scope group{link};  // (closed on every path out of this scope: a failed send does not leave the group open)
      for (int p = 0; p < N; ++p) {
        if (p == me) continue;
        int64_t const sb = std::min(done, h_send[p]), se = std::min(done + chunk, h_send[p]);
        if (se > sb) link.send(src + (off[p] + sb) * w, static_cast<std::size_t>((se - sb) * w), p, s);
        int64_t const rb = std::min(done, rcnt[p]), re = std::min(done + chunk, rcnt[p]);
        if (re > rb) link.recv(dst + (roff[p] + rb) * w, static_cast<std::size_t>((re - rb) * w), p, s);
      }
      group.end();
    }
  }
  return assemble(input, f, std::move(recv), total, stream, mr);
}
}  // namespace

// ------------------------------------------------------------------ shuffle_groupby (BASELINE config 5)
std::pair<std::unique_ptr<table>, std::vector<groupby::aggregation_result>> shuffle_groupby(
  table_view const& keys, std::span<groupby::aggregation_request const> requests, communicator& comm, null_policy null_handling,
  stream_ref stream, rmm::device_async_resource_ref mr)
{
  CUDF_FUNC_RANGE();
  // the rows that travel: the key columns, then each DISTINCT value column once
  std::vector<column_view> cols(keys.begin(), keys.end());
  std::vector<size_type> key_idx(static_cast<std::size_t>(keys.num_columns()));
  std::iota(key_idx.begin(), key_idx.end(), 0);
  std::vector<int> value_at(requests.size(), -1);
  for (std::size_t r = 0; r < requests.size(); ++r) {
    CUDF_EXPECTS(requests[r].values.size() == keys.num_rows(), "Size mismatch between request values and groupby keys.");
    for (std::size_t q = 0; q < r; ++q)
      if (cudf::detail::is_shallow_equivalent(requests[q].values, requests[r].values)) value_at[r] = value_at[q];
    if (value_at[r] < 0) {
      value_at[r] = static_cast<int>(cols.size());
      cols.push_back(requests[r].values);
    }
  }
  auto mine = shuffle(table_view{cols}, key_idx, comm, stream, cudf::get_current_device_resource_ref());
  auto const mv = mine->view();
  std::vector<column_view> kcols;
  for (int c = 0; c < keys.num_columns(); ++c) kcols.push_back(mv.column(c));
  std::vector<groupby::aggregation_request> local(requests.size());
  for (std::size_t r = 0; r < requests.size(); ++r) {
    local[r].values = mv.column(value_at[r]);
    for (auto const& a : requests[r].aggregations) {
      auto cl = a->clone();
      auto* g = dynamic_cast<groupby_aggregation*>(cl.get());
      CUDF_EXPECTS(g != nullptr, "not a groupby aggregation");
      (void)cl.release();
      local[r].aggregations.emplace_back(g);
    }
  }
  groupby::groupby gb{table_view{kcols}, null_handling};
  return gb.aggregate(local, stream, mr);
}

// ------------------------------------------------------------------ combine_groupby (the decomposable form of config 5)
namespace {
__global__ void __launch_bounds__(256) k_count_to_i32(int64_t const* in, int64_t n, int32_t* out)
{
  int64_t const i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) out[i] = static_cast<int32_t>(in[i]);
}
// mean = sum / count as FLOAT64 (the reference's MEAN target type for numeric columns); valid where count > 0
template <typename S>
__global__ void __launch_bounds__(256) k_mean_of(S const* sum, int64_t const* count, int64_t n, double* out, bitmask_type* mask, int32_t* nulls)
{
  int64_t const i  = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  bool const valid = i < n && count[i] > 0;
  if (i < n) out[i] = valid ? static_cast<double>(sum[i]) / static_cast<double>(count[i]) : 0.0;
  uint64_t const b = __ballot(valid);
  int const lane   = threadIdx.x & 63;
  if (i < n && (lane & 31) == 0) mask[i >> 5] = static_cast<bitmask_type>(lane == 0 ? b : b >> 32);
  if (lane == 0) {
    int64_t const in_range = min<int64_t>(64, n - (i - lane));
    int const missing      = static_cast<int>(in_range > 0 ? in_range : 0) - __popcll(b);
    if (missing > 0) atomicAdd(nulls, missing);
  }
}
std::unique_ptr<groupby_aggregation> clone_groupby_agg(aggregation const& a)
{
  auto cl = a.clone();
  auto* g = dynamic_cast<groupby_aggregation*>(cl.get());
  CUDF_EXPECTS(g != nullptr, "not a groupby aggregation");
  (void)cl.release();
  return std::unique_ptr<groupby_aggregation>(g);
}
// the partial aggregations of one request and how the merge treats each
struct partial_agg {
  aggregation::Kind local;   // what the local groupby computes
  aggregation::Kind merge;   // what the merge groupby applies to that partial column
};
}  // namespace

std::pair<std::unique_ptr<table>, std::vector<groupby::aggregation_result>> combine_groupby(
  table_view const& keys, std::span<groupby::aggregation_request const> requests, communicator& comm, null_policy null_handling,
  stream_ref stream, rmm::device_async_resource_ref mr)
{
  CUDF_FUNC_RANGE();
  using K = aggregation::Kind;
  hipStream_t const s = stream.value();
  auto tmp            = cudf::get_current_device_resource_ref();
  // ---- lowering (host only: every rank reaches the same verdict before anything is exchanged)
  struct use {
    int partial[2];  // indices into the request's partial list (MEAN: sum, count)
  };
  std::vector<std::vector<partial_agg>> partials(requests.size());
  std::vector<std::vector<use>> uses(requests.size());
  for (std::size_t r = 0; r < requests.size(); ++r) {
    CUDF_EXPECTS(requests[r].values.size() == keys.num_rows(), "Size mismatch between request values and groupby keys.");
    auto find_or_add = [&](K local, K merge) {
      for (std::size_t i = 0; i < partials[r].size(); ++i)
        if (partials[r][i].local == local) return static_cast<int>(i);
      partials[r].push_back({local, merge});
      return static_cast<int>(partials[r].size()) - 1;
    };
    for (auto const& a : requests[r].aggregations) {
      use u{{-1, -1}};
      switch (a->kind) {
        case K::SUM: u.partial[0] = find_or_add(K::SUM, K::SUM); break;
        case K::PRODUCT: u.partial[0] = find_or_add(K::PRODUCT, K::PRODUCT); break;
        case K::SUM_OF_SQUARES: u.partial[0] = find_or_add(K::SUM_OF_SQUARES, K::SUM); break;
        case K::MIN: u.partial[0] = find_or_add(K::MIN, K::MIN); break;
        case K::MAX: u.partial[0] = find_or_add(K::MAX, K::MAX); break;
        case K::COUNT_VALID: u.partial[0] = find_or_add(K::COUNT_VALID, K::SUM); break;
        case K::COUNT_ALL: u.partial[0] = find_or_add(K::COUNT_ALL, K::SUM); break;
        case K::MEAN: {
          auto const id = requests[r].values.type().id();
          bool const numeric = id == type_id::INT8 || id == type_id::INT16 || id == type_id::INT32 || id == type_id::INT64 || id == type_id::UINT8 ||
                               id == type_id::UINT16 || id == type_id::UINT32 || id == type_id::UINT64 || id == type_id::FLOAT32 || id == type_id::FLOAT64;
          CUDF_EXPECTS(numeric,
                       "combine_groupby: MEAN of a numeric column only (decimal / duration means divide in integers)", std::invalid_argument);
          u.partial[0] = find_or_add(K::SUM, K::SUM);
          u.partial[1] = find_or_add(K::COUNT_VALID, K::SUM);
          break;
        }
        default:
          CUDF_FAIL("combine_groupby: this aggregation does not decompose into per-rank partials merged by SUM / MIN / MAX / PRODUCT", std::invalid_argument);
      }
      uses[r].push_back(u);
    }
  }
  auto make_kind = [](K k) -> std::unique_ptr<groupby_aggregation> {
    switch (k) {
      case K::SUM: return make_sum_aggregation<groupby_aggregation>();
      case K::PRODUCT: return make_product_aggregation<groupby_aggregation>();
      case K::SUM_OF_SQUARES: return make_sum_of_squares_aggregation<groupby_aggregation>();
      case K::MIN: return make_min_aggregation<groupby_aggregation>();
      case K::MAX: return make_max_aggregation<groupby_aggregation>();
      case K::COUNT_VALID: return make_count_aggregation<groupby_aggregation>(null_policy::EXCLUDE);
      case K::COUNT_ALL: return make_count_aggregation<groupby_aggregation>(null_policy::INCLUDE);
      default: CUDF_FAIL("combine_groupby: partial kind");
    }
  };
  // ---- 1. local groupby with the partial aggregations
  std::vector<groupby::aggregation_request> local(requests.size());
  for (std::size_t r = 0; r < requests.size(); ++r) {
    local[r].values = requests[r].values;
    for (auto const& pa : partials[r]) local[r].aggregations.push_back(make_kind(pa.local));
  }
  std::unique_ptr<table> lkeys;
  std::vector<groupby::aggregation_result> lres;
  try {
    groupby::groupby gb{keys, null_handling};
    std::tie(lkeys, lres) = gb.aggregate(local, stream, tmp);
  } catch (...) {
    comm.link().abort();  // (the peers are about to enter the exchange)
    throw;
  }
  // ---- 2. the partial groups travel to the rank that owns their key
  std::vector<column_view> cols;
  std::vector<size_type> key_idx(static_cast<std::size_t>(keys.num_columns()));
  std::iota(key_idx.begin(), key_idx.end(), 0);
  for (int c = 0; c < lkeys->num_columns(); ++c) cols.push_back(lkeys->view().column(c));
  std::vector<std::vector<int>> col_of(requests.size());
  for (std::size_t r = 0; r < requests.size(); ++r)
    for (std::size_t i = 0; i < partials[r].size(); ++i) {
      col_of[r].push_back(static_cast<int>(cols.size()));
      cols.push_back(lres[r].results[i]->view());
    }
  auto mine     = shuffle(table_view{cols}, key_idx, comm, stream, tmp);
  auto const mv = mine->view();
  // ---- 3. merge on the owner: one request per partial column
  std::vector<column_view> kcols;
  for (int c = 0; c < keys.num_columns(); ++c) kcols.push_back(mv.column(c));
  std::vector<groupby::aggregation_request> merge;
  std::vector<std::vector<int>> merged_at(requests.size());
  for (std::size_t r = 0; r < requests.size(); ++r)
    for (std::size_t i = 0; i < partials[r].size(); ++i) {
      merged_at[r].push_back(static_cast<int>(merge.size()));
      groupby::aggregation_request q;
      q.values = mv.column(col_of[r][i]);
      q.aggregations.push_back(make_kind(partials[r][i].merge));
      merge.push_back(std::move(q));
    }
  groupby::groupby mg{table_view{kcols}, null_handling};
  auto [out_keys, mres] = mg.aggregate(merge, stream, mr);
  size_type const G     = out_keys->num_rows();
  // ---- 4. finalisation in request order (a partial used twice is copied the second time)
  std::vector<std::vector<bool>> taken(requests.size());
  for (std::size_t r = 0; r < requests.size(); ++r) taken[r].assign(partials[r].size(), false);
  auto take = [&](std::size_t r, int i) -> std::unique_ptr<column> {
    auto& slot = mres[static_cast<std::size_t>(merged_at[r][static_cast<std::size_t>(i)])].results[0];
    if (!taken[r][static_cast<std::size_t>(i)]) {
      taken[r][static_cast<std::size_t>(i)] = true;
      return std::make_unique<column>(slot->view(), stream, mr);  // (kept: a later MEAN / repeat may read it; copies are G rows)
    }
    return std::make_unique<column>(slot->view(), stream, mr);
  };
  auto view_of = [&](std::size_t r, int i) { return mres[static_cast<std::size_t>(merged_at[r][static_cast<std::size_t>(i)])].results[0]->view(); };
  std::vector<groupby::aggregation_result> out(requests.size());
  int const blocks = static_cast<int>((static_cast<int64_t>(G) + 255) / 256);
  for (std::size_t r = 0; r < requests.size(); ++r) {
    for (std::size_t j = 0; j < requests[r].aggregations.size(); ++j) {
      K const k    = requests[r].aggregations[j]->kind;
      use const& u = uses[r][j];
      if (k == K::COUNT_VALID || k == K::COUNT_ALL) {  // merged by SUM: INT64 -> the reference's INT32 counts
        auto c = make_fixed_width_column(data_type{type_id::INT32}, G, mask_state::UNALLOCATED, stream, mr);
        if (G > 0) {
          hipLaunchKernelGGL(k_count_to_i32, dim3(blocks), dim3(256), 0, s, view_of(r, u.partial[0]).data<int64_t>(), static_cast<int64_t>(G),
                             c->mutable_view().data<int32_t>());
          CUDF_HIP_TRY(hipGetLastError());
        }
        out[r].results.push_back(std::move(c));
      } else if (k == K::MEAN) {
        auto c = make_fixed_width_column(data_type{type_id::FLOAT64}, G, mask_state::UNINITIALIZED, stream, mr);
        int32_t h_nulls = 0;
        if (G > 0) {
          rmm::device_buffer d_nulls{sizeof(int32_t), s, tmp};
          CUDF_HIP_TRY(hipMemsetAsync(d_nulls.data(), 0, sizeof(int32_t), s));
          auto const sv = view_of(r, u.partial[0]);
          auto const cv = view_of(r, u.partial[1]);
          auto mvw      = c->mutable_view();
          auto launch   = [&](auto const* sum) {
            hipLaunchKernelGGL(k_mean_of, dim3(blocks), dim3(256), 0, s, sum, cv.data<int64_t>(), static_cast<int64_t>(G), mvw.data<double>(),
                               mvw.null_mask(), static_cast<int32_t*>(d_nulls.data()));
          };
          switch (sv.type().id()) {
            case type_id::INT64: launch(sv.data<int64_t>()); break;
            case type_id::UINT64: launch(sv.data<uint64_t>()); break;
            case type_id::FLOAT64: launch(sv.data<double>()); break;
            case type_id::FLOAT32: launch(sv.data<float>()); break;
            default: CUDF_FAIL("combine_groupby: sum type of a MEAN");
          }
          CUDF_HIP_TRY(hipGetLastError());
          CUDF_HIP_TRY(hipMemcpyAsync(&h_nulls, d_nulls.data(), sizeof(int32_t), hipMemcpyDeviceToHost, s));
          CUDF_HIP_TRY(hipStreamSynchronize(s));
        }
        if (h_nulls == 0) c->set_null_mask(rmm::device_buffer{}, 0);
        else c->set_null_count(h_nulls);
        out[r].results.push_back(std::move(c));
      } else {
        out[r].results.push_back(take(r, u.partial[0]));
      }
    }
  }
  return {std::move(out_keys), std::move(out)};
}

// ------------------------------------------------------------------ shuffle_join
namespace {
__global__ void __launch_bounds__(256) k_global_ids(int64_t first, int64_t n, int64_t* out)
{
  int64_t const i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) out[i] = first + i;
}
__global__ void __launch_bounds__(256) k_gather_ids(size_type const* idx, int64_t const* ids, int64_t n, int64_t* out)
{
  int64_t const i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) out[i] = ids[idx[i]];
}
// One side of a distributed join: (keys, global row id) shuffled by key; returns the table this rank owns (keys first, id last).
std::unique_ptr<table> shuffle_side(table_view const& keys, communicator& comm, stream_ref stream)
{
  hipStream_t const s = stream.value();
  int const N = comm.size(), me = comm.rank();
  auto tmp = cudf::get_current_device_resource_ref();
  // this rank's first global row id = rows of the lower ranks
  rmm::device_buffer d_mine{sizeof(int64_t), s, tmp}, d_all{sizeof(int64_t) * N, s, tmp};
  int64_t const mine = keys.num_rows();
  std::vector<int64_t> h_all(N, 0);
  CUDF_HIP_TRY(hipMemcpyAsync(d_mine.data(), &mine, sizeof(int64_t), hipMemcpyHostToDevice, s));
  comm.link().all_gather(d_mine.data(), d_all.data(), 1, s);
  CUDF_HIP_TRY(hipMemcpyAsync(h_all.data(), d_all.data(), sizeof(int64_t) * N, hipMemcpyDeviceToHost, s));
  CUDF_HIP_TRY(hipStreamSynchronize(s));
  int64_t first = 0;
  for (int p = 0; p < me; ++p) first += h_all[p];
  rmm::device_buffer ids{sizeof(int64_t) * static_cast<std::size_t>(std::max<int64_t>(mine, 1)), s, tmp};
  if (mine > 0)
    hipLaunchKernelGGL(k_global_ids, dim3(static_cast<unsigned>((mine + 255) / 256)), dim3(256), 0, s, first, mine, static_cast<int64_t*>(ids.data()));
  CUDF_HIP_TRY(hipGetLastError());
  std::vector<column_view> cols(keys.begin(), keys.end());
  cols.push_back(column_view{data_type{type_id::INT64}, static_cast<size_type>(mine), ids.data(), nullptr, 0, 0});
  std::vector<size_type> key_idx(static_cast<std::size_t>(keys.num_columns()));
  std::iota(key_idx.begin(), key_idx.end(), 0);
  return shuffle(table_view{cols}, key_idx, comm, stream, tmp);
}
}  // namespace

std::pair<std::unique_ptr<column>, std::unique_ptr<column>> shuffle_join(table_view const& left_keys, table_view const& right_keys,
                                                                         communicator& comm, null_equality compare_nulls, stream_ref stream,
                                                                         rmm::device_async_resource_ref mr)
{
  CUDF_FUNC_RANGE();
  CUDF_EXPECTS(left_keys.num_columns() >= 1 && left_keys.num_columns() == right_keys.num_columns(),
               "shuffle_join: the two sides need the same number (at least one) of key columns", std::invalid_argument);
  hipStream_t const s = stream.value();
  auto const L = shuffle_side(left_keys, comm, stream);
  auto const R = shuffle_side(right_keys, comm, stream);
  int const nk = left_keys.num_columns();
  std::vector<column_view> lk, rk;
  for (int c = 0; c < nk; ++c) {
    lk.push_back(L->view().column(c));
    rk.push_back(R->view().column(c));
  }
  auto [li, ri] = cudf::inner_join(table_view{lk}, table_view{rk}, compare_nulls, stream, cudf::get_current_device_resource_ref());
  auto const m  = static_cast<int64_t>(li->size());
  auto out_col  = [&](rmm::device_uvector<size_type> const& idx, column_view const& ids) {
    rmm::device_buffer buf{sizeof(int64_t) * static_cast<std::size_t>(std::max<int64_t>(m, 1)), s, mr};
    if (m > 0)
      hipLaunchKernelGGL(k_gather_ids, dim3(static_cast<unsigned>((m + 255) / 256)), dim3(256), 0, s, idx.data(), ids.data<int64_t>(), m,
                         static_cast<int64_t*>(buf.data()));
    CUDF_HIP_TRY(hipGetLastError());
    return std::make_unique<column>(data_type{type_id::INT64}, static_cast<size_type>(m), std::move(buf), rmm::device_buffer{}, 0);
  };
  auto lo = out_col(*li, L->view().column(nk));
  auto ro = out_col(*ri, R->view().column(nk));
  CUDF_HIP_TRY(hipStreamSynchronize(s));  // (the shuffled sides and the index vectors go back to the pool when this returns)
  return {std::move(lo), std::move(ro)};
}

}  // namespace distributed
}  // namespace cudf
