// SPDX-License-Identifier: Apache-2.0
// cudf::hash_partition, cudf::hashing::murmurhash3_x86_32 and cudf::gather for fixed-width tables on gfx950.
// Reference: cpp/src/partitioning/partitioning.cu:54-92 (partition = row_hash % P, power-of-two fast path),
// :119-171 (compute_row_partition_numbers), :195-235 (compute_row_output_locations), :252-337
// (copy_block_partitions), :569-760 (driver); gather: cpp/include/cudf/detail/gather.cuh:119-127,525.
// MI355X design: one histogram pass over the hashed columns, one scan, then ONE multi-split pass that moves
// every column through an LDS staging tile so that each partition's rows leave as contiguous runs
// (64-lane waves, 4096-row tiles); the gather map of the split is produced in the same pass and only used
// for validity bits.
#include "../common/device_table.hpp"
#include "../common/profiler.hpp"

#include <cudf/copying.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/partitioning.hpp>
#include <cudf/utilities/error.hpp>

#include <algorithm>
#include <mutex>
#include <numeric>

namespace cudf {
namespace detail {
namespace {

constexpr int PART_BLOCK = 1024;
constexpr int PART_RPT   = 4;                      // rows per thread per tile
constexpr int PART_TILE  = PART_BLOCK * PART_RPT;  // 4096 rows staged per tile
constexpr int MAX_LDS_PARTITIONS = 4096;

struct hp_args {
  device_table hashed;   // columns that define the partition
  device_table all;      // columns to move
  void* out_data[MAX_COLS];
  int32_t nparts;
  uint32_t seed;
  int32_t pow2;
  int32_t slices;
  int64_t nrows;
  uint32_t* counts;      // [slices][P]
  int64_t* item_base;    // [slices][P]
  int64_t* offsets;      // [P + 1]
  size_type* order;      // gather map (out row -> source row) or nullptr
};

template <typename T>
__global__ void k_store_args(T v, T* dst)
{
  *dst = v;
}

__device__ __forceinline__ uint32_t partition_of(hp_args const& a, int64_t row)
{
  uint32_t const h = row_hash(a.hashed, row, a.seed);
  return a.pow2 ? (h & static_cast<uint32_t>(a.nparts - 1)) : (h % static_cast<uint32_t>(a.nparts));
}

__device__ __forceinline__ void slice_bounds(hp_args const& a, int item, int64_t& begin, int64_t& end)
{
  int64_t per = (a.nrows + a.slices - 1) / a.slices;
  per         = (per + PART_TILE - 1) / PART_TILE * PART_TILE;  // whole tiles per slice
  begin       = min(a.nrows, per * item);
  end         = min(a.nrows, begin + per);
}

__global__ void __launch_bounds__(PART_BLOCK) k_hp_hist(hp_args const* __restrict__ ap)
{
  extern __shared__ uint32_t hist[];
  hp_args const& a = *ap;
  int const P      = a.nparts;
  for (int d = threadIdx.x; d < P; d += blockDim.x) hist[d] = 0;
  __syncthreads();
  int64_t begin, end;
  slice_bounds(a, blockIdx.x, begin, end);
  for (int64_t r = begin + threadIdx.x; r < end; r += blockDim.x) atomicAdd(&hist[partition_of(a, r)], 1u);
  __syncthreads();
  for (int d = threadIdx.x; d < P; d += blockDim.x) a.counts[static_cast<int64_t>(blockIdx.x) * P + d] = hist[d];
}

// offsets[p] = rows in partitions < p; item_base[s][p] = offsets[p] + rows of p in slices < s. One block.
__global__ void __launch_bounds__(1024) k_hp_scan(hp_args const* __restrict__ ap)
{
  hp_args const& a = *ap;
  int const P = a.nparts, S = a.slices;
  extern __shared__ int64_t tot[];
  for (int d = threadIdx.x; d < P; d += blockDim.x) {
    int64_t t = 0;
    for (int s = 0; s < S; ++s) t += a.counts[static_cast<int64_t>(s) * P + d];
    tot[d] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int d = 0; d < P; ++d) {
      int64_t const t = tot[d];
      tot[d]          = run;
      a.offsets[d]    = run;
      run += t;
    }
    a.offsets[P] = run;
  }
  __syncthreads();
  for (int d = threadIdx.x; d < P; d += blockDim.x) {
    int64_t run = tot[d];
    for (int s = 0; s < S; ++s) {
      a.item_base[static_cast<int64_t>(s) * P + d] = run;
      run += a.counts[static_cast<int64_t>(s) * P + d];
    }
  }
}

// LDS: stage[TILE] u64 | pid[TILE] u16 | hist[P] u32 | delta[P] i64 | cursor[P] i64
__global__ void __launch_bounds__(PART_BLOCK) k_hp_scatter(hp_args const* __restrict__ ap)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  hp_args const& a = *ap;
  int const P      = a.nparts;
  uint64_t* stage  = reinterpret_cast<uint64_t*>(lds_raw);
  int64_t* delta   = reinterpret_cast<int64_t*>(stage + PART_TILE);
  int64_t* cursor  = delta + P;
  uint32_t* hist   = reinterpret_cast<uint32_t*>(cursor + P);
  uint32_t* loff   = hist + P;
  uint16_t* pid    = reinterpret_cast<uint16_t*>(loff + P);
  __shared__ uint32_t s_tile_count;

  int64_t begin, end;
  slice_bounds(a, blockIdx.x, begin, end);
  for (int d = threadIdx.x; d < P; d += blockDim.x) {
    cursor[d] = a.item_base[static_cast<int64_t>(blockIdx.x) * P + d];
    hist[d]   = 0;
  }
  __syncthreads();
  for (int64_t tile = begin; tile < end; tile += PART_TILE) {
    uint32_t dig[PART_RPT], rank[PART_RPT];
    bool live[PART_RPT];
#pragma unroll
    for (int k = 0; k < PART_RPT; ++k) {
      int64_t const r = tile + static_cast<int64_t>(k) * PART_BLOCK + threadIdx.x;
      live[k]         = r < end;
      if (live[k]) {
        dig[k]  = partition_of(a, r);
        rank[k] = atomicAdd(&hist[dig[k]], 1u);
      }
    }
    __syncthreads();
    // exclusive scan of hist (P <= 4096): thread 0..P-1 style serial-by-wave scan kept simple: one wave scans
    if (threadIdx.x < 64) {
      int const lane = threadIdx.x;
      uint32_t carry = 0;
      for (int base = 0; base < P; base += 64) {
        int const d      = base + lane;
        uint32_t const v = d < P ? hist[d] : 0;
        uint32_t inc     = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          uint32_t const t = __shfl_up(inc, o);
          if (lane >= o) inc += t;
        }
        if (d < P) {
          uint32_t const excl = carry + inc - v;
          loff[d]             = excl;
          delta[d]            = cursor[d] - static_cast<int64_t>(excl);
          cursor[d] += v;
          hist[d] = 0;
        }
        carry += __shfl(inc, 63);
      }
      if (lane == 0) s_tile_count = carry;
    }
    __syncthreads();
    uint32_t pos[PART_RPT];
#pragma unroll
    for (int k = 0; k < PART_RPT; ++k) {
      if (live[k]) {
        pos[k]      = loff[dig[k]] + rank[k];
        pid[pos[k]] = static_cast<uint16_t>(dig[k]);
      }
    }
    __syncthreads();
    uint32_t const tile_count = s_tile_count;
    // gather map: out row -> source row
    if (a.order != nullptr) {
#pragma unroll
      for (int k = 0; k < PART_RPT; ++k)
        if (live[k]) reinterpret_cast<uint32_t*>(stage)[pos[k]] = static_cast<uint32_t>(tile + static_cast<int64_t>(k) * PART_BLOCK + threadIdx.x);
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < tile_count; j += PART_BLOCK)
        gstore(a.order + (delta[pid[j]] + j), static_cast<size_type>(reinterpret_cast<uint32_t*>(stage)[j]));
      __syncthreads();
    }
    // every column goes through the same staging tile
    for (int c = 0; c < a.all.ncols; ++c) {
      device_column const col = a.all.col[c];
#pragma unroll
      for (int k = 0; k < PART_RPT; ++k)
        if (live[k]) stage[pos[k]] = col_load_bits(col, tile + static_cast<int64_t>(k) * PART_BLOCK + threadIdx.x);
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < tile_count; j += PART_BLOCK) {
        int64_t const dst = delta[pid[j]] + j;
        uint64_t const v  = stage[j];
        switch (col.width) {
          case 1: gstore(static_cast<uint8_t*>(a.out_data[c]) + dst, static_cast<uint8_t>(v)); break;
          case 2: gstore(static_cast<uint16_t*>(a.out_data[c]) + dst, static_cast<uint16_t>(v)); break;
          case 4: gstore(static_cast<uint32_t*>(a.out_data[c]) + dst, static_cast<uint32_t>(v)); break;
          default: gstore(static_cast<uint64_t*>(a.out_data[c]) + dst, v);
        }
      }
      __syncthreads();
    }
  }
}

// out_mask bit i = validity of source row map[i] (NULLIFY: out-of-range -> null); counts nulls.
__global__ void __launch_bounds__(256) k_gather_mask(device_column src, int32_t src_rows, size_type const* map, int64_t n,
                                                     bitmask_type* out_mask, int32_t* null_count, int32_t nullify)
{
  int64_t const i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  bool valid      = false;
  if (i < n) {
    int64_t s = map[i];
    if (s < 0) s += src_rows;  // negative indices count from the end (reference copying/gather.cu:81-82)
    bool const in   = s >= 0 && s < src_rows;
    valid           = in ? col_is_valid(src, s) : !nullify;
  }
  unsigned long long const ballot = __ballot(valid);
  unsigned long long const live   = __ballot(i < n);
  int const lane                  = threadIdx.x & 63;
  if (i < n && (lane & 31) == 0) out_mask[i >> 5] = static_cast<uint32_t>(ballot >> (lane & 32));
  int const nulls = __popcll(live & ~ballot);
  if (lane == 0 && nulls) atomicAdd(null_count, nulls);
}

__global__ void __launch_bounds__(256) k_gather_data(device_column src, int32_t src_rows, size_type const* map, int64_t n,
                                                     void* out)
{
  int64_t const i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  int64_t s = map[i];
  if (s < 0) s += src_rows;
  uint64_t const v = (s >= 0 && s < src_rows) ? col_load_bits(src, s) : 0;
  switch (src.width) {
    case 1: static_cast<uint8_t*>(out)[i] = static_cast<uint8_t>(v); break;
    case 2: static_cast<uint16_t*>(out)[i] = static_cast<uint16_t>(v); break;
    case 4: static_cast<uint32_t*>(out)[i] = static_cast<uint32_t>(v); break;
    default: static_cast<uint64_t*>(out)[i] = v;
  }
}

// Gather of up to GATHER_COLS columns that share one map: R rows per thread, the map elements first, then every
// column's R random loads back to back (the random reads are what the pass waits for: bytes in flight buy throughput,
// and the map is read once instead of once per column), then the stores.
constexpr int GATHER_COLS = 4;
struct gather_cols {
  device_column src[GATHER_COLS];
  void* out[GATHER_COLS];
  int32_t ncols;
  int32_t src_rows;
};
__global__ void __launch_bounds__(256) k_gather_multi(gather_cols g, size_type const* map, int64_t n)
{
  constexpr int R = 4;
  int64_t const base = (blockIdx.x * static_cast<int64_t>(blockDim.x)) * R + threadIdx.x;
  int64_t s[R];
  bool in[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    int64_t const i = base + static_cast<int64_t>(k) * blockDim.x;
    s[k]            = i < n ? gload(map + i) : INT32_MIN;
    if (s[k] < 0) s[k] += g.src_rows;
    in[k]           = s[k] >= 0 && s[k] < g.src_rows;
  }
#pragma unroll
  for (int c = 0; c < GATHER_COLS; ++c) {
    if (c >= g.ncols) break;
    device_column const col = g.src[c];
    uint64_t v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = in[k] ? col_load_bits(col, s[k]) : 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      int64_t const i = base + static_cast<int64_t>(k) * blockDim.x;
      if (i >= n) continue;
      switch (col.width) {
        case 1: gstore(static_cast<uint8_t*>(g.out[c]) + i, static_cast<uint8_t>(v[k])); break;
        case 2: gstore(static_cast<uint16_t*>(g.out[c]) + i, static_cast<uint16_t>(v[k])); break;
        case 4: gstore(static_cast<uint32_t*>(g.out[c]) + i, static_cast<uint32_t>(v[k])); break;
        default: gstore(static_cast<uint64_t*>(g.out[c]) + i, v[k]);
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_row_hash(device_table t, uint32_t seed, uint32_t* out)
{
  int64_t const i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < t.nrows) out[i] = row_hash(t, i, seed);
}

std::pair<rmm::device_buffer, size_type> gather_mask(column_view const& src, size_type const* map, int64_t n, bool nullify,
                                                     stream_ref stream, rmm::device_async_resource_ref mr)
{
  auto mask = create_null_mask(static_cast<size_type>(n), mask_state::UNINITIALIZED, stream, mr);
  if (n == 0) return {std::move(mask), 0};
  rmm::device_buffer counter{sizeof(int32_t), stream.value(), cudf::get_current_device_resource_ref()};
  CUDF_HIP_TRY(hipMemsetAsync(counter.data(), 0, sizeof(int32_t), stream.value()));
  auto dc = make_device_column(src);
  if (!src.has_nulls()) dc.mask = nullptr;
  hipLaunchKernelGGL(k_gather_mask, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream.value(), dc, src.size(),
                     map, n, static_cast<bitmask_type*>(mask.data()), static_cast<int32_t*>(counter.data()), nullify ? 1 : 0);
  CUDF_HIP_TRY(hipGetLastError());
  int32_t h = 0;
  CUDF_HIP_TRY(hipMemcpyAsync(&h, counter.data(), sizeof(int32_t), hipMemcpyDeviceToHost, stream.value()));
  CUDF_HIP_TRY(hipStreamSynchronize(stream.value()));
  return {std::move(mask), h};
}
}  // namespace
}  // namespace detail

std::pair<std::unique_ptr<table>, std::vector<size_type>> hash_partition(table_view const& input,
                                                                         std::vector<size_type> const& columns_to_hash,
                                                                         int num_partitions, hash_id hash_function, uint32_t seed,
                                                                         stream_ref stream, rmm::device_async_resource_ref mr)
{
  // (an index outside the table: std::out_of_range from select, as in the reference - hash_partition_test.cpp:48-58)
  return hash_partition(input, input.select(columns_to_hash), num_partitions, hash_function, seed, stream, mr);
}

std::pair<std::unique_ptr<table>, std::vector<size_type>> hash_partition(table_view const& input, table_view const& table_to_hash,
                                                                         int num_partitions, hash_id hash_function, uint32_t seed,
                                                                         stream_ref stream, rmm::device_async_resource_ref mr)
{
  using namespace detail;
  // reference partitioning.cu:925-947
  CUDF_EXPECTS(table_to_hash.num_columns() == 0 || input.num_rows() == table_to_hash.num_rows(),
               "Input table and key table must have same number of rows, or key table should have no columns.", std::invalid_argument);
  CUDF_EXPECTS(hash_function == hash_id::HASH_MURMUR3, "Only HASH_MURMUR3 is implemented on this path.");
  // Return empty result if there are no partitions or nothing to hash; the offsets vector always has num_partitions + 1
  // entries (reference partitioning.cu:880-888)
  if (num_partitions <= 0 || input.num_rows() == 0 || table_to_hash.num_columns() == 0) {
    return {empty_like(input), std::vector<size_type>(static_cast<std::size_t>(std::max(num_partitions, 0)) + 1, 0)};
  }
  CUDF_EXPECTS(num_partitions <= MAX_LDS_PARTITIONS, "hash_partition: more than 4096 partitions are not implemented.");
  hipStream_t const s = stream.value();
  auto tmp            = cudf::get_current_device_resource_ref();
  int64_t const n     = input.num_rows();

  hp_args a{};
  a.hashed = make_device_table(table_to_hash);
  a.all    = make_device_table(input);
  a.nparts = num_partitions;
  a.seed   = seed;
  a.pow2   = (num_partitions & (num_partitions - 1)) == 0;
  a.nrows  = n;
  a.slices = static_cast<int32_t>(std::clamp<int64_t>((n + PART_TILE - 1) / PART_TILE, 1, 512));
  bool any_nullable = false;
  for (auto const& c : input) any_nullable = any_nullable || c.nullable();

  std::vector<std::unique_ptr<column>> out_cols;
  for (int c = 0; c < input.num_columns(); ++c) {
    auto const& col = input.column(c);
    out_cols.push_back(std::make_unique<column>(col.type(), static_cast<size_type>(n),
                                                rmm::device_buffer{static_cast<size_t>(n) * size_of(col.type()), s, mr},
                                                rmm::device_buffer{}, 0));
    a.out_data[c] = out_cols.back()->mutable_view().head();
  }
  size_t const P = static_cast<size_t>(num_partitions);
  rmm::device_buffer counts{static_cast<size_t>(a.slices) * P * sizeof(uint32_t), s, tmp};
  rmm::device_buffer item_base{static_cast<size_t>(a.slices) * P * sizeof(int64_t), s, tmp};
  rmm::device_buffer offsets{(P + 1) * sizeof(int64_t), s, tmp};
  rmm::device_buffer order{};
  if (any_nullable) order = rmm::device_buffer{static_cast<size_t>(n) * sizeof(size_type), s, tmp};
  a.counts    = static_cast<uint32_t*>(counts.data());
  a.item_base = static_cast<int64_t*>(item_base.data());
  a.offsets   = static_cast<int64_t*>(offsets.data());
  a.order     = any_nullable ? static_cast<size_type*>(order.data()) : nullptr;
  rmm::device_buffer d_args{sizeof(hp_args), s, tmp};
  auto* da = static_cast<hp_args*>(d_args.data());
  hipLaunchKernelGGL(k_store_args<hp_args>, dim3(1), dim3(1), 0, s, a, da);
  {
    prof::scope p_{"hash_partition_hist", s};
    hipLaunchKernelGGL(k_hp_hist, dim3(a.slices), dim3(PART_BLOCK), P * sizeof(uint32_t), s, da);
  }
  hipLaunchKernelGGL(k_hp_scan, dim3(1), dim3(1024), P * sizeof(int64_t), s, da);
  {
    size_t const lds = PART_TILE * 8 + P * (8 + 8 + 4 + 4) + PART_TILE * 2;
    // beyond 1024 partitions the tile needs more than the default 64 KiB of dynamic LDS (139 KiB at 4096): opt in once
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
      hipFuncAttributes attr{};
      CUDF_HIP_TRY(hipFuncGetAttributes(&attr, reinterpret_cast<void const*>(&k_hp_scatter)));
      CUDF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<void const*>(&k_hp_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024 - static_cast<int>(attr.sharedSizeBytes)));
    });
    prof::scope p_{"hash_partition_scatter", s};
    hipLaunchKernelGGL(k_hp_scatter, dim3(a.slices), dim3(PART_BLOCK), lds, s, da);
    CUDF_HIP_TRY(hipGetLastError());
  }
  std::vector<int64_t> h_off(P + 1);
  CUDF_HIP_TRY(hipMemcpyAsync(h_off.data(), a.offsets, (P + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, s));
  CUDF_HIP_TRY(hipStreamSynchronize(s));
  // validity of the moved rows
  for (int c = 0; c < input.num_columns(); ++c) {
    auto const& col = input.column(c);
    if (!col.nullable()) continue;
    auto [mask, nulls] = gather_mask(col, a.order, n, false, stream, mr);
    out_cols[c]->set_null_mask(std::move(mask), nulls);
  }
  // num_partitions + 1 offsets: partition i = rows [offsets[i], offsets[i + 1]); the last one is the row count
  // (reference partitioning.hpp:84-101, partitioning.cu:688-690)
  std::vector<size_type> starts(P + 1);
  for (size_t p = 0; p <= P; ++p) starts[p] = static_cast<size_type>(h_off[p]);
  return {std::make_unique<table>(std::move(out_cols)), std::move(starts)};
}

namespace hashing {
std::unique_ptr<column> murmurhash3_x86_32(table_view const& input, uint32_t seed, stream_ref stream,
                                           rmm::device_async_resource_ref mr)
{
  auto out = make_fixed_width_column(data_type{type_id::UINT32}, input.num_rows(), mask_state::UNALLOCATED, stream, mr);
  if (input.num_rows() == 0 || input.num_columns() == 0) return out;
  auto const t = detail::make_device_table(input);
  hipLaunchKernelGGL(detail::k_row_hash, dim3(static_cast<unsigned>((input.num_rows() + 255) / 256)), dim3(256), 0, stream.value(),
                     t, seed, out->mutable_view().data<uint32_t>());
  CUDF_HIP_TRY(hipGetLastError());
  return out;
}
}  // namespace hashing

std::unique_ptr<table> gather(table_view const& source_table, column_view const& gather_map, out_of_bounds_policy bounds_policy,
                              stream_ref stream, rmm::device_async_resource_ref mr)
{
  using namespace detail;
  CUDF_EXPECTS(!gather_map.has_nulls(), "gather_map contains nulls", std::invalid_argument);
  CUDF_EXPECTS(gather_map.type().id() == type_id::INT32, "gather_map must be an INT32 column on this path.");
  int64_t const n = gather_map.size();
  auto const* map = gather_map.data<size_type>();
  bool const nullify = bounds_policy == out_of_bounds_policy::NULLIFY;
  std::vector<std::unique_ptr<column>> out_cols;
  for (auto const& col : source_table) {
    auto const w = size_of(col.type());
    CUDF_EXPECTS(w <= 8, "gather: only fixed-width columns of at most 8 bytes.");
    out_cols.push_back(std::make_unique<column>(col.type(), static_cast<size_type>(n),
                                                rmm::device_buffer{static_cast<size_t>(n) * w, stream.value(), mr}, rmm::device_buffer{}, 0));
  }
  if (n > 0) {
    // data: columns in groups of GATHER_COLS per launch
    for (size_type first = 0; first < source_table.num_columns(); first += GATHER_COLS) {
      gather_cols g{};
      g.ncols    = std::min<int32_t>(GATHER_COLS, source_table.num_columns() - first);
      g.src_rows = source_table.num_rows();
      for (int c = 0; c < g.ncols; ++c) {
        g.src[c] = make_device_column(source_table.column(first + c));
        g.out[c] = out_cols[static_cast<size_t>(first + c)]->mutable_view().head();
      }
      cudf::detail::prof::scope p_{"gather", stream.value()};
      hipLaunchKernelGGL(k_gather_multi, dim3(static_cast<unsigned>((n + 1023) / 1024)), dim3(256), 0, stream.value(), g, map, n);
      CUDF_HIP_TRY(hipGetLastError());
    }
    // validity
    for (size_type c = 0; c < source_table.num_columns(); ++c) {
      auto const& col = source_table.column(c);
      if (col.nullable() || nullify) {
        auto [mask, nulls] = gather_mask(col, map, n, nullify, stream, mr);
        if (col.nullable() || nulls > 0) out_cols[static_cast<size_t>(c)]->set_null_mask(std::move(mask), nulls);
      }
    }
  }
  return std::make_unique<table>(std::move(out_cols));
}
}  // namespace cudf
