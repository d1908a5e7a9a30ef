// SPDX-License-Identifier: Apache-2.0
// The sort-based groupby: what the reference falls back to when a request holds a kind the hash tables cannot serve
// (cpp/src/groupby/groupby.cu:64-69 -> cpp/src/groupby/sort/aggregate.cpp, sort_helper.cu). Built here from one device primitive,
// a stable least-significant-digit radix sort of (64-bit word, 32-bit payload) pairs in 8-bit digits:
//   keys   -> order-preserving words, one key column at a time from the last to the first (nulls after, as sort_helper.cu:93-111)
//   order  -> group boundaries (adjacent rows unequal) -> labels and offsets (sort_helper.cu:121-160)
//   values -> a second sort by (label, null, value) for the kinds that read order statistics (sort_helper.cu:205-225)
// Digits in which all words agree are skipped (one OR/AND reduction per word column tells), so a 32-bit key column of a few
// million distinct values costs three passes, not eight. The kinds the hash engine serves are handed to it with the group label
// as the key and come back aligned to the sorted unique keys.
#include "call.hpp"
#include "../common/profiler.hpp"

#include <cudf/copying.hpp>
#include <cudf/groupby.hpp>
#include <cudf/null_mask.hpp>
#include <cudf/utilities/error.hpp>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>

namespace cudf::groupby::detail {
namespace {
using cudf::detail::CLS_BOOL;
using cudf::detail::CLS_F32;
using cudf::detail::CLS_F64;
using cudf::detail::CLS_SINT;
using cudf::detail::CLS_UINT;
using cudf::detail::col_is_valid;
using cudf::detail::col_load_bits;
using cudf::detail::device_column;
using cudf::detail::device_table;
using cudf::detail::gload;
using cudf::detail::gstore;
namespace prof = cudf::detail::prof;

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES   = RS_THREADS / 64;
constexpr int RS_ITEMS   = 16;                     // keys per thread
constexpr int RS_TILE    = RS_THREADS * RS_ITEMS;  // keys per workgroup

// ---------------------------------------------------------------------------------------------------------
// order-preserving 64-bit word of one element: unsigned comparison of words == the reference's ascending order of values
// (NaN above every number and all NaNs one value, -0 == +0: cpp/include/cudf/detail/row_operator/lexicographic.cuh)
__device__ __forceinline__ uint64_t sortable_word(device_column const& c, int64_t row)
{
  uint64_t const raw = col_load_bits(c, row);
  switch (c.cls) {
    case CLS_SINT: {
      int const sh = 64 - 8 * c.width;
      return static_cast<uint64_t>(static_cast<int64_t>(raw << sh) >> sh) ^ 0x8000000000000000ull;
    }
    case CLS_BOOL: return raw != 0;
    case CLS_F32: {
      uint32_t b = static_cast<uint32_t>(cudf::detail::normalize_key_bits(raw, CLS_F32));
      return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
    }
    case CLS_F64: {
      uint64_t b = cudf::detail::normalize_key_bits(raw, CLS_F64);
      return b ^ ((b >> 63) ? 0xffffffffffffffffull : 0x8000000000000000ull);
    }
    default: return raw;
  }
}

__device__ __forceinline__ uint64_t wave_or(uint64_t v)
{
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint64_t wave_and(uint64_t v)
{
  for (int o = 32; o > 0; o >>= 1) v &= __shfl_xor(v, o, 64);
  return v;
}

// what to make a word of
enum word_source : int32_t {
  WORD_VALUE      = 0,  // sortable_word(col[row]); 0 for a null
  WORD_NULL_FLAG  = 1,  // 1 for a null of col
  WORD_ANY_NULL   = 2,  // 1 if any column of the table holds a null in the row
  WORD_LABEL_NULL = 3   // (label[payload] << 1) | null flag of col[row]
};

struct fill_args {
  device_table keys;         // WORD_ANY_NULL
  device_column col;         // the others
  uint32_t const* payload;   // nullptr: identity
  uint32_t const* order;     // WORD_LABEL_NULL: row = order[payload]; otherwise row = payload
  int32_t const* labels;     // WORD_LABEL_NULL
  int64_t n;
  uint64_t* words;
  uint64_t* or_and;          // [0] |= word, [1] &= word
  uint32_t* ones;            // WORD_ANY_NULL: number of words equal to 1
  int32_t source;
};

__global__ void __launch_bounds__(256) k_fill_words(fill_args a)
{
  uint64_t vor = 0, vand = ~0ull;
  uint32_t ones = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < a.n; i += static_cast<int64_t>(gridDim.x) * 256) {
    uint32_t const p = a.payload ? gload(a.payload + i) : static_cast<uint32_t>(i);
    uint64_t w;
    if (a.source == WORD_VALUE) {
      w = col_is_valid(a.col, p) ? sortable_word(a.col, p) : 0;
    } else if (a.source == WORD_NULL_FLAG) {
      w = col_is_valid(a.col, p) ? 0 : 1;
    } else if (a.source == WORD_ANY_NULL) {
      w = cudf::detail::row_has_null(a.keys, p) ? 1 : 0;
      ones += static_cast<uint32_t>(w);
    } else {
      uint32_t const row = gload(a.order + p);
      w = (static_cast<uint64_t>(static_cast<uint32_t>(gload(a.labels + p))) << 1) | (col_is_valid(a.col, row) ? 0 : 1);
    }
    gstore(a.words + i, w);
    vor |= w;
    vand &= w;
  }
  vor  = wave_or(vor);
  vand = wave_and(vand);
  if ((threadIdx.x & 63) == 0) {
    atomicOr(reinterpret_cast<unsigned long long*>(a.or_and), static_cast<unsigned long long>(vor));
    atomicAnd(reinterpret_cast<unsigned long long*>(a.or_and + 1), static_cast<unsigned long long>(vand));
  }
  if (a.ones != nullptr) {
    for (int o = 32; o > 0; o >>= 1) ones += __shfl_xor(ones, o, 64);
    if ((threadIdx.x & 63) == 0 && ones) atomicAdd(a.ones, ones);
  }
}

// WORD_VALUE through order: word of values[order[payload]] (the second sort)
__global__ void __launch_bounds__(256) k_fill_value_words(device_column col, uint32_t const* order, int64_t n, uint64_t* words,
                                                          uint64_t* or_and)
{
  uint64_t vor = 0, vand = ~0ull;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) {
    uint32_t const row = gload(order + i);
    uint64_t const w   = col_is_valid(col, row) ? sortable_word(col, row) : 0;
    gstore(words + i, w);
    vor |= w;
    vand &= w;
  }
  vor  = wave_or(vor);
  vand = wave_and(vand);
  if ((threadIdx.x & 63) == 0) {
    atomicOr(reinterpret_cast<unsigned long long*>(or_and), static_cast<unsigned long long>(vor));
    atomicAnd(reinterpret_cast<unsigned long long*>(or_and + 1), static_cast<unsigned long long>(vand));
  }
}

__global__ void __launch_bounds__(256) k_iota(uint32_t* out, int64_t n)
{
  int64_t const i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) gstore(out + i, static_cast<uint32_t>(i));
}

__global__ void k_init_or_and(uint64_t* or_and, uint32_t* ones)
{
  or_and[0] = 0;
  or_and[1] = ~0ull;
  if (ones) *ones = 0;
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t const t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// exclusive prefix of `v` over the workgroup's threads; `total` = the workgroup's sum
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t& total, uint32_t* wave_sums /* THREADS / 64 + 1 */)
{
  int const lane     = threadIdx.x & 63;
  int const w        = threadIdx.x >> 6;
  uint32_t const inc = wave_inclusive_scan(v, lane);
  __syncthreads();  // wave_sums may still be read from the previous use
  if (lane == 63) wave_sums[w] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int i = 0; i < THREADS / 64; ++i) {
    uint32_t const s = wave_sums[i];
    if (i < w) before += s;
    all += s;
  }
  total = all;
  return before + inc - v;
}

// ---------------------------------------------------------------------------------------------------------
// radix pass. Lanes of one wave that hold the same digit find each other with eight ballots (one per digit bit); the lowest such lane
// updates the wave's counter for the digit, so no two lanes of a wave ever touch one counter and a run of equal keys costs what a
// random one does.
__device__ __forceinline__ uint64_t digit_peers(uint32_t d, bool active)
{
  uint64_t m = __ballot(active);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    bool const bit    = (d >> b) & 1u;
    uint64_t const bb = __ballot(bit);
    m &= bit ? bb : ~bb;
  }
  return m;
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t m, int lane) { return __popcll(m & ((1ull << lane) - 1ull)); }

// hist[d * nblocks + b] = number of keys of workgroup b's tile whose digit is d
__global__ void __launch_bounds__(RS_THREADS) k_radix_hist(uint64_t const* keys, int64_t n, int shift, uint32_t* hist, uint32_t nblocks)
{
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  int const lane     = threadIdx.x & 63;
  int64_t const base = static_cast<int64_t>(blockIdx.x) * RS_TILE + (threadIdx.x >> 6) * (64 * RS_ITEMS);
#pragma unroll 4
  for (int r = 0; r < RS_ITEMS; ++r) {
    int64_t const i    = base + r * 64 + lane;
    bool const active  = i < n;
    uint32_t const d   = active ? static_cast<uint32_t>(gload(keys + i) >> shift) & 255u : 0u;
    uint64_t const m   = digit_peers(d, active);
    if (active && lanes_below(m, lane) == 0) atomicAdd(&h[d], static_cast<uint32_t>(__popcll(m)));
  }
  __syncthreads();
  hist[static_cast<size_t>(threadIdx.x) * nblocks + blockIdx.x] = h[threadIdx.x];
}

// stable scatter of one tile: element order inside the tile is (wave, round, lane). The tile is first put in digit order inside
// LDS, then written out: consecutive lanes then write consecutive addresses inside each digit's run (16 elements on average)
// instead of 64 lanes writing to 64 runs.
__global__ void __launch_bounds__(RS_THREADS) k_radix_scatter(uint64_t const* kin, uint32_t const* pin, uint64_t* kout, uint32_t* pout,
                                                              int64_t n, int shift, uint32_t const* bases, uint32_t nblocks)
{
  __shared__ uint32_t cnt[RS_WAVES][256];
  __shared__ uint32_t delta[256];  // global position of a digit's run minus its position inside the tile
  __shared__ uint32_t ws[RS_THREADS / 64 + 1];
  __shared__ uint64_t skey[RS_TILE];
  __shared__ uint32_t spay[RS_TILE];
  for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();
  int const lane          = threadIdx.x & 63;
  int const w             = threadIdx.x >> 6;
  int64_t const tile_base = static_cast<int64_t>(blockIdx.x) * RS_TILE;
  int64_t const base      = tile_base + w * (64 * RS_ITEMS);
  uint64_t key[RS_ITEMS];
  uint32_t pay[RS_ITEMS];
  uint32_t loc[RS_ITEMS];  // digit << 16 | rank among the wave's keys of that digit
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    int64_t const i = base + r * 64 + lane;
    key[r]          = i < n ? gload(kin + i) : 0;
    pay[r]          = i < n ? (pin ? gload(pin + i) : static_cast<uint32_t>(i)) : 0;
  }
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    bool const active   = base + r * 64 + lane < n;
    uint32_t const d    = static_cast<uint32_t>(key[r] >> shift) & 255u;
    uint64_t const m    = digit_peers(d, active);
    uint32_t const rank = lanes_below(m, lane);
    uint32_t const seen = cnt[w][d];
    __builtin_amdgcn_wave_barrier();
    if (active && rank == 0) cnt[w][d] = seen + static_cast<uint32_t>(__popcll(m));
    __builtin_amdgcn_wave_barrier();
    loc[r] = (d << 16) | (seen + rank);
  }
  __syncthreads();
  {
    uint32_t const d = threadIdx.x;
    uint32_t c[RS_WAVES];
    uint32_t all = 0;
#pragma unroll
    for (int ww = 0; ww < RS_WAVES; ++ww) {
      c[ww] = cnt[ww][d];
      all += c[ww];
    }
    uint32_t total;
    uint32_t start = block_exclusive_scan<RS_THREADS>(all, total, ws);  // the digit's run inside the tile
    delta[d]       = gload(bases + static_cast<size_t>(d) * nblocks + blockIdx.x) - start;
#pragma unroll
    for (int ww = 0; ww < RS_WAVES; ++ww) {
      cnt[ww][d] = start;
      start += c[ww];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    if (base + r * 64 + lane < n) {
      uint32_t const at = cnt[w][loc[r] >> 16] + (loc[r] & 0xffffu);
      skey[at]          = key[r];
      spay[at]          = pay[r];
    }
  }
  __syncthreads();
  int const live = static_cast<int>(n - tile_base < RS_TILE ? n - tile_base : RS_TILE);
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    int const at = r * RS_THREADS + threadIdx.x;
    if (at < live) {
      uint64_t const k   = skey[at];
      uint32_t const pos = delta[static_cast<uint32_t>(k >> shift) & 255u] + static_cast<uint32_t>(at);
      gstore(kout + pos, k);
      gstore(pout + pos, spay[at]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// exclusive scan of uint32 (counts of at most 2^31 in total): tile sums -> scan of the tile sums -> tiles again with their base
constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS   = 16;
constexpr int SC_TILE    = SC_THREADS * SC_ITEMS;

__global__ void __launch_bounds__(SC_THREADS) k_scan_tile_sums(uint32_t const* in, int64_t n, uint32_t* tile_sums)
{
  __shared__ uint32_t ws[SC_THREADS / 64 + 1];
  int64_t const base = static_cast<int64_t>(blockIdx.x) * SC_TILE;
  uint32_t v         = 0;
#pragma unroll
  for (int r = 0; r < SC_ITEMS; ++r) {
    int64_t const i = base + r * SC_THREADS + threadIdx.x;
    if (i < n) v += gload(in + i);
  }
  uint32_t total;
  block_exclusive_scan<SC_THREADS>(v, total, ws);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// in place, one workgroup; total (optional) receives the sum of everything
__global__ void __launch_bounds__(1024) k_scan_spine(uint32_t* sums, int64_t count, uint32_t* total_out)
{
  __shared__ uint32_t ws[1024 / 64 + 1];
  uint32_t carry = 0;
  for (int64_t first = 0; first < count; first += 1024) {
    int64_t const i  = first + threadIdx.x;
    uint32_t const v = i < count ? sums[i] : 0;
    uint32_t total;
    uint32_t const ex = block_exclusive_scan<1024>(v, total, ws);
    if (i < count) sums[i] = carry + ex;
    carry += total;
  }
  if (total_out != nullptr && threadIdx.x == 0) *total_out = carry;
}

__global__ void __launch_bounds__(SC_THREADS) k_scan_apply(uint32_t const* in, uint32_t* out, int64_t n, uint32_t const* tile_bases)
{
  __shared__ uint32_t ws[SC_THREADS / 64 + 1];
  int64_t const first = static_cast<int64_t>(blockIdx.x) * SC_TILE + static_cast<int64_t>(threadIdx.x) * SC_ITEMS;
  uint32_t v[SC_ITEMS];
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < SC_ITEMS; ++k) {
    v[k] = first + k < n ? gload(in + first + k) : 0;
    sum += v[k];
  }
  uint32_t total;
  uint32_t run = gload(tile_bases + blockIdx.x) + block_exclusive_scan<SC_THREADS>(sum, total, ws);
#pragma unroll
  for (int k = 0; k < SC_ITEMS; ++k) {
    if (first + k < n) gstore(out + first + k, run);
    run += v[k];
  }
}

// out[0 .. n) = exclusive prefix sums of in[0 .. n); *total (device, optional) = the sum. in == out is allowed.
void exclusive_scan(uint32_t const* in, uint32_t* out, int64_t n, uint32_t* total, scratch& sc, hipStream_t s)
{
  if (n <= 0) {
    if (total) CUDF_HIP_TRY(hipMemsetAsync(total, 0, sizeof(uint32_t), s));
    return;
  }
  int64_t const tiles = (n + SC_TILE - 1) / SC_TILE;
  auto* sums          = sc.alloc<uint32_t>(static_cast<std::size_t>(tiles));
  prof::scope p_{"sort_scan", s};
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(static_cast<unsigned>(tiles)), dim3(SC_THREADS), 0, s, in, n, sums);
  hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, s, sums, tiles, total);
  hipLaunchKernelGGL(k_scan_apply, dim3(static_cast<unsigned>(tiles)), dim3(SC_THREADS), 0, s, in, out, n, sums);
  CUDF_HIP_TRY(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------
// (word, payload) pairs in two buffers each; sort_by() runs the passes one word column needs
class pair_sorter {
 public:
  pair_sorter(int64_t n, scratch& sc, hipStream_t s) : _n{n}, _sc{sc}, _s{s}
  {
    _nblocks = static_cast<uint32_t>((n + RS_TILE - 1) / RS_TILE);
    for (int i = 0; i < 2; ++i) {
      _k[i] = sc.alloc<uint64_t>(static_cast<std::size_t>(n));
      _p[i] = sc.alloc<uint32_t>(static_cast<std::size_t>(n));
    }
    _hist   = sc.alloc<uint32_t>(static_cast<std::size_t>(_nblocks) * 256);
    _or_and = sc.alloc<uint64_t>(2);
    _ones   = sc.alloc<uint32_t>(1);
  }
  // the payload in front (nullptr before the first pass: identity)
  [[nodiscard]] uint32_t const* payload() const { return _identity ? nullptr : _p[_cur]; }
  [[nodiscard]] uint64_t* words() { return _k[_cur]; }
  [[nodiscard]] uint64_t* or_and() { return _or_and; }
  [[nodiscard]] uint32_t* ones() { return _ones; }
  void reset_reduction() { hipLaunchKernelGGL(k_init_or_and, dim3(1), dim3(1), 0, _s, _or_and, _ones); }
  // reads back the OR / AND of the words just filled (one synchronisation) and sorts by the digits that differ;
  // returns the count of WORD_ANY_NULL ones
  uint32_t sort_filled()
  {
    struct {
      uint64_t or_and[2];
      uint32_t ones;
    } h{};
    CUDF_HIP_TRY(hipMemcpyAsync(h.or_and, _or_and, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, _s));
    CUDF_HIP_TRY(hipMemcpyAsync(&h.ones, _ones, sizeof(uint32_t), hipMemcpyDeviceToHost, _s));
    CUDF_HIP_TRY(hipStreamSynchronize(_s));
    uint64_t const differ = h.or_and[0] ^ h.or_and[1];
    for (int b = 0; b < 8; ++b)
      if ((differ >> (8 * b)) & 0xffu) pass(8 * b);
    return h.ones;
  }
  // payload after the last pass, materialised (identity if nothing ever moved)
  uint32_t* finish()
  {
    if (_identity) {
      hipLaunchKernelGGL(k_iota, dim3(static_cast<unsigned>((_n + 255) / 256)), dim3(256), 0, _s, _p[_cur], _n);
      _identity = false;
    }
    return _p[_cur];
  }
  int passes{0};

 private:
  void pass(int shift)
  {
    {
      prof::scope p_{"sort_hist", _s};
      hipLaunchKernelGGL(k_radix_hist, dim3(_nblocks), dim3(RS_THREADS), 0, _s, _k[_cur], _n, shift, _hist, _nblocks);
    }
    exclusive_scan(_hist, _hist, static_cast<int64_t>(_nblocks) * 256, nullptr, _sc, _s);
    {
      prof::scope p_{"sort_scatter", _s};
      hipLaunchKernelGGL(k_radix_scatter, dim3(_nblocks), dim3(RS_THREADS), 0, _s, _k[_cur], payload(), _k[_cur ^ 1], _p[_cur ^ 1], _n, shift,
                         _hist, _nblocks);
    }
    CUDF_HIP_TRY(hipGetLastError());
    _cur ^= 1;
    _identity = false;
    ++passes;
  }
  int64_t _n;
  scratch& _sc;
  hipStream_t _s;
  uint32_t _nblocks{0};
  uint64_t* _k[2]{};
  uint32_t* _p[2]{};
  uint32_t* _hist{nullptr};
  uint64_t* _or_and{nullptr};
  uint32_t* _ones{nullptr};
  int _cur{0};
  bool _identity{true};
};

unsigned fill_grid(int64_t n) { return static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 256 * 16)); }

// ---------------------------------------------------------------------------------------------------------
// groups
__global__ void __launch_bounds__(256) k_boundaries(device_table keys, uint32_t const* order, int64_t nk, uint32_t* flags)
{
  int64_t const i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= nk) return;
  uint32_t f = 1;
  if (i > 0) f = cudf::detail::rows_equal(keys, gload(order + i - 1), keys, gload(order + i), true) ? 0u : 1u;
  gstore(flags + i, f);
}

// one key column without nulls: its sorted words are still in the sorter's front buffer
__global__ void __launch_bounds__(256) k_boundaries_of_words(uint64_t const* words, int64_t nk, uint32_t* flags)
{
  int64_t const i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < nk) gstore(flags + i, (i == 0 || gload(words + i) != gload(words + i - 1)) ? 1u : 0u);
}

// labels[i] = (number of boundaries at or before i) - 1; offsets[label] = i at a boundary; first_row[label] = order[i]
__global__ void __launch_bounds__(256) k_labels(uint32_t const* flags, uint32_t const* before, uint32_t const* order, int64_t nk,
                                                int32_t* labels, int32_t* offsets, int32_t* first_row)
{
  int64_t const i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= nk) return;
  uint32_t const f = gload(flags + i);
  int32_t const g  = static_cast<int32_t>(gload(before + i) + f) - 1;
  gstore(labels + i, g);
  if (f) {
    gstore(offsets + g, static_cast<int32_t>(i));
    gstore(first_row + g, static_cast<int32_t>(gload(order + i)));
  }
}

__global__ void __launch_bounds__(256) k_fill_i32(int32_t* out, int64_t n, int32_t v)
{
  int64_t const i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) gstore(out + i, v);
}

// row_label[order[i]] = labels[i]
__global__ void __launch_bounds__(256) k_row_labels(uint32_t const* order, int32_t const* labels, int64_t nk, int32_t* row_label)
{
  int64_t const i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < nk) gstore(row_label + gload(order + i), gload(labels + i));
}

// where[label] = j for the engine's j-th result row; the label G (rows of excluded keys) is dropped
__global__ void __launch_bounds__(256) k_invert_labels(int32_t const* result_labels, int64_t rows, int32_t G, int32_t* where)
{
  int64_t const j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (j >= rows) return;
  int32_t const g = gload(result_labels + j);
  if (g >= 0 && g < G) gstore(where + g, static_cast<int32_t>(j));
}

// ---------------------------------------------------------------------------------------------------------
// kinds
__global__ void __launch_bounds__(256) k_valid_flags(device_column col, uint32_t const* order, int64_t nk, uint32_t* flags)
{
  int64_t const i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < nk) gstore(flags + i, col_is_valid(col, gload(order + i)) ? 1u : 0u);
}

// NTH_ELEMENT counting every row (reference group_nth_element.cu:58-72): index = the group's n-th row in key order
__global__ void __launch_bounds__(256) k_nth_of_all(int32_t const* offsets, int32_t G, int32_t n, uint32_t const* order, int32_t* index,
                                                    int32_t none)
{
  int32_t const g = blockIdx.x * 256 + threadIdx.x;
  if (g >= G) return;
  int32_t const first = gload(offsets + g), size = gload(offsets + g + 1) - first;
  bool const inside   = n < 0 ? size >= -n : size > n;
  gstore(index + g, inside ? static_cast<int32_t>(gload(order + first + (n < 0 ? size + n : n))) : none);
}

// NTH_ELEMENT skipping nulls (:73-116): the row whose count of valid rows before it inside the group equals n
__global__ void __launch_bounds__(256) k_nth_of_valid(uint32_t const* valid, uint32_t const* valid_before /* nk + 1 */, int32_t const* labels,
                                                      int32_t const* offsets, int64_t nk, int32_t n, uint32_t const* order, int32_t* index)
{
  int64_t const i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= nk || !gload(valid + i)) return;
  int32_t const g     = gload(labels + i);
  uint32_t const base = gload(valid_before + gload(offsets + g));
  int32_t const intra = static_cast<int32_t>(gload(valid_before + i) - base);
  int32_t nth         = n;
  if (n < 0) nth = static_cast<int32_t>(gload(valid_before + gload(offsets + g + 1)) - base) + n;
  if (intra == nth) gstore(index + g, static_cast<int32_t>(gload(order + i)));
}

// vrow[i] = order[q[i]]: the row behind position i of the (label, null, value) order
__global__ void __launch_bounds__(256) k_compose(uint32_t const* order, uint32_t const* q, int64_t nk, uint32_t* vrow)
{
  int64_t const i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < nk) gstore(vrow + i, gload(order + gload(q + i)));
}

// NUNIQUE (group_nunique.cu:50-59): 1 where a countable row starts a run of equal values inside its group
__global__ void __launch_bounds__(256) k_unique_flags(device_column col, uint32_t const* vrow, int32_t const* labels, int32_t const* offsets,
                                                      int64_t nk, bool count_nulls, uint32_t* flags)
{
  int64_t const i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= nk) return;
  uint32_t const row = gload(vrow + i);
  bool const valid   = col_is_valid(col, row);
  uint32_t f         = 0;
  if (valid || count_nulls) {
    if (gload(offsets + gload(labels + i)) == i) {
      f = 1;
    } else {
      uint32_t const prev   = gload(vrow + i - 1);
      bool const prev_valid = col_is_valid(col, prev);
      bool equal            = valid == prev_valid;
      if (equal && valid) equal = sortable_word(col, row) == sortable_word(col, prev);
      f = equal ? 0u : 1u;
    }
  }
  gstore(flags + i, f);
}

// out[g] = scanned[offsets[g + 1]] - scanned[offsets[g]]  (scanned holds nk + 1 entries)
__global__ void __launch_bounds__(256) k_segment_sums(uint32_t const* scanned, int32_t const* offsets, int32_t G, int32_t* out)
{
  int32_t const g = blockIdx.x * 256 + threadIdx.x;
  if (g < G) gstore(out + g, static_cast<int32_t>(gload(scanned + gload(offsets + g + 1)) - gload(scanned + gload(offsets + g))));
}
__global__ void __launch_bounds__(256) k_group_sizes(int32_t const* offsets, int32_t G, int32_t* out)
{
  int32_t const g = blockIdx.x * 256 + threadIdx.x;
  if (g < G) gstore(out + g, gload(offsets + g + 1) - gload(offsets + g));
}

__device__ __forceinline__ double as_double(device_column const& c, uint32_t row)
{
  uint64_t const raw = col_load_bits(c, row);
  switch (c.cls) {
    case CLS_SINT: {
      int const sh = 64 - 8 * c.width;
      return static_cast<double>(static_cast<int64_t>(raw << sh) >> sh);
    }
    case CLS_UINT: return static_cast<double>(raw);
    case CLS_BOOL: return raw != 0 ? 1.0 : 0.0;
    case CLS_F32: return static_cast<double>(__uint_as_float(static_cast<uint32_t>(raw)));
    default: return __longlong_as_double(static_cast<long long>(raw));
  }
}

struct quantile_args {
  device_column col;
  uint32_t const* vrow;
  int32_t const* offsets;
  int32_t const* counts;  // valid rows per group (they lead each group of vrow)
  double const* q;
  int32_t nq;
  int32_t interp;
  int64_t total;          // G * nq
  double* out;
  bitmask_type* mask;
  int32_t* null_count;
};

// one (group, quantile) per thread (reference group_quantiles.cu:43-71, quantiles_util.hpp:20-176)
__global__ void __launch_bounds__(256) k_quantiles(quantile_args a)
{
  int64_t const e   = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  bool const inside = e < a.total;
  bool valid        = false;
  if (inside) {
    int32_t const g    = static_cast<int32_t>(e / a.nq);
    int32_t const size = gload(a.counts + g);
    double result      = 0.0;
    if (size > 0) {
      valid                 = true;
      uint32_t const* first = a.vrow + gload(a.offsets + g);
      double const quantile = fmin(fmax(gload(a.q + e % a.nq), 0.0), 1.0);
      double const pos      = quantile * (size - 1);
      int32_t const lower   = static_cast<int32_t>(floor(pos));
      int32_t const higher  = static_cast<int32_t>(ceil(pos));
      double const fraction = pos - lower;
      auto at               = [&](int32_t k) { return as_double(a.col, gload(first + k)); };
      switch (a.interp) {
        case static_cast<int32_t>(interpolation::LOWER): result = at(lower); break;
        case static_cast<int32_t>(interpolation::HIGHER): result = at(higher); break;
        case static_cast<int32_t>(interpolation::NEAREST): result = at(static_cast<int32_t>(nearbyint(pos))); break;
        case static_cast<int32_t>(interpolation::NEAREST_HALF_UP): result = at(static_cast<int32_t>(round(pos))); break;
        case static_cast<int32_t>(interpolation::MIDPOINT):
          if (a.col.cls == CLS_SINT && a.col.width == 8) {
            // halves and remainders apart, so that no sum leaves the int64 range (quantiles_util.hpp:46-53)
            int64_t const l = static_cast<int64_t>(col_load_bits(a.col, gload(first + lower)));
            int64_t const h = static_cast<int64_t>(col_load_bits(a.col, gload(first + higher)));
            result          = static_cast<double>(l / 2 + h / 2) + static_cast<double>(l % 2 + h % 2) * 0.5;
          } else {
            result = at(lower) / 2 + at(higher) / 2;
          }
          break;
        default: {  // LINEAR
          double const one_minus = 1.0 - fraction;
          result                 = one_minus * at(lower) + fraction * at(higher);
        }
      }
    }
    gstore(a.out + e, result);
  }
  uint64_t const m = __ballot(valid);
  int const lane   = threadIdx.x & 63;
  int64_t const w0 = (e - lane) >> 5;  // first mask word of this wave's 64 elements
  if (lane == 0 && e < a.total) {
    gstore(a.mask + w0, static_cast<bitmask_type>(m));
    int64_t const live = a.total - e < 64 ? a.total - e : 64;
    int32_t const nulls = static_cast<int32_t>(live) - __popcll(m);
    if (nulls) atomicAdd(a.null_count, nulls);
  }
  if (lane == 32 && e < a.total) gstore(a.mask + w0 + 1, static_cast<bitmask_type>(m >> 32));
}

bool arithmetic(type_id t) { return t >= type_id::INT8 && t <= type_id::BOOL8; }

unsigned blocks_of(int64_t n) { return static_cast<unsigned>(std::max<int64_t>((n + 255) / 256, 1)); }

// The reference's sort_groupby_helper (cpp/include/cudf/detail/groupby/sort_helper.hpp): the key order, the group labels and offsets,
// and per values column the order inside the groups.
struct helper {
  table_view keys;
  device_table dkeys;
  int64_t n{0};    // rows
  int64_t nk{0};   // rows whose key takes part (n minus the rows of excluded null keys)
  int32_t G{0};    // groups
  uint32_t* order{nullptr};     // position in key order -> row, nk used
  int32_t* labels{nullptr};     // per position
  int32_t* offsets{nullptr};    // G + 1
  int32_t* first_row{nullptr};  // per group: a row holding its key
  scratch keep;                 // the arrays above
  hipStream_t s;

  helper(table_view const& k, hipStream_t stream) : keys{k}, dkeys{cudf::detail::make_device_table(k)}, keep{stream, get_current_device_resource_ref(), {}}, s{stream} {}

  // inner_values: a values column whose order INSIDE the groups is wanted and nobody needs the rows of a group in their original
  // order (no NTH_ELEMENT in the call): its words are sorted first, the key columns after them - one sort yields the groups AND that
  // column's (key, null, value) order, instead of a key sort plus a second sort by (label, null, value): 11 instead of 14 passes and
  // one gather instead of two at 100M rows on 1M groups.
  void build(bool exclude_null_keys, bool presorted, column_view const* inner_values = nullptr)
  {
    n = keys.num_rows();
    bool const keys_have_nulls = std::any_of(keys.begin(), keys.end(), [](auto const& c) { return c.has_nulls(); });
    // pre-sorted keys with nulls to drop are sorted like any others (reference sort_helper.cu:51-54)
    if (presorted && exclude_null_keys && keys_have_nulls) presorted = false;
    order = keep.alloc<uint32_t>(static_cast<std::size_t>(n));
    nk    = n;
    scratch tmp{s, get_current_device_resource_ref(), {}};
    auto* flags       = tmp.alloc<uint32_t>(static_cast<std::size_t>(n));
    bool flags_filled = false;
    if (presorted) {  // the rows as they lie: no sorter, no scratch for one
      hipLaunchKernelGGL(k_iota, dim3(blocks_of(n)), dim3(256), 0, s, order, n);
    } else {
      scratch sort_tmp{s, get_current_device_resource_ref(), {}};
      pair_sorter sorter{n, sort_tmp, s};
      if (!presorted && inner_values != nullptr) {
        fill_args fa{};
        fa.col    = cudf::detail::make_device_column(*inner_values);
        fa.n      = n;
        fa.or_and = sorter.or_and();
        fa.source = WORD_VALUE;
        sorter.reset_reduction();
        fa.payload = sorter.payload();
        fa.words   = sorter.words();
        {
          prof::scope p_{"sort_words", s};
          hipLaunchKernelGGL(k_fill_words, dim3(fill_grid(n)), dim3(256), 0, s, fa);
        }
        sorter.sort_filled();
        if (inner_values->has_nulls()) {  // nulls after the values
          sorter.reset_reduction();
          fa.source  = WORD_NULL_FLAG;
          fa.payload = sorter.payload();
          fa.words   = sorter.words();
          hipLaunchKernelGGL(k_fill_words, dim3(fill_grid(n)), dim3(256), 0, s, fa);
          sorter.sort_filled();
        }
        _value_orders_in_key_order = std::make_pair(inner_values->head(), inner_values->offset());
      }
      if (!presorted) {
        for (int c = dkeys.ncols - 1; c >= 0; --c) {
          fill_args fa{};
          fa.col     = dkeys.col[c];
          fa.n       = n;
          fa.or_and  = sorter.or_and();
          fa.source  = WORD_VALUE;
          sorter.reset_reduction();
          fa.payload = sorter.payload();
          fa.words   = sorter.words();
          {
            prof::scope p_{"sort_words", s};
            hipLaunchKernelGGL(k_fill_words, dim3(fill_grid(n)), dim3(256), 0, s, fa);
          }
          sorter.sort_filled();
          if (keys.column(c).has_nulls()) {  // nulls after the values of the column
            sorter.reset_reduction();
            fa.source  = WORD_NULL_FLAG;
            fa.payload = sorter.payload();
            fa.words   = sorter.words();
            hipLaunchKernelGGL(k_fill_words, dim3(fill_grid(n)), dim3(256), 0, s, fa);
            sorter.sort_filled();
          }
        }
        if (exclude_null_keys && keys_have_nulls) {  // rows with a null in any key column to the back, then off the end
          fill_args fa{};
          fa.keys   = dkeys;
          fa.n      = n;
          fa.or_and = sorter.or_and();
          fa.ones   = sorter.ones();
          fa.source = WORD_ANY_NULL;
          sorter.reset_reduction();
          fa.payload = sorter.payload();
          fa.words   = sorter.words();
          hipLaunchKernelGGL(k_fill_words, dim3(fill_grid(n)), dim3(256), 0, s, fa);
          nk = n - sorter.sort_filled();
        }
      }
      CUDF_HIP_TRY(hipMemcpyAsync(order, sorter.finish(), static_cast<std::size_t>(n) * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
      if (!presorted && dkeys.ncols == 1 && !keys_have_nulls) {
        prof::scope p_{"sort_boundaries", s};
        hipLaunchKernelGGL(k_boundaries_of_words, dim3(blocks_of(n)), dim3(256), 0, s, sorter.words(), n, flags);
        flags_filled = true;
      }
      CUDF_HIP_TRY(hipGetLastError());
    }
    if (nk == 0) {
      G = 0;
      return;
    }
    labels       = keep.alloc<int32_t>(static_cast<std::size_t>(nk));
    auto* before = tmp.alloc<uint32_t>(static_cast<std::size_t>(nk) + 1);
    if (!flags_filled) {
      prof::scope p_{"sort_boundaries", s};
      hipLaunchKernelGGL(k_boundaries, dim3(blocks_of(nk)), dim3(256), 0, s, dkeys, order, nk, flags);
    }
    exclusive_scan(flags, before, nk, before + nk, tmp, s);
    uint32_t h_groups = 0;
    CUDF_HIP_TRY(hipMemcpyAsync(&h_groups, before + nk, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    G         = static_cast<int32_t>(h_groups);
    offsets   = keep.alloc<int32_t>(static_cast<std::size_t>(G) + 1);
    first_row = keep.alloc<int32_t>(static_cast<std::size_t>(G));
    hipLaunchKernelGGL(k_labels, dim3(blocks_of(nk)), dim3(256), 0, s, flags, before, order, nk, labels, offsets, first_row);
    hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(256), 0, s, offsets + G, 1, static_cast<int32_t>(nk));
    CUDF_HIP_TRY(hipGetLastError());
  }

  // rows in (label, null, value) order (reference sort_helper.cu:205-225 sorted_values): vrow[i] for i in the group's range
  uint32_t* value_order(column_view const& values)
  {
    auto const key = std::make_pair(values.head(), values.offset());
    if (_value_orders_in_key_order == key) return order;  // (build() sorted this column's words before the keys)
    if (auto it = _value_orders.find(key); it != _value_orders.end()) return it->second;
    auto* vrow = keep.alloc<uint32_t>(static_cast<std::size_t>(nk));
    {
      auto const col = cudf::detail::make_device_column(values);
      scratch tmp{s, get_current_device_resource_ref(), {}};
      pair_sorter sorter{nk, tmp, s};
      sorter.reset_reduction();
      {
        prof::scope p_{"sort_words", s};
        hipLaunchKernelGGL(k_fill_value_words, dim3(fill_grid(nk)), dim3(256), 0, s, col, order, nk, sorter.words(), sorter.or_and());
      }
      sorter.sort_filled();
      fill_args fa{};
      fa.col     = col;
      fa.order   = order;
      fa.labels  = labels;
      fa.n       = nk;
      fa.or_and  = sorter.or_and();
      fa.source  = WORD_LABEL_NULL;
      sorter.reset_reduction();
      fa.payload = sorter.payload();
      fa.words   = sorter.words();
      {
        prof::scope p_{"sort_words", s};
        hipLaunchKernelGGL(k_fill_words, dim3(fill_grid(nk)), dim3(256), 0, s, fa);
      }
      sorter.sort_filled();
      hipLaunchKernelGGL(k_compose, dim3(blocks_of(nk)), dim3(256), 0, s, order, sorter.finish(), nk, vrow);
      CUDF_HIP_TRY(hipGetLastError());
    }
    _value_orders.emplace(key, vrow);
    return vrow;
  }

  // valid rows of `values` before each position of the key order (nk + 1 entries), and per group
  uint32_t* valid_before(column_view const& values, scratch& tmp, uint32_t** flags_out = nullptr)
  {
    auto* flags  = tmp.alloc<uint32_t>(static_cast<std::size_t>(nk));
    auto* before = tmp.alloc<uint32_t>(static_cast<std::size_t>(nk) + 1);
    hipLaunchKernelGGL(k_valid_flags, dim3(blocks_of(nk)), dim3(256), 0, s, cudf::detail::make_device_column(values), order, nk, flags);
    exclusive_scan(flags, before, nk, before + nk, tmp, s);
    if (flags_out) *flags_out = flags;
    return before;
  }

 private:
  std::map<std::pair<void const*, size_type>, uint32_t*> _value_orders;
  std::pair<void const*, size_type> _value_orders_in_key_order{nullptr, 0};
};

std::unique_ptr<column> int32_column(int64_t size, stream_ref stream, rmm::device_async_resource_ref mr)
{
  return std::make_unique<column>(data_type{type_id::INT32}, static_cast<size_type>(size),
                                  rmm::device_buffer{static_cast<std::size_t>(size) * sizeof(int32_t), stream.value(), mr}, rmm::device_buffer{}, 0);
}

// result column j of the engine's answer, reordered by `where` (one INT32 per group)
std::unique_ptr<column> reorder(column_view const& c, column_view const& where, stream_ref stream, rmm::device_async_resource_ref mr)
{
  if (c.type().id() == type_id::STRUCT) {
    // SUM_OVERFLOW {sum, overflow}: the children carry no masks, the struct's mask travels with the first child and returns to the struct
    column_view const sum_with_mask{c.child(0).type(), c.size(), c.child(0).head(), c.null_mask(), c.null_count(), c.child(0).offset()};
    auto t    = cudf::gather(table_view{{sum_with_mask, c.child(1)}}, where, out_of_bounds_policy::DONT_CHECK, stream, mr);
    auto cols = t->release();
    auto const nulls = cols[0]->null_count();
    auto sum         = cols[0]->release();
    std::vector<std::unique_ptr<column>> children;
    children.push_back(std::make_unique<column>(c.child(0).type(), where.size(), std::move(*sum.data), rmm::device_buffer{}, 0));
    children.push_back(std::move(cols[1]));
    return std::make_unique<column>(data_type{type_id::STRUCT}, where.size(), rmm::device_buffer{},
                                    c.nullable() ? std::move(*sum.null_mask) : rmm::device_buffer{}, c.nullable() ? nulls : 0, std::move(children));
  }
  auto t = cudf::gather(table_view{{c}}, where, out_of_bounds_policy::DONT_CHECK, stream, mr);
  return std::move(t->release()[0]);
}
}  // namespace

namespace {
__global__ void __launch_bounds__(256) k_count_runs(device_table keys, int64_t n, unsigned long long* out)
{
  unsigned long long mine = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256)
    mine += (i == 0 || !cudf::detail::rows_equal(keys, i - 1, keys, i, true)) ? 1u : 0u;
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}
}  // namespace

// the number of runs of equal adjacent key rows (nulls equal nulls): one coalesced pass over the key columns
int64_t count_key_runs(table_view const& keys, stream_ref stream)
{
  int64_t const n = keys.num_rows();
  if (n == 0) return 0;
  hipStream_t const s = stream.value();
  rmm::device_buffer d{sizeof(unsigned long long), s, get_current_device_resource_ref()};
  CUDF_HIP_TRY(hipMemsetAsync(d.data(), 0, sizeof(unsigned long long), s));
  {
    prof::scope p_{"sort_boundaries", s};
    hipLaunchKernelGGL(k_count_runs, dim3(static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 256 * 32))), dim3(256), 0, s,
                       cudf::detail::make_device_table(keys), n, static_cast<unsigned long long*>(d.data()));
  }
  CUDF_HIP_TRY(hipGetLastError());
  unsigned long long h = 0;
  CUDF_HIP_TRY(hipMemcpyAsync(&h, d.data(), sizeof(h), hipMemcpyDeviceToHost, s));
  CUDF_HIP_TRY(hipStreamSynchronize(s));
  return static_cast<int64_t>(h);
}

std::pair<std::unique_ptr<table>, std::vector<aggregation_result>> sort_aggregate(table_view const& keys, null_policy include_null_keys,
                                                                                  bool keys_are_sorted,
                                                                                  std::span<aggregation_request const> requests,
                                                                                  stream_ref stream, rmm::device_async_resource_ref mr)
{
  hipStream_t const s = stream.value();
  CUDF_EXPECTS(keys.num_columns() > 0, "The sort-based groupby needs at least one key column.");
  helper h{keys, s};
  // which order the one sort should produce (helper::build)
  column_view const* inner_values = nullptr;
  bool any_nth                    = false;
  for (auto const& r : requests)
    for (auto const& a : r.aggregations) {
      any_nth = any_nth || a->kind == aggregation::NTH_ELEMENT;
      if (inner_values == nullptr && (a->kind == aggregation::MEDIAN || a->kind == aggregation::QUANTILE || a->kind == aggregation::NUNIQUE) &&
          r.values.size() == keys.num_rows() && r.values.head() != nullptr)
        inner_values = &r.values;
    }
  h.build(include_null_keys == null_policy::EXCLUDE, keys_are_sorted, any_nth ? nullptr : inner_values);
  int32_t const G = h.G;

  std::vector<aggregation_result> results(requests.size());
  for (std::size_t r = 0; r < requests.size(); ++r) results[r].results.resize(requests[r].aggregations.size());

  if (G == 0) {  // every key excluded: typed empty results
    for (std::size_t r = 0; r < requests.size(); ++r)
      for (std::size_t j = 0; j < requests[r].aggregations.size(); ++j) {
        auto const kind = requests[r].aggregations[j]->kind;
        if (kind == aggregation::SUM_OVERFLOW) {
          std::vector<std::unique_ptr<column>> children;
          children.push_back(make_empty_column(requests[r].values.type()));
          children.push_back(make_empty_column(data_type{type_id::BOOL8}));
          results[r].results[j] =
            std::make_unique<column>(data_type{type_id::STRUCT}, 0, rmm::device_buffer{}, rmm::device_buffer{}, 0, std::move(children));
        } else {
          results[r].results[j] = make_empty_column(cudf::detail::target_type(requests[r].values.type(), kind));
        }
      }
    return {empty_like(keys), std::move(results)};
  }

  // the unique keys, ascending
  column_view const first_rows{data_type{type_id::INT32}, G, h.first_row, nullptr, 0};
  auto unique_keys = cudf::gather(keys, first_rows, out_of_bounds_policy::DONT_CHECK, stream, mr);

  // ---- the kinds the hash engine serves: keyed on the row's group label
  {
    std::vector<aggregation_request> engine_requests;
    std::vector<std::pair<std::size_t, std::size_t>> slots;  // (request, aggregation) of every engine result, in order
    for (std::size_t r = 0; r < requests.size(); ++r) {
      aggregation_request er{requests[r].values, {}};
      for (std::size_t j = 0; j < requests[r].aggregations.size(); ++j) {
        auto const& a = requests[r].aggregations[j];
        if (!is_engine_kind(a->kind)) continue;
        auto clone = a->clone();
        auto* as_groupby = dynamic_cast<groupby_aggregation*>(clone.get());
        CUDF_EXPECTS(as_groupby != nullptr, "not a groupby aggregation");
        clone.release();
        er.aggregations.emplace_back(as_groupby);
        slots.emplace_back(r, j);
      }
      if (!er.aggregations.empty()) engine_requests.push_back(std::move(er));
    }
    if (!engine_requests.empty()) {
      scratch tmp{s, get_current_device_resource_ref(), {}};
      auto* row_label = tmp.alloc<int32_t>(static_cast<std::size_t>(h.n));
      if (h.nk < h.n) hipLaunchKernelGGL(k_fill_i32, dim3(blocks_of(h.n)), dim3(256), 0, s, row_label, h.n, G);
      hipLaunchKernelGGL(k_row_labels, dim3(blocks_of(h.nk)), dim3(256), 0, s, h.order, h.labels, h.nk, row_label);
      CUDF_HIP_TRY(hipGetLastError());
      column_view const label_col{data_type{type_id::INT32}, static_cast<size_type>(h.n), row_label, nullptr, 0};
      cudf::groupby::groupby by_label{table_view{{label_col}}, null_policy::INCLUDE};
      auto [label_keys, engine_results] = by_label.aggregate(engine_requests, stream, get_current_device_resource_ref());
      auto where = int32_column(G, stream, get_current_device_resource_ref());
      hipLaunchKernelGGL(k_invert_labels, dim3(blocks_of(label_keys->num_rows())), dim3(256), 0, s,
                         label_keys->get_column(0).view().data<int32_t>(), static_cast<int64_t>(label_keys->num_rows()), G,
                         where->mutable_view().data<int32_t>());
      CUDF_HIP_TRY(hipGetLastError());
      std::size_t slot = 0;
      for (auto& er : engine_results)
        for (auto& c : er.results) {
          auto const [r, j]     = slots[slot++];
          results[r].results[j] = reorder(c->view(), where->view(), stream, mr);
        }
    }
  }

  // ---- the kinds that need the order
  for (std::size_t r = 0; r < requests.size(); ++r) {
    auto const& values = requests[r].values;
    auto const dcol    = cudf::detail::make_device_column(values);
    for (std::size_t j = 0; j < requests[r].aggregations.size(); ++j) {
      auto const& a = *requests[r].aggregations[j];
      if (!is_sort_kind(a.kind)) continue;
      scratch tmp{s, get_current_device_resource_ref(), {}};
      if (a.kind == aggregation::NTH_ELEMENT) {
        auto const& nth = dynamic_cast<cudf::detail::nth_element_aggregation const&>(a);
        auto index      = int32_column(G, stream, get_current_device_resource_ref());
        auto* d_index   = index->mutable_view().data<int32_t>();
        int32_t const none = values.size();  // out of bounds: a null row of the gather
        if (nth._null_handling == null_policy::INCLUDE || !values.has_nulls()) {
          hipLaunchKernelGGL(k_nth_of_all, dim3(blocks_of(G)), dim3(256), 0, s, h.offsets, G, nth._n, h.order, d_index, none);
        } else {
          uint32_t* valid = nullptr;
          auto* before    = h.valid_before(values, tmp, &valid);
          hipLaunchKernelGGL(k_fill_i32, dim3(blocks_of(G)), dim3(256), 0, s, d_index, G, none);
          hipLaunchKernelGGL(k_nth_of_valid, dim3(blocks_of(h.nk)), dim3(256), 0, s, valid, before, h.labels, h.offsets, h.nk, nth._n, h.order,
                             d_index);
        }
        CUDF_HIP_TRY(hipGetLastError());
        auto t   = cudf::gather(table_view{{values}}, index->view(), out_of_bounds_policy::NULLIFY, stream, mr);
        auto col = std::move(t->release()[0]);
        if (!col->has_nulls()) col->set_null_mask(rmm::device_buffer{}, 0);
        results[r].results[j] = std::move(col);
      } else if (a.kind == aggregation::NUNIQUE) {
        auto const& nu = dynamic_cast<cudf::detail::nunique_aggregation const&>(a);
        auto* vrow     = h.value_order(values);
        auto* flags    = tmp.alloc<uint32_t>(static_cast<std::size_t>(h.nk) + 1);
        hipLaunchKernelGGL(k_unique_flags, dim3(blocks_of(h.nk)), dim3(256), 0, s, dcol, vrow, h.labels, h.offsets, h.nk,
                           nu._null_handling == null_policy::INCLUDE, flags);
        exclusive_scan(flags, flags, h.nk, flags + h.nk, tmp, s);
        auto out = int32_column(G, stream, mr);
        hipLaunchKernelGGL(k_segment_sums, dim3(blocks_of(G)), dim3(256), 0, s, flags, h.offsets, G, out->mutable_view().data<int32_t>());
        CUDF_HIP_TRY(hipGetLastError());
        results[r].results[j] = std::move(out);
      } else {  // MEDIAN, QUANTILE
        CUDF_EXPECTS(arithmetic(values.type().id()), "Only arithmetic types are supported in quantiles");
        std::vector<double> q{0.5};
        auto interp = interpolation::LINEAR;
        if (a.kind == aggregation::QUANTILE) {
          auto const& qa = dynamic_cast<cudf::detail::quantile_aggregation const&>(a);
          q              = qa._quantiles;
          interp         = qa._interpolation;
        }
        auto* vrow   = h.value_order(values);
        auto* counts = tmp.alloc<int32_t>(static_cast<std::size_t>(G));
        if (values.has_nulls()) {
          auto* before = h.valid_before(values, tmp);
          hipLaunchKernelGGL(k_segment_sums, dim3(blocks_of(G)), dim3(256), 0, s, before, h.offsets, G, counts);
        } else {
          hipLaunchKernelGGL(k_group_sizes, dim3(blocks_of(G)), dim3(256), 0, s, h.offsets, G, counts);
        }
        int64_t const total = static_cast<int64_t>(G) * static_cast<int64_t>(q.size());
        CUDF_EXPECTS(total <= std::numeric_limits<size_type>::max(), "quantile result exceeds the column size limit", std::overflow_error);
        auto out = make_fixed_width_column(data_type{type_id::FLOAT64}, static_cast<size_type>(total), mask_state::UNINITIALIZED, stream, mr);
        if (total > 0) {
          auto* d_q     = tmp.alloc<double>(q.size());
          auto* d_nulls = tmp.alloc<int32_t>(1);
          CUDF_HIP_TRY(hipMemcpyAsync(d_q, q.data(), q.size() * sizeof(double), hipMemcpyHostToDevice, s));
          CUDF_HIP_TRY(hipMemsetAsync(d_nulls, 0, sizeof(int32_t), s));
          quantile_args qa{};
          qa.col        = dcol;
          qa.vrow       = vrow;
          qa.offsets    = h.offsets;
          qa.counts     = counts;
          qa.q          = d_q;
          qa.nq         = static_cast<int32_t>(q.size());
          qa.interp     = static_cast<int32_t>(interp);
          qa.total      = total;
          qa.out        = out->mutable_view().data<double>();
          qa.mask       = out->mutable_view().null_mask();
          qa.null_count = d_nulls;
          {
            prof::scope p_{"sort_quantiles", s};
            hipLaunchKernelGGL(k_quantiles, dim3(blocks_of(total)), dim3(256), 0, s, qa);
          }
          CUDF_HIP_TRY(hipGetLastError());
          int32_t h_nulls = 0;
          CUDF_HIP_TRY(hipMemcpyAsync(&h_nulls, d_nulls, sizeof(int32_t), hipMemcpyDeviceToHost, s));
          CUDF_HIP_TRY(hipStreamSynchronize(s));
          out->set_null_count(h_nulls);
        }
        results[r].results[j] = std::move(out);
      }
    }
  }
  return {std::move(unique_keys), std::move(results)};
}

}  // namespace cudf::groupby::detail
