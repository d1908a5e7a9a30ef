// SPDX-License-Identifier: Apache-2.0
// Path A front end for the plain shape (one 8-byte integer key, 8-byte values, no NULLs): runs of equal keys collapse to one
// partial record each WITHOUT a hash table. A wave reads 64 consecutive rows per row set, marks the rows that start a run
// (key != the key one lane down; lane 0 always starts one), reduces every run with a segmented scan over the lanes (six steps;
// a lane adds the lane `o` below only if that lane is not below its run's first lane) and the last lane of each run writes
// [key | accumulators] to the chunk's output region (positions from one LDS atomic per wave and batch). A run that straddles
// two row sets yields two records: the merge of the partial records (the exact partition pipeline, as before) does not care.
// The single-pass kernel that did this before probed its LDS table for every row and then combined the lanes that met in a
// slot: 1B rows in runs of 64 took 12.8 ms there against 7.1 ms for uniformly random keys (profiles/r3_sorted_keys.txt).
// Reference: none - the reference's global hash set (compute_global_memory_aggs.cuh:74-187) does not care about row order.
#include "engine.hpp"
#include "device_common.hpp"
#include "../common/profiler.hpp"

#include <cudf/utilities/error.hpp>

namespace cudf::groupby::detail {
namespace {

__global__ void k_store_collapse_args(collapse_args v, collapse_args* dst) { *dst = v; }

// inclusive segmented scan over the lanes of R row sets: x[j] of a lane becomes the combination of its run's rows up to the lane
template <int OP, int R>
__device__ __forceinline__ void seg_scan(uint64_t (&x)[R], int const (&first)[R], int lane)
{
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      uint64_t const other = __shfl_up(x[j], o);
      if (lane - o >= first[j]) x[j] = combine_values(OP, x[j], other);
    }
  }
}

template <int NACCT>
__global__ void __launch_bounds__(256) k_collapse_runs(collapse_args const* __restrict__ ap)
{
  collapse_args const& a = *ap;
  plan_dev const& p      = a.plan;
  __shared__ uint32_t s_cursor;
  __shared__ int s_overflow;
  constexpr int R = NACCT >= 3 ? 2 : 4;  // row sets in flight per wave (four with three or four accumulators took 172 VGPRs)
  // (tried: the scan's steps inside a row of 16 lanes as DPP row_shr moves + three readlane carries across rows instead of six
  // ds_bpermute steps - one scanned accumulator 3.5 -> 4.1 ms per 1B rows, three 7.2 -> 7.2 ms: the kernel is bound by instruction
  // issue - ~280 wave instructions per 64 rows with three scans - not by the LDS permutes)
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  int const item = blockIdx.x;
  if (threadIdx.x == 0) {
    s_cursor   = 0;
    s_overflow = 0;
  }
  __syncthreads();
  int op[NACCT], src[NACCT], pay[NACCT];
#pragma unroll
  for (int q = 0; q < NACCT; ++q) {
    op[q]  = p.acc[q].op;
    src[q] = p.acc[q].src;
    pay[q] = p.acc[q].pay;
  }
  uint64_t const* keys = p.simple_base[0];
  uint64_t const* val0 = p.NPAY > 0 ? p.simple_base[1] : keys;
  uint64_t const* val1 = p.NPAY > 1 ? p.simple_base[2] : val0;
  int const PU         = 1 + NACCT;
  uint64_t* out        = a.out_records + static_cast<int64_t>(item) * a.out_stride * PU;
  int64_t const begin = static_cast<int64_t>(item) * a.chunk, end = min(a.nrows, begin + a.chunk);
  uint64_t const below_incl = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
  for (int64_t base = begin + static_cast<int64_t>(wave) * 64 * R; base < end; base += static_cast<int64_t>(nwaves) * 64 * R) {
    uint64_t k[R], v0[R], v1[R];
    bool live[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      int64_t const r = base + j * 64 + lane;
      live[j]         = r < end;
      k[j]            = live[j] ? gload(keys + r) : 0;
      v0[j]           = live[j] && p.NPAY > 0 ? gload(val0 + r) : 0;
      v1[j]           = live[j] && p.NPAY > 1 ? gload(val1 + r) : 0;
    }
    unsigned long long tails[R];
    uint64_t acc[NACCT][R];
    int first[R];
    int tot = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      unsigned long long const lm = __ballot(live[j]);  // (a prefix of the lanes)
      uint64_t const kd           = __shfl_up(k[j], 1);
      unsigned long long const hm = __ballot(live[j] && (lane == 0 || kd != k[j]));      // lanes that start a run
      first[j]                    = 63 - __builtin_clzll((hm & below_incl) | 1ull);     // first lane of this lane's run
      tails[j]                    = ((hm >> 1) | (1ull << 63)) & lm;                      // last lanes of the runs ...
      if (lm != ~0ull && lm != 0) tails[j] |= 1ull << (__popcll(lm) - 1);                 // ... and of a short last row set
      tails[j] &= lm;
      tot += __popcll(tails[j]);
    }
#pragma unroll
    for (int q = 0; q < NACCT; ++q) {
      if (src[q] == SRC_ONE && op[q] == ADD_I64) {  // a row count: the lane's distance from its run's first lane
#pragma unroll
        for (int j = 0; j < R; ++j) acc[q][j] = static_cast<uint64_t>(lane - first[j] + 1);
        continue;
      }
#pragma unroll
      for (int j = 0; j < R; ++j) acc[q][j] = src[q] == SRC_ONE ? 1ull : acc_contribution(src[q], op[q], pay[q] == 1 ? v1[j] : v0[j]);
      switch (op[q]) {  // (one uniform branch per accumulator and batch; the scans themselves are straight-line)
        case ADD_I64: seg_scan<ADD_I64, R>(acc[q], first, lane); break;
        case ADD_F64: seg_scan<ADD_F64, R>(acc[q], first, lane); break;
        case MIN_I64: seg_scan<MIN_I64, R>(acc[q], first, lane); break;
        case MIN_U64: seg_scan<MIN_U64, R>(acc[q], first, lane); break;
        case MAX_I64: seg_scan<MAX_I64, R>(acc[q], first, lane); break;
        case MAX_U64: seg_scan<MAX_U64, R>(acc[q], first, lane); break;
        case MIN_F64: seg_scan<MIN_F64, R>(acc[q], first, lane); break;
        case MAX_F64: seg_scan<MAX_F64, R>(acc[q], first, lane); break;
        case MUL_I64: seg_scan<MUL_I64, R>(acc[q], first, lane); break;
        case MUL_F64: seg_scan<MUL_F64, R>(acc[q], first, lane); break;
        default: break;  // ANY_U64: every row of a run will do
      }
    }
    if (tot == 0) continue;
    uint32_t pos = 0;
    if (lane == 0) pos = atomicAdd(&s_cursor, static_cast<uint32_t>(tot));
    pos = __builtin_amdgcn_readfirstlane(pos);
    if (static_cast<int64_t>(pos) + tot > a.out_stride) {  // more runs than planned for: the caller takes another path
      if (lane == 0) s_overflow = 1;
      continue;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if ((tails[j] >> lane) & 1ull) {
        uint64_t* o = out + static_cast<int64_t>(pos + __popcll(tails[j] & (below_incl >> 1))) * PU;
        gstore(o, k[j]);
#pragma unroll
        for (int q = 0; q < NACCT; ++q) gstore(o + 1 + q, acc[q][j]);
      }
      pos += __popcll(tails[j]);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.out_count[item] = s_overflow ? 0 : static_cast<int32_t>(s_cursor);
    if (s_overflow) atomicOr(a.overflow, 2);
  }
}

}  // namespace

bool collapse_runs_applies(plan_dev const& p)
{
  if (!p.simple || p.KU != 1 || p.NPAY > 2 || p.NACC < 1 || p.NACC > 4 || p.narg != 0) return false;
  for (int q = 0; q < p.NACC; ++q) {
    auto const& d = p.acc[q];
    if (d.valid_bit >= 0 || d.pay > 1 || (d.src != SRC_VALUE && d.src != SRC_ONE && d.src != SRC_SQUARE)) return false;
    if (d.src != SRC_ONE && d.pay < 0) return false;
  }
  return true;
}

void launch_collapse_runs(collapse_args const& a, collapse_args* d_args, hipStream_t stream)
{
  CUDF_EXPECTS(collapse_runs_applies(a.plan) && a.nitems >= 1 && a.chunk >= 1 && a.out_stride >= 64, "groupby run collapse: arguments");
  hipLaunchKernelGGL(k_store_collapse_args, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"collapse_runs", stream};
  switch (a.plan.NACC) {
    case 1: hipLaunchKernelGGL(k_collapse_runs<1>, dim3(a.nitems), dim3(256), 0, stream, d_args); break;
    case 2: hipLaunchKernelGGL(k_collapse_runs<2>, dim3(a.nitems), dim3(256), 0, stream, d_args); break;
    case 3: hipLaunchKernelGGL(k_collapse_runs<3>, dim3(a.nitems), dim3(256), 0, stream, d_args); break;
    default: hipLaunchKernelGGL(k_collapse_runs<4>, dim3(a.nitems), dim3(256), 0, stream, d_args); break;
  }
  CUDF_HIP_TRY(hipGetLastError());
}

}  // namespace cudf::groupby::detail
