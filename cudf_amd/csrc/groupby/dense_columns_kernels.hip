// SPDX-License-Identifier: Apache-2.0
// gfx950 kernel of the dense-key hash-groupby path for LOW cardinality: the whole key range fits one direct-address LDS table
// (engine.hpp launch_aggregate_dense_columns). Every workgroup aggregates its row tiles straight from the columns into a table of
// its own - index = key - lo is the slot: no hash, no probe, no key compare - and leaves the table image for k_dense_merge_dump.
// Most of a wave on one slot (a heavy key, few groups) is reduced across the wave first, as in the hash-table kernel.
// Replaces the reference's shared-memory aggregation (cpp/src/groupby/hash/compute_shared_memory_aggs.cu:260-353,
// compute_mapping_indices.cuh:92-151) for such keys.
#include "device_common.hpp"
#include "dense_loader.hpp"

namespace cudf::groupby::detail {
namespace {

constexpr int DCOL_SIMPLE = 0;  // one plain 8-byte integer key column, one plain 8-byte value column
constexpr int DCOL_COLS   = 1;  // composite dense keys (dense_loader.hpp)

// NV: value columns (DCOL_SIMPLE only): the row's values sit in registers, an accumulator takes the one its descriptor names
// (a compile-time selection for the signatures with an instantiation of their own).
template <uint64_t SIG, int NACCT, int SRC, int NV = 1>
__global__ void __launch_bounds__(1024) k_aggregate_dense_columns(dense_agg_args const* __restrict__ ap)
{
  static_assert(NV == 1 || SRC == DCOL_SIMPLE, "several value columns: plain columns only");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  dense_agg_args const& a = *ap;
  plan_dev const& p       = a.plan;
  constexpr bool STATIC_SIG = SIG != 0;
  // (three value columns: two rows per thread - with four, two tiles of prefetched rows took 128 VGPRs and spilled)
  constexpr int B = 1024, R = NV >= 3 ? 2 : 4;
  int const NACC  = STATIC_SIG ? sig_n(SIG) : p.NACC;
  int const slots = a.slots, occ_acc = a.occ_acc;
  int acc_op[NACCT], acc_src[NACCT], acc_pay[NACCT], acc_vbit[NACCT];
  uint32_t acc_off[NACCT];
  bool acc_narrow[NACCT];
  uint32_t off8 = 0, off4 = 0;  // (the layout of make_dense_layout: 8-byte arrays in accumulator order, then the 4-byte ones, then the bitmap)
#pragma unroll
  for (int j = 0; j < NACCT; ++j) {
    uint32_t const w = j < NACC ? reinterpret_cast<uint32_t const*>(p.acc)[j] : 0u;
    acc_op[j]        = STATIC_SIG ? sig_op(SIG, j) : static_cast<int8_t>(w);
    acc_src[j]       = STATIC_SIG ? sig_src(SIG, j) : static_cast<int8_t>(w >> 8);
    acc_pay[j]       = STATIC_SIG ? sig_pay(SIG, j) : static_cast<int8_t>(w >> 16);
    acc_vbit[j]      = STATIC_SIG ? sig_vbit(SIG, j) : static_cast<int8_t>(w >> 24);
    acc_narrow[j]    = j < NACC && acc_is_narrow(acc_op[j], acc_src[j]);
    if (j < NACC && !acc_narrow[j]) off4 += static_cast<uint32_t>(slots) * 8u;
  }
#pragma unroll
  for (int j = 0; j < NACCT; ++j) {
    if (j >= NACC) { acc_off[j] = 0; continue; }
    if (acc_narrow[j]) { acc_off[j] = off4; off4 += static_cast<uint32_t>(slots) * 4u; }
    else { acc_off[j] = off8; off8 += static_cast<uint32_t>(slots) * 8u; }
  }
  uint32_t* occ = reinterpret_cast<uint32_t*>(lds_raw + off4);  // (only if occ_acc < 0)
  auto acc64 = [&](int q) { return reinterpret_cast<uint64_t*>(lds_raw + acc_off[q]); };
  auto acc32 = [&](int q) { return reinterpret_cast<uint32_t*>(lds_raw + acc_off[q]); };
#pragma unroll
  for (int q = 0; q < NACCT; ++q) {
    if (q >= NACC) break;
    if (acc_narrow[q]) {
      for (int s = threadIdx.x; s < slots; s += B) acc32(q)[s] = 0;
    } else {
      uint64_t const id = acc_identity(acc_op[q]);
      for (int s = threadIdx.x; s < slots; s += B) acc64(q)[s] = id;
    }
  }
  if (occ_acc < 0)
    for (int s = threadIdx.x; s < (slots + 31) / 32; s += B) occ[s] = 0;
  __syncthreads();

  int const lane = threadIdx.x & 63;
  // one row into its slot (LDS atomics)
  // the value an accumulator reads: the row's only value, or the one of its column (static indices only: a run-time index into
  // the register array sent it to scratch)
  auto value_of = [&](int q, uint64_t const (&values)[NV]) -> uint64_t {
    if constexpr (NV == 1) return values[0];
    uint64_t v = values[0];
#pragma unroll
    for (int j = 1; j < NV; ++j)
      if (acc_pay[q] == j) v = values[j];
    return v;
  };
  auto accumulate = [&](uint32_t s, bool val_valid, uint64_t const (&values)[NV]) {
#pragma unroll
    for (int q = 0; q < NACCT; ++q) {
      if (q >= NACC) break;
      bool const counts_all = acc_src[q] == SRC_ONE;
      if (!counts_all && acc_vbit[q] >= 0 && !val_valid) continue;  // a NULL value reaches nothing but COUNT_ALL
      if (acc_narrow[q]) {
        atomicAdd(acc32(q) + s, 1u);
        continue;
      }
      lds_merge(acc64(q) + s, acc_op[q], acc_contribution(acc_src[q], acc_op[q], value_of(q, values)));
    }
    if (occ_acc < 0) {
      uint32_t const bit = 1u << (s & 31);
      if (!(occ[s >> 5] & bit)) atomicOr(&occ[s >> 5], bit);
    }
  };
  // the lanes in `mine` all go to slot s: every accumulator is reduced across the wave and lane `leader` issues one atomic
  // (same-address LDS atomics serialise: a key with percents of the rows, or a handful of groups)
  auto accumulate_wave = [&](uint32_t s, bool mine, bool leader, bool val_valid, uint64_t const (&values)[NV]) {
#pragma unroll
    for (int q = 0; q < NACCT; ++q) {
      if (q >= NACC) break;
      bool const valid            = mine && (acc_src[q] == SRC_ONE || acc_vbit[q] < 0 || val_valid);
      unsigned long long const vm = __ballot(valid);
      if (vm == 0) continue;  // (wave-uniform)
      if (acc_narrow[q]) {
        if (leader) atomicAdd(acc32(q) + s, static_cast<uint32_t>(__popcll(vm)));
        continue;
      }
      uint64_t v = valid ? acc_contribution(acc_src[q], acc_op[q], value_of(q, values)) : acc_identity(acc_op[q]);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
        v = combine_values(acc_op[q], v, static_cast<uint64_t>(__shfl_xor(static_cast<unsigned long long>(v), off)));
      if (leader) lds_merge(acc64(q) + s, acc_op[q], v);
    }
    if (occ_acc < 0 && leader) atomicOr(&occ[s >> 5], 1u << (s & 31));
  };
  // R row sets of a tile: slot, validity of the value, value
  auto accumulate_tile = [&](bool const (&act)[R], uint32_t const (&s)[R], uint32_t const (&valid)[R], uint64_t const (&val)[R][NV]) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      unsigned long long const am = __ballot(act[k]);
      if (am == 0) continue;  // (wave-uniform)
      uint32_t const lead = __shfl(s[k], __ffsll(static_cast<long long>(am)) - 1);
      bool const mine     = act[k] && s[k] == lead;
      unsigned long long const same = __ballot(mine);
      bool combined = false;
      if (__popcll(same) >= 16) {
        accumulate_wave(lead, mine, lane == __ffsll(static_cast<long long>(same)) - 1, valid[k] != 0, val[k]);
        combined = mine;
      }
      if (act[k] && !combined) accumulate(s[k], valid[k] != 0, val[k]);
    }
  };

  int64_t const n = a.nrows, T = static_cast<int64_t>(B) * R, step = static_cast<int64_t>(gridDim.x) * T;
  uint32_t const smask = static_cast<uint32_t>(slots - 1);
  if constexpr (SRC == DCOL_SIMPLE) {
    uint64_t const lo = a.map.lo, range = a.map.range;
    uint64_t const* kbase = p.simple_base[0];
    uint64_t const* vbase[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) vbase[j] = p.simple_base[1 + j];
    constexpr int D = 2;
    uint64_t pk[D][R], pv[D][R][NV];
    auto issue = [&](int64_t tile, uint64_t (&kk)[R], uint64_t (&vv)[R][NV]) {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        int64_t const row = min(tile + static_cast<int64_t>(k) * B + threadIdx.x, n - 1);  // (a row past the end reads the last row)
        kk[k]             = gload(kbase + row);
#pragma unroll
        for (int j = 0; j < NV; ++j) vv[k][j] = gload(vbase[j] + row);
      }
    };
    int64_t const begin = static_cast<int64_t>(blockIdx.x) * T;
#pragma unroll
    for (int j = 0; j < D; ++j) issue(begin + j * step, pk[j], pv[j]);
    for (int64_t tile = begin; tile < n; tile += D * step) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        int64_t const t0 = tile + j * step;
        bool act[R];
        uint32_t s[R], valid[R];
        uint64_t val[R][NV];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          act[k]             = t0 + static_cast<int64_t>(k) * B + threadIdx.x < n;
          uint64_t const idx = pk[j][k] - lo;
          bad                = bad || (act[k] && idx >= range);  // the sampled key range was wrong: this call is void
          s[k]               = static_cast<uint32_t>(idx) & smask;
          valid[k]           = 1;
#pragma unroll
          for (int c = 0; c < NV; ++c) val[k][c] = pv[j][k][c];
        }
        issue(t0 + D * step, pk[j], pv[j]);
        if (t0 >= n) break;  // (uniform)
        if (bad) atomicOr(a.overflow, 4);
        accumulate_tile(act, s, valid, val);
      }
    }
  } else {
    dense_local const L = make_dense_local(p, a.map, a.ones);
    dense_raw_tile<R> raw;
    int64_t const begin = static_cast<int64_t>(blockIdx.x) * T;
    issue_dense_local<R>(L, begin, B, n, raw);
    for (int64_t t0 = begin; t0 < n; t0 += step) {
      bool act[R];
      uint32_t s[R], valid[R];
      uint64_t val1[R], val[R][NV];
      bool bad = false;
      decode_dense_local<R>(p, a.map, L, t0, B, n, raw, act, s, valid, val1, bad);
      issue_dense_local<R>(L, t0 + step, B, n, raw);
      if (bad) atomicOr(a.overflow, 4);
#pragma unroll
      for (int k = 0; k < R; ++k) {
        s[k] &= smask;
        val[k][0] = val1[k];
      }
      accumulate_tile(act, s, valid, val);
    }
  }
  __syncthreads();
  uint64_t* image = a.tables + static_cast<int64_t>(blockIdx.x) * (a.image_bytes / 8);
  for (int i = threadIdx.x; i < a.image_bytes / 16; i += B)
    gstore(reinterpret_cast<u64x2*>(image) + i, reinterpret_cast<u64x2 const*>(lds_raw)[i]);
}

template <uint64_t SIG, int NACCT, int SRC, int NV = 1>
void launch_dcol_t(dense_agg_args const& a, dense_agg_args const* d_args, hipStream_t stream)
{
  static std::once_flag attr_once;  // (the API is re-entrant across objects: two threads may launch this kernel first)
  std::call_once(attr_once, [] { allow_full_lds(reinterpret_cast<void const*>(&k_aggregate_dense_columns<SIG, NACCT, SRC, NV>)); });
  cudf::detail::prof::scope prof_{"aggregate", stream};
  hipLaunchKernelGGL((k_aggregate_dense_columns<SIG, NACCT, SRC, NV>), dim3(a.nsplit), dim3(1024), a.image_bytes, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}
template <uint64_t SIG, int NACCT>
void launch_dcol_n(dense_agg_args const& a, dense_agg_args const* d_args, hipStream_t stream)
{
  if (a.map.nkeys > 0) return launch_dcol_t<SIG, NACCT, DCOL_COLS>(a, d_args, stream);
  return launch_dcol_t<SIG, NACCT, DCOL_SIMPLE>(a, d_args, stream);
}

}  // namespace

void launch_aggregate_dense_columns(dense_agg_args const& a, dense_agg_args const* d_args, hipStream_t stream)
{
  CUDF_EXPECTS(a.plan.narg == 0 && (a.map.nkeys > 0 || (a.plan.simple && a.plan.KU == 1 && a.plan.NPAY >= 1 && a.plan.NPAY <= RING_MAX_VALUES)),
               "dense keys: one plain key column and one to three plain value columns, or composite integer keys and one value column");
  CUDF_EXPECTS(a.nitems == 1 && a.nsplit >= 1 && a.nrows >= 1 && a.image_bytes <= 159 * 1024 && (a.slots & (a.slots - 1)) == 0 &&
                 a.map.log2P == 0 && (uint64_t{1} << a.map.bits) == static_cast<uint64_t>(a.slots) && a.map.range <= static_cast<uint64_t>(a.slots) &&
                 a.occ_acc == dense_occ_acc(a.plan) && a.image_bytes == static_cast<int32_t>(dense_table_bytes(a.plan, a.slots)) &&
                 (a.map.nkeys == 0 || a.ones != nullptr),
               "dense keys: one table for the whole key range");
  uint64_t const sig = plan_sig(a.plan);
  if (a.map.nkeys == 0 && a.plan.NPAY == 2) {  // two value columns: the common SUM / MEAN plans, then the descriptor-reading form
    if (sig == SIG_SUMF_CNT_SUMF) return launch_dcol_t<SIG_SUMF_CNT_SUMF, 4, DCOL_SIMPLE, 2>(a, d_args, stream);
    if (sig == SIG_SUMF_SUMF) return launch_dcol_t<SIG_SUMF_SUMF, 2, DCOL_SIMPLE, 2>(a, d_args, stream);
    if (a.plan.NACC <= 4) return launch_dcol_t<0, 4, DCOL_SIMPLE, 2>(a, d_args, stream);
    return launch_dcol_t<0, MAX_ACC, DCOL_SIMPLE, 2>(a, d_args, stream);
  }
  if (a.map.nkeys == 0 && a.plan.NPAY == 3) {
    if (sig == SIG_SUMF_CNT_SUMF_SUMF) return launch_dcol_t<SIG_SUMF_CNT_SUMF_SUMF, 4, DCOL_SIMPLE, 3>(a, d_args, stream);
    if (sig == SIG_SUMF_SUMF_SUMF) return launch_dcol_t<SIG_SUMF_SUMF_SUMF, 4, DCOL_SIMPLE, 3>(a, d_args, stream);
    if (a.plan.NACC <= 4) return launch_dcol_t<0, 4, DCOL_SIMPLE, 3>(a, d_args, stream);
    return launch_dcol_t<0, MAX_ACC, DCOL_SIMPLE, 3>(a, d_args, stream);
  }
  if (sig == SIG_SUMF_CNT) return launch_dcol_n<SIG_SUMF_CNT, 2>(a, d_args, stream);
  if (sig == SIG_SUMI_CNT) return launch_dcol_n<SIG_SUMI_CNT, 2>(a, d_args, stream);
  if (sig == SIG_MEAN_MIN_MAX_F) return launch_dcol_n<SIG_MEAN_MIN_MAX_F, 4>(a, d_args, stream);
  if (sig == SIG_MEAN_MIN_MAX_F_NULLS) return launch_dcol_n<SIG_MEAN_MIN_MAX_F_NULLS, 4>(a, d_args, stream);
  if (sig == SIG_SUMF_CNT_NULLS) return launch_dcol_n<SIG_SUMF_CNT_NULLS, 2>(a, d_args, stream);
  if (sig == SIG_SUMI_CNT_NULLS) return launch_dcol_n<SIG_SUMI_CNT_NULLS, 2>(a, d_args, stream);
  if (a.plan.NACC <= 4) return launch_dcol_n<0, 4>(a, d_args, stream);
  return launch_dcol_n<0, MAX_ACC>(a, d_args, stream);
}

}  // namespace cudf::groupby::detail
