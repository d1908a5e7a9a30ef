// SPDX-License-Identifier: Apache-2.0
// gfx950 kernels of the dense-key hash-groupby path for SEVERAL value columns (engine.hpp ring_multi_args,
// dense_multi_merge_args): the ring scatter of dense_ring_kernels.hip with one value stream per column next to the shared 16-bit
// tag stream, and the fold of the per-column table images into partial records.
//
// df.groupby(k).agg({a: sum, b: sum}) used to fall off the dense path (24- / 32-byte hash records: 21.5 / 35.2 ms per 1B rows
// on 1M groups against 6.7 ms for one column, profiles/r2_multi_value.txt). Here the input is still read ONCE: every partition
// owns one ring per value column and one tag ring in LDS; a row reserves ONE position (one returning ds_add) that holds in all
// of them; owner lanes flush whole aligned 128-byte granules of every stream. 160 KiB of LDS carry 128 partitions x
// (nval x CAP x 8 + 128 x 2) bytes: CAP = 48 records for two columns, 32 for three (one column: 64), so the tile shrinks with
// the ring - 2048 rows for two columns, 1024 for three - and loads stay two / four tiles ahead.
// The reference does all (column, aggregation) pairs in one pass too (cpp/src/groupby/hash/compute_global_memory_aggs.cuh:139-147),
// with one global atomic per pair and row.
#include "device_common.hpp"
#include "../common/ring_scatter.hpp"

namespace cudf::groupby::detail {
namespace {

// (DENSE: the key is loaded beside the NV streamed value columns; hash keys: the key IS stream 0)
template <int NV, bool DENSE, int RPT>
struct multi_tile {
  uint64_t k[DENSE ? RPT : 1];
  uint64_t v[NV][RPT];
};

// NV: 8-byte streams written. DENSE: dense keys - the streams are the value columns, the partition digit and the 16-bit tag come
// from the dense map. Otherwise (sparse single 8-byte keys, hash_ring path of paths_hash.cpp): stream 0 is the KEY column, the
// others the value columns, the digit is the top bits of the key hash and there are no tags - the aggregate (k_aggregate_k64)
// hashes the key again. CAP: records per partition ring (a multiple of 16, not necessarily a power of two: positions are taken
// modulo CAP).
template <int NV, bool DENSE, int CAP, int RPT, int D>
__global__ void __launch_bounds__(1024) k_dense_ring_scatter_multi(ring_multi_args const* __restrict__ ap)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int s_pending, s_abort;
  ring_multi_args const& a = *ap;
  plan_dev const& p        = a.plan;
  constexpr int B          = 1024;
  constexpr uint32_t G = 16, GT = 64, TPL = 8, TCAP = 128;  // records per value granule / per tag granule (both 128 bytes); tag ring length
  static_assert(CAP % 16 == 0 && CAP + GT - G <= TCAP, "a tag waits for its granule of 64 while its value may already have left");
  int const P = a.P;
  uint64_t* rval  = reinterpret_cast<uint64_t*>(lds_raw);                         // [NV][P * CAP]
  uint16_t* rtag  = reinterpret_cast<uint16_t*>(rval + static_cast<uint32_t>(NV) * P * CAP);  // [P * TCAP] (DENSE)
  uint32_t* tail  = reinterpret_cast<uint32_t*>(rtag + (DENSE ? static_cast<uint32_t>(P) * TCAP : 0u));  // [P] next virtual position
  uint32_t* limit = tail + P;                                                     // [P] head + CAP as of the last flush
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = B >> 6;
  int const PW = P / nwaves;  // partitions owned by a wave (1 ... 16): owner lane l < PW holds partition wave * PW + l
  uint32_t head = 0, headt = 0;  // owner lanes: values / tags of the partition flushed so far
  for (int d = threadIdx.x; d < P; d += B) {
    tail[d]  = 0;
    limit[d] = CAP;
  }
  if (threadIdx.x == 0) {
    s_pending = 0;
    s_abort   = 0;
  }
  __syncthreads();
  constexpr int64_t T = static_cast<int64_t>(B) * RPT;
  int const item      = blockIdx.x;
  // workgroup w takes the row tiles w, w + slices, ...: sorted or clustered keys spread over every workgroup's regions
  int64_t const begin = static_cast<int64_t>(item) * T, end = a.nrows, step = static_cast<int64_t>(a.slices) * T;
  uint32_t const region_cap = static_cast<uint32_t>(a.region_cap);
  uint64_t* const out_val   = a.out_val;
  uint16_t* const out_tag   = a.out_tag;
  int64_t const sstride     = a.stream_stride;
  [[maybe_unused]] uint64_t const dense_lo = a.map.lo, dense_range = a.map.range;
  [[maybe_unused]] uint32_t const mult = a.map.mult, bmask = DENSE ? (1u << a.map.bits) - 1u : 0u;
  int const shift = a.shift;
  [[maybe_unused]] uint32_t const pmask = static_cast<uint32_t>(P - 1), tmask = DENSE ? (1u << shift) - 1u : 0u;
  [[maybe_unused]] uint64_t const* kbase = p.simple_base[0];
  [[maybe_unused]] uint64_t const kmask0 = p.key_mask[0];
  uint64_t const* vbase[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) vbase[j] = p.simple_base[(DENSE ? 1 : 0) + j];

  auto issue = [&](int64_t tile, multi_tile<NV, DENSE, RPT>& r) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      int64_t const row = tile + static_cast<int64_t>(k) * B + threadIdx.x;
      if (row < end) {
        if constexpr (DENSE) r.k[k] = gload(kbase + row);
#pragma unroll
        for (int j = 0; j < NV; ++j) r.v[j][k] = gload(vbase[j] + row);
      }
    }
  };
  // flush every complete granule of this wave's partitions (common/ring_scatter.hpp); final: also the partial last granule
  auto flush = [&](bool final) {
    int const dmine = wave * PW + lane;
    auto const f    = cudf::detail::ring::plan_flush<G, GT>(tail, lane < PW, dmine, head, headt, CAP, region_cap, final, s_abort);
    auto const rbase_of = [&](int d) { return (static_cast<int64_t>(d) * a.slices + item) * a.region_cap; };
    // values: one stream after the other (pos is even and CAP is even: a lane's pair does not wrap)
#pragma unroll
    for (int j = 0; j < NV; ++j)
      cudf::detail::ring::flush_stream(rval + static_cast<uint32_t>(j) * P * CAP, out_val + static_cast<int64_t>(j) * sstride, wave, PW, lane, f.nrec,
                                       head, f.ab, [&](uint32_t d, uint32_t pos) { return d * CAP + pos % CAP; }, rbase_of);
    if constexpr (DENSE)
      cudf::detail::ring::flush_stream(rtag, out_tag, wave, PW, lane, f.nrect, headt, f.ab,
                                       [&](uint32_t d, uint32_t pos) { return d * TCAP + (pos & (TCAP - 1u)); }, rbase_of);
    cudf::detail::ring::commit_flush(limit, lane < PW, dmine, head, headt, f, CAP);
  };
  auto put = [&](uint32_t d, uint32_t pos, uint32_t tg, uint64_t const (&val)[NV]) {
    uint32_t const w = d * CAP + pos % CAP;
#pragma unroll
    for (int j = 0; j < NV; ++j) rval[static_cast<uint32_t>(j) * P * CAP + w] = val[j];
    if constexpr (DENSE) rtag[d * TCAP + (pos & (TCAP - 1u))] = static_cast<uint16_t>(tg);
  };

  multi_tile<NV, DENSE, RPT> pre[D];
#pragma unroll
  for (int j = 0; j < D; ++j) issue(begin + j * step, pre[j]);
  for (int64_t tile = begin; tile < end; tile += D * step) {
#pragma unroll
    for (int jt = 0; jt < D; ++jt) {
      int64_t const t0 = tile + jt * step;
      // ---- this tile's rows: partition digit, tag and values (waits for the tile's loads)
      bool keep[RPT];
      uint32_t d[RPT], tg[RPT];
      uint64_t val[RPT][NV];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        keep[k] = t0 + static_cast<int64_t>(k) * B + threadIdx.x < end;
        d[k]    = 0;
        tg[k]   = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) val[k][j] = pre[jt].v[j][k];
        if (keep[k]) {
          if constexpr (DENSE) {
            uint64_t idx = pre[jt].k[k] - dense_lo;
            if (idx >= dense_range) {  // the sampled key range was wrong: this call is void (redone by hash)
              atomicOr(a.overflow, 4);
              idx = 0;
            }
            uint32_t const x = (static_cast<uint32_t>(idx) * mult) & bmask;
            d[k]  = (x >> shift) & pmask;
            tg[k] = (x & tmask) | (1u << 15);
          } else {  // the engine's key hash (device_common.hpp hash_key_units), its top bits
            uint64_t const h = mix64(0x9e3779b97f4a7c15ull ^ (val[k][0] & kmask0));
            d[k]             = static_cast<uint32_t>(h >> shift) & pmask;
          }
        }
      }
      issue(t0 + D * step, pre[jt]);
      if (t0 >= end) break;  // (uniform)
      // ---- reserve ring positions, write the records, flush; rows whose position lies beyond the ring wait for the flush
      cudf::detail::ring::place_tile<RPT>(tail, limit, keep, d, s_pending, s_abort, [&](int k, uint32_t pos) { put(d[k], pos, tg[k], val[k]); },
                                          [&] { flush(false); }, [] {}, [] {});
      if (s_abort) {  // a region would overflow (skewed or clustered keys): the caller redoes the call
        if (threadIdx.x == 0) atomicOr(a.overflow, 1);
        return;
      }
    }
  }
  flush(true);
  lds_barrier();
  if (s_abort) {
    if (threadIdx.x == 0) atomicOr(a.overflow, 1);
    return;
  }
  if (lane < PW) a.region_count[static_cast<int64_t>(wave * PW + lane) * a.slices + item] = static_cast<int32_t>(head);
}

// ------------------------------------------------------------------ K_dense_merge_dump_multi
// The table images of every value column (left by k_aggregate_dense, one launch per column) -> partial records of the whole
// plan. Work item (d, j): slots [j, j + 1) * slots / dsplit of partition d; one thread per slot.
__global__ void __launch_bounds__(1024) k_dense_merge_dump_multi(dense_multi_merge_args const* __restrict__ ap, int dsplit)
{
  __shared__ uint32_t s_dump;
  dense_multi_merge_args const& a = *ap;
  plan_dev const& p               = a.plan;
  int const NACC = p.NACC, slots = a.slots, nsplit = a.nsplit, B = blockDim.x;
  int const d = blockIdx.x / dsplit, per = slots / dsplit, s0 = (static_cast<int>(blockIdx.x) % dsplit) * per;
  if (threadIdx.x == 0) s_dump = 0;
  __syncthreads();
  int const PU        = 1 + NACC;
  uint64_t* out       = a.out_records + static_cast<int64_t>(blockIdx.x) * per * PU;
  uint32_t const hi   = static_cast<uint32_t>(d) << (a.map.bits - a.map.log2P);
  uint32_t const bmask = (1u << a.map.bits) - 1u;
  for (int s = s0 + threadIdx.x; s < s0 + per; s += B) {
    uint64_t acc[MAX_ACC];
    bool occupied = false;
    for (int q = 0; q < NACC; ++q) {
      int const op = p.acc[q].op, c = a.acc_col[q];
      bool const narrow = a.acc_narrow[q] != 0;
      unsigned char const* images = reinterpret_cast<unsigned char const*>(a.tables[c]) + static_cast<int64_t>(d) * nsplit * a.image_bytes[c];
      uint64_t v = narrow ? 0 : acc_identity(op);
      for (int h = 0; h < nsplit; ++h) {
        unsigned char const* img = images + static_cast<int64_t>(h) * a.image_bytes[c] + a.acc_off[q];
        if (narrow) v += gload(reinterpret_cast<uint32_t const*>(img) + s);
        else v = combine_values(op, v, gload(reinterpret_cast<uint64_t const*>(img) + s));
      }
      acc[q] = v;
      if (q == a.occ_acc) occupied = v != 0;
    }
    if (a.occ_acc < 0) {
      unsigned char const* images = reinterpret_cast<unsigned char const*>(a.tables[0]) + static_cast<int64_t>(d) * nsplit * a.image_bytes[0];
      for (int h = 0; h < nsplit; ++h)
        occupied = occupied || ((gload(reinterpret_cast<uint32_t const*>(images + static_cast<int64_t>(h) * a.image_bytes[0] + a.occ_off0) + (s >> 5)) >> (s & 31)) & 1u);
    }
    if (!occupied) continue;
    uint32_t const pos = atomicAdd(&s_dump, 1u);
    uint64_t* o        = out + static_cast<int64_t>(pos) * PU;
    gstore(o, a.map.lo + (((hi | static_cast<uint32_t>(s)) * a.map.mult_inv) & bmask));
    for (int q = 0; q < NACC; ++q) gstore(o + 1 + q, acc[q]);
  }
  __syncthreads();
  if (threadIdx.x == 0) a.out_count[blockIdx.x] = static_cast<int32_t>(s_dump);
}

template <int NV, bool DENSE, int CAP, int RPT, int D>
void launch_multi_t(ring_multi_args const& a, ring_multi_args const* d_args, hipStream_t stream)
{
  static std::once_flag attr_once;
  std::call_once(attr_once, [] { allow_full_lds(reinterpret_cast<void const*>(&k_dense_ring_scatter_multi<NV, DENSE, CAP, RPT, D>)); });
  static_assert(DENSE, "the sparse-key (hash) form of this scatter was measured slower and removed (profiles/r3_sparse_ring.txt)");
  std::size_t const lds = dense_ring_multi_lds_bytes(NV, a.P, CAP);
  cudf::detail::prof::scope prof_{"partition_scatter", stream};
  hipLaunchKernelGGL((k_dense_ring_scatter_multi<NV, DENSE, CAP, RPT, D>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

}  // namespace

std::size_t dense_ring_multi_lds_bytes(int nval, int P, int cap)
{
  return static_cast<std::size_t>(P) * (static_cast<std::size_t>(nval) * cap * 8 + 128 * 2 + 8);
}
int dense_ring_multi_cap(int nval) { return nval <= 2 ? 48 : 32; }

void store_args(ring_multi_args const& a, ring_multi_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<ring_multi_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}
void store_args(dense_multi_merge_args const& a, dense_multi_merge_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<dense_multi_merge_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_dense_ring_scatter_multi(ring_multi_args const& a, ring_multi_args const* d_args, hipStream_t stream)
{
  CUDF_EXPECTS(a.plan.simple && a.plan.KU == 1 && a.plan.NPAY == a.nval && (a.nval == 2 || a.nval == 3),
               "ring scatter: one plain key column and two or three plain value columns");
  CUDF_EXPECTS(a.P >= 16 && a.P <= 256 && (a.P & (a.P - 1)) == 0 && a.cap == dense_ring_multi_cap(a.nval) &&
                 dense_ring_multi_lds_bytes(a.nval, a.P, a.cap) + 64 <= 160 * 1024,
               "ring scatter: fan-out 16 ... 256 within the LDS");
  CUDF_EXPECTS(a.region_cap % 64 == 0 && a.shift >= 0 && a.shift <= 15 && a.slices >= 1 && a.stream_stride % 16 == 0 && a.nrows >= 1,
               "ring scatter: region geometry");
  if (a.nval == 2) return launch_multi_t<2, true, 48, 2, 2>(a, d_args, stream);
  return launch_multi_t<3, true, 32, 1, 4>(a, d_args, stream);
}

void launch_dense_merge_dump_multi(dense_multi_merge_args const& a, dense_multi_merge_args const* d_args, int dsplit, hipStream_t stream)
{
  CUDF_EXPECTS(a.ncols >= 2 && a.ncols <= RING_MAX_VALUES && a.nsplit >= 1 && dsplit >= 1 && a.slots % dsplit == 0 && a.map.nkeys == 0,
               "dense keys, several value columns: merge geometry");
  cudf::detail::prof::scope prof_{"aggregate_merge", stream};
  hipLaunchKernelGGL(k_dense_merge_dump_multi, dim3(a.nitems * dsplit), dim3(1024), 0, stream, d_args, dsplit);
  CUDF_HIP_TRY(hipGetLastError());
}

}  // namespace cudf::groupby::detail
