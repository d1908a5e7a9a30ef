// SPDX-License-Identifier: Apache-2.0
// gfx950 kernel of the LINE scatter of the dense-key hash-groupby path (engine.hpp dense_line_args): the one-level ring scatter of
// dense_ring_kernels.hip for the plain shape - one plain 8-byte integer key, one plain 8-byte value, at most 2^20 indices, no heavy
// hitters - at a fixed fan-out of 256 (one direct-address table per CU afterwards: no table images, no merge).
//
// What differs from the ring scatter it came from (prototype and measurements: bench_micro/tail_groupby_micro.hip,
// profiles/r4_c2_line_path.txt; on one box 4.49 ms against 5.13 ms for 1B rows):
//  * ONE stream: a record leaves inside a 128-byte LINE of 12 records, [12 x value | 12 x 16-bit tag | 8 spare bytes] (10.67 bytes
//    per row). The ring of a partition is three such lines in LDS, already in their memory layout; a flush is a straight copy of
//    whole lines, 8 lanes x 16 bytes, written through (sc1) - nothing lingers in the XCD's L2 for a later write-back burst.
//    One ring instead of a value ring plus a four times longer tag ring: 98 KB of LDS for 256 partitions.
//  * a row that finds its ring full is placed right after the tile's flush, behind the same barrier (the owner lane raises a flag
//    only when a row STILL does not fit): no extra round of four barriers for the quarter of the tiles that overflow a ring.
//  * the tile's loads are buffer loads from a scalar base (rows past the end read as 0: no per-row address arithmetic or clamp);
//    the line stores are inline asm the compiler does not track - a store loop of variable length made it wait vmcnt(0) at every
//    use of a loaded register.
// Together with dense_kernels.hip (k_aggregate_dense, FMT_LINES) this replaces the reference's global hash-set insert + global atomics
// (cpp/src/groupby/hash/compute_global_memory_aggs.cuh:74-187, single_pass_functors.cuh:86-157).
#include "device_common.hpp"

namespace cudf::groupby::detail {
namespace {

constexpr int LN_P      = DENSE_LINE_FANOUT;   // partitions
constexpr int LN_RPL    = DENSE_LINE_RECORDS;  // records per line
constexpr int LN_RLINES = 3;                   // ring lines per partition
constexpr int LN_RCAP   = LN_RPL * LN_RLINES;

__device__ __forceinline__ uint32_t div12(uint32_t x) { return __umulhi(x, 0xAAAAAAABu) >> 3; }
__device__ __forceinline__ uint32_t mod3(uint32_t x) { return x - 3u * (__umulhi(x, 0xAAAAAAABu) >> 1); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void const* p, uint32_t bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, static_cast<int>(bytes), 0x00020000);
}
__device__ __forceinline__ void store16_sc1(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

template <int RPT, int D>
__global__ void __launch_bounds__(1024) k_dense_line_scatter(dense_line_args const* __restrict__ ap)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int s_pending[2], s_soft[2], s_abort;
  dense_line_args const& a = *ap;
  constexpr int B = 1024;
  unsigned char* ring = lds_raw;                                                          // LN_P * 384
  uint32_t* tail      = reinterpret_cast<uint32_t*>(lds_raw + LN_P * LN_RLINES * 128);    // [LN_P] next virtual position
  uint32_t* limit     = tail + LN_P;                                                      // [LN_P] first position beyond the ring
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = threadIdx.x & 7;
  int const me = blockIdx.x;
  for (int i = threadIdx.x; i < LN_P; i += B) {
    tail[i]  = 0;
    limit[i] = LN_RCAP;
  }
  if (threadIdx.x < 2) {
    s_pending[threadIdx.x] = 0;
    s_soft[threadIdx.x]    = 0;
  }
  if (threadIdx.x == 0) s_abort = 0;
  __syncthreads();
  constexpr int64_t T = static_cast<int64_t>(B) * RPT;
  int64_t const step = static_cast<int64_t>(a.slices) * T, end = a.nrows;
  uint64_t const* const kbase = a.plan.simple_base[0];
  uint64_t const* const vbase = a.plan.simple_base[1];
  uint64_t pk[D][RPT], pv[D][RPT];
  uint32_t const voff = threadIdx.x * 8u;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  auto issue = [&](int64_t tile, uint64_t (&kk)[RPT], uint64_t (&vv)[RPT]) {
    int64_t left       = end - tile;
    int64_t const base = left > 0 ? tile : 0;
    left               = left > 0 ? left : 0;
    uint32_t const bytes = left > 0x1fffffff ? 0xfffffff8u : static_cast<uint32_t>(left) * 8u;
    __amdgpu_buffer_rsrc_t const rk = make_rsrc(kbase + base, bytes), rv = make_rsrc(vbase + base, bytes);
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      kk[k] = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(rk, static_cast<int>(voff), k * B * 8, 0));
      vv[k] = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(rv, static_cast<int>(voff), k * B * 8, 0));
    }
  };
  uint32_t head = 0;  // owner lanes (lane < 16): lines of partition wave * 16 + lane flushed
  uint32_t const region_lines = static_cast<uint32_t>(a.region_lines);
  uint64_t const dense_lo = a.map.lo, dense_range = a.map.range;
  uint32_t const mult = a.map.mult, bmask = (1u << a.map.bits) - 1u;
  int const shift = a.map.bits - 8;  // partition = the top 8 bits of the scrambled index, tag = the bits below | valid
  uint32_t const tmask = (1u << shift) - 1u;
  bool bad_key = false;

  // flush every complete line of this wave's partitions
  auto flush = [&](int ph) {
    uint32_t nl = 0;
    int const dmine = wave * 16 + lane;
    if (lane < 16) {
      uint32_t const t = tail[dmine], limv = head * LN_RPL + LN_RCAP;
      uint32_t const c = static_cast<int32_t>(t - limv) < 0 ? t : limv;
      nl = div12(c) - head;
      if (head + nl > region_lines) {  // the region would overflow (skewed or clustered keys): the caller redoes the call
        s_abort = 1;
        nl      = 0;
      }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      int const pl = b * 8 + (lane >> 3);
      uint32_t const mn = __shfl(nl, pl), mh = __shfl(head, pl);
      uint32_t const d  = static_cast<uint32_t>(wave * 16 + pl);
      // (source-major: the 256 cells a workgroup writes at one time lie region_lines * 128 bytes apart, region_lines odd - with the
      // partition-major order of the ring scatter they lay a multiple of 2^15 bytes apart: the same few HBM channels, 5.3 against 4.5 ms)
      unsigned char* const region = a.out_lines + (static_cast<uint64_t>(me) * LN_P + d) * region_lines * 128u;
      for (uint32_t g = 0;; ++g) {
        bool const act = g < mn;
        if (__ballot(act) == 0) break;
        if (act) {
          uint32_t const L = mh + g;
          u32x4 const v    = *reinterpret_cast<u32x4 const*>(ring + d * (LN_RLINES * 128) + mod3(L) * 128 + sub * 16);
          store16_sc1(region + static_cast<uint64_t>(L) * 128u + sub * 16, v);
        }
      }
    }
    if (lane < 16) {
      head += nl;
      uint32_t const newlim = head * LN_RPL + LN_RCAP;
      limit[dmine] = newlim;
      if (static_cast<int32_t>(tail[dmine] - newlim) > 0) s_pending[ph] = 1;  // rows that still do not fit: another round
    }
  };

  int ph = 0;  // parity of the place attempt (which s_pending / s_soft word it raises)
#pragma unroll
  for (int j = 0; j < D; ++j) issue(static_cast<int64_t>(me) * T + j * step, pk[j], pv[j]);
  for (int64_t tile = static_cast<int64_t>(me) * T; tile < end; tile += D * step) {
#pragma unroll
    for (int j = 0; j < D; ++j) {  // (a part past the end runs empty: no break inside the unrolled pair - it cost register copies)
      int64_t const t0     = tile + j * step;
      int64_t const left64 = end - t0;
      uint32_t const rows_left = left64 <= 0 ? 0u : (left64 > 0x7fffffff ? 0x7fffffffu : static_cast<uint32_t>(left64));
      bool pend[RPT];
      uint32_t d[RPT], tg[RPT], pos[RPT];
      uint64_t val[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        pend[k] = static_cast<uint32_t>(k * B) + threadIdx.x < rows_left;  // (until placed)
        uint64_t idx = pk[j][k] - dense_lo;
        if (idx >= dense_range) {  // the sampled key range was wrong: this call is void (redone by hash); no atomic here - a
          bad_key = bad_key || pend[k];  // vector-memory operation that may or may not issue costs every later wait its count
          idx     = 0;
        }
        uint32_t const x = (static_cast<uint32_t>(idx) * mult) & bmask;
        d[k]   = x >> shift;
        tg[k]  = (x & tmask) | 0x8000u;
        val[k] = pv[j][k];
      }
      issue(t0 + D * step, pk[j], pv[j]);
      auto place = [&](int k) {
        uint32_t const q = div12(pos[k]), r = pos[k] - q * LN_RPL;
        unsigned char* line = ring + d[k] * (LN_RLINES * 128) + mod3(q) * 128;
        *reinterpret_cast<uint64_t*>(line + r * 8)       = val[k];
        *reinterpret_cast<uint16_t*>(line + 96 + r * 2) = static_cast<uint16_t>(tg[k]);
      };
      // reserve ring positions; rows whose position lies beyond the ring wait for the flush
      uint32_t lim[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        pos[k] = 0;
        lim[k] = 0;
        if (pend[k]) {
          pos[k] = atomicAdd(&tail[d[k]], 1u);
          lim[k] = limit[d[k]];
        }
      }
      bool waits = false;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        if (pend[k] && static_cast<int32_t>(pos[k] - lim[k]) < 0) {
          place(k);
          pend[k] = false;
        }
        waits = waits || pend[k];
      }
      if (waits) s_soft[ph] = 1;
      lds_barrier();
      if (threadIdx.x == 0) {
        s_pending[ph ^ 1] = 0;
        s_soft[ph ^ 1]    = 0;
      }
      flush(ph);
      lds_barrier();
      // rows that found their ring full: its lines have left by now (the owner raised s_pending[ph] if some row still does not fit)
      if (s_soft[ph]) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          if (pend[k] && static_cast<int32_t>(pos[k] - limit[d[k]]) < 0) {
            place(k);
            pend[k] = false;
          }
        }
      }
      while (s_pending[ph] && !s_abort) {  // (rare: a partition took more rows of one tile than a ring holds)
        ph ^= 1;
        lds_barrier();
        if (threadIdx.x == 0) {
          s_pending[ph ^ 1] = 0;
          s_soft[ph ^ 1]    = 0;
        }
        flush(ph);
        lds_barrier();
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          if (pend[k] && static_cast<int32_t>(pos[k] - limit[d[k]]) < 0) {
            place(k);
            pend[k] = false;
          }
        }
      }
      ph ^= 1;
    }
    if (s_abort) break;
  }
  if (bad_key) atomicOr(a.overflow, 4);
  // pad the partial lines (a padding record carries DENSE_LINE_PAD_TAG: the aggregate skips it), flush them, leave the line counts
  lds_barrier();
  if (lane < 16) {
    int const dmine = wave * 16 + lane;
    uint32_t const t = tail[dmine], q = div12(t), r = t - q * LN_RPL;
    if (r != 0) {
      unsigned char* line = ring + dmine * (LN_RLINES * 128) + mod3(q) * 128;
      for (uint32_t e = r; e < LN_RPL; ++e) *reinterpret_cast<uint16_t*>(line + 96 + e * 2) = static_cast<uint16_t>(DENSE_LINE_PAD_TAG);
      tail[dmine] = t + LN_RPL - r;
    }
  }
  lds_barrier();
  flush(0);
  lds_barrier();
  if (s_abort) {
    if (threadIdx.x == 0) atomicOr(a.overflow, 1);
    return;
  }
  if (lane < 16) a.region_count[static_cast<int64_t>(wave * 16 + lane) * a.slices + me] = static_cast<int32_t>(head);
}

}  // namespace

void store_args(dense_line_args const& a, dense_line_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<dense_line_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_dense_line_scatter(dense_line_args const& a, dense_line_args const* d_args, hipStream_t stream)
{
  CUDF_EXPECTS(a.plan.simple && a.plan.KU == 1 && a.plan.NPAY == 1 && a.map.nkeys == 0, "line scatter: one plain key column and one plain value column");
  CUDF_EXPECTS(a.map.bits >= 8 + 6 && a.map.bits <= 8 + 15 && a.map.log2P == 8 && a.slices >= 1 && a.region_lines >= 1 && a.region_lines < 65536 && (a.region_lines & 1) == 1,
               "line scatter: 256 partitions of 64 ... 32768 table slots, regions of an odd number of lines below 65536");
  std::size_t const lds = static_cast<std::size_t>(LN_P) * LN_RLINES * 128 + 2 * LN_P * 4;
  cudf::detail::prof::scope prof_{"partition_scatter", stream};
  // (3 rows per thread: 121 registers, nothing spilled; 4 rows per thread spill 7 and, with a mean of 16 rows per partition and tile,
  // almost every tile has a row that waits for the flush)
  static std::once_flag attr_once;
  std::call_once(attr_once, [] { allow_full_lds(reinterpret_cast<void const*>(&k_dense_line_scatter<3, 2>)); });
  hipLaunchKernelGGL((k_dense_line_scatter<3, 2>), dim3(a.slices), dim3(1024), lds, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

}  // namespace cudf::groupby::detail
