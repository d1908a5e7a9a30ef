// SPDX-License-Identifier: Apache-2.0
// gfx950 kernel of the RING scatter of the dense-key hash-groupby path (engine.hpp dense_ring_args; measurements:
// profiles/r2_ring_scatter_microbench.txt, prototype bench_micro/ring_scatter_micro.hip).
//
// Every partition owns a ring of CAP record slots in LDS. A row reserves the next virtual position of its partition with ONE
// returning LDS atomic and writes its record there; after a barrier the owner lanes flush every COMPLETE granule of 16 records
// to the partition's region (global record index = region base + virtual position: nothing ever moves inside LDS); a second
// barrier ends the tile. A row that finds its ring full waits one flush round. Two barriers and ~4 LDS operations per row,
// against seven barriers and ~8 operations for the rank / scan / stage / write-out / carry-move scatter of
// common/wc_scatter.hpp. Records leave as two streams - granules of 16 values and of 64 16-bit tags (32 32-bit tags on the
// first of two levels), both one whole aligned 128-byte line (the tag ring is four / two times as long as the value ring so that a
// partition's tags can wait for their granule) - so a row costs 10 bytes in the record buffer instead of 16 (12 on the first of
// two levels): the scatter writes, and the aggregate reads, three eighths less.
// Together with dense_kernels.hip this replaces the reference's global hash-set insert + global atomics
// (cpp/src/groupby/hash/compute_global_memory_aggs.cuh:74-187, single_pass_functors.cuh:86-157).
#include "device_common.hpp"
#include "dense_loader.hpp"
#include "../common/ring_scatter.hpp"

namespace cudf::groupby::detail {
namespace {

constexpr int STATIC_TILES_AHEAD = 2;  // tiles of loads in flight with a compile-time column shape (the run-time loader: 1)
constexpr int RING_SRC_SIMPLE  = 0;  // one plain 8-byte integer key column, one plain 8-byte value column
constexpr int RING_SRC_COLS    = 1;  // composite dense keys (dense_loader.hpp)
constexpr int RING_SRC_REGIONS = 2;  // level 2: the regions of a level-1 partition
constexpr int RING_SRC_COLS_STATIC = 3;  // composite dense keys whose column shape is a template argument (dense_loader.hpp dense_shape)

// The loads of one tile, issued D tiles ahead of their use.
template <int SRC, int RPT, typename SHAPE = void>
struct ring_tile;
template <int RPT>
struct ring_tile<RING_SRC_SIMPLE, RPT, void> {
  uint64_t k[RPT], v[RPT];
};
template <int RPT>
struct ring_tile<RING_SRC_COLS, RPT, void> {
  dense_raw_tile<RPT> t;
};
template <int RPT>
struct ring_tile<RING_SRC_REGIONS, RPT, void> {
  uint64_t v[RPT];
  uint32_t t[RPT];
};
template <int RPT, typename SHAPE>
struct ring_tile<RING_SRC_COLS_STATIC, RPT, SHAPE> {
  dense_static_tile<SHAPE, RPT> t;
};

// home slot of a key in the heavy-hitter table (every row of a call with heavy hitters pays it: two 32-bit multiplies instead
// of mix64's eight - the scatter with heavy hitters ran at half the plain one's rate)
static_assert((HOT_SLOTS & (HOT_SLOTS - 1)) == 0 && HOT_SLOTS >= 2, "heavy-hitter table: a power of two");
__device__ __forceinline__ uint32_t hot_slot_of(uint64_t key)
{
  uint32_t const h = static_cast<uint32_t>(key) * 0x9e3779b1u + static_cast<uint32_t>(key >> 32) * 0x85ebca77u;
  return h >> (32 - __builtin_ctz(HOT_SLOTS));
}

// TAG: uint16_t for the last level (a table has at most 2^15 slots: slot | validity << 15), uint32_t for the first of two levels
// (the bits of the scrambled index below its digit | validity << 31).
// HOT: heavy-hitter keys are aggregated in a small LDS table behind the rings and leave the scatter (a key with percents of the
// rows would overflow its partition's regions and leave one aggregate workgroup with its rows alone).
template <int SRC, int RPT, int D, typename TAG, bool HOT = false, typename SHAPE = void>
__global__ void __launch_bounds__(1024) k_dense_ring_scatter(dense_ring_args const* __restrict__ ap)
{
  constexpr bool COLS = SRC == RING_SRC_COLS || SRC == RING_SRC_COLS_STATIC;
  static_assert(!HOT || SRC == RING_SRC_SIMPLE, "heavy hitters: one plain key column and one plain value column");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int s_pending, s_abort;
  __shared__ uint32_t s_hot_dumped;
  __shared__ int32_t s_pre[MAX_REGION_LIST + 1];
  dense_ring_args const& a = *ap;
  plan_dev const& p        = a.plan;
  constexpr int B          = 1024;
  constexpr uint32_t G = 16, GT = 128 / sizeof(TAG);  // records per value granule / per tag granule (both 128 bytes)
  constexpr int TL      = sizeof(TAG) == 2 ? 2 : 1;   // log2(tag ring length / value ring length): 8 bytes of tags per value slot
  constexpr uint32_t TPL = 16 / sizeof(TAG);          // tags per lane of a flush (16 bytes)
  constexpr int VBIT    = 8 * sizeof(TAG) - 1;        // validity bit of an output tag
  int const P = a.P, capl = a.capl;
  uint32_t const CAP = 1u << capl, cmask = CAP - 1u, tcmask = (CAP << TL) - 1u;
  uint64_t* rval  = reinterpret_cast<uint64_t*>(lds_raw);                      // [P << capl] = DENSE_RING_SLOTS values
  TAG* rtag       = reinterpret_cast<TAG*>(rval + DENSE_RING_SLOTS);          // [P << (capl + TL)] tags: 8 * DENSE_RING_SLOTS bytes
  uint32_t* tail  = reinterpret_cast<uint32_t*>(rval + 2 * DENSE_RING_SLOTS); // [P] next virtual position
  uint32_t* limit = tail + P;                                                  // [P] head + CAP as of the last flush
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = B >> 6;
  int const PW = P / nwaves;  // partitions owned by a wave (1 ... 16): owner lane l < PW holds partition wave * PW + l
  // owner lanes: values / tags of the partition flushed so far (multiples of G / GT until the final flush; headt <= head < headt + GT)
  uint32_t head = 0, headt = 0;
  for (int d = threadIdx.x; d < P; d += B) {
    tail[d]  = 0;
    limit[d] = CAP;
  }
  if (threadIdx.x == 0) {
    s_pending    = 0;
    s_abort      = 0;
    s_hot_dumped = 0;
  }
  // ---- heavy hitters: open-addressing table [HOT_SLOTS] of key / sum / count behind the rings and their counters
  constexpr uint64_t HOT_EMPTY = ~uint64_t{0};
  [[maybe_unused]] uint64_t* hkeys = reinterpret_cast<uint64_t*>(lds_raw + static_cast<std::size_t>(DENSE_RING_SLOTS) * 16 + 2048);
  [[maybe_unused]] uint64_t* hsum  = hkeys + HOT_SLOTS;
  [[maybe_unused]] uint32_t* hcnt  = reinterpret_cast<uint32_t*>(hsum + HOT_SLOTS);
  [[maybe_unused]] bool hot_float  = false;
  if constexpr (HOT) {
    for (int q = 0; q < p.NACC; ++q) hot_float = hot_float || p.acc[q].op == ADD_F64;
    for (int t = threadIdx.x; t < HOT_SLOTS; t += B) {
      hkeys[t] = HOT_EMPTY;
      hsum[t]  = 0;
      hcnt[t]  = 0;
    }
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < a.hot_n) {
      uint64_t const key = a.hot_keys[threadIdx.x];
      uint32_t slot      = hot_slot_of(key);
      while (atomicCAS(reinterpret_cast<unsigned long long*>(hkeys + slot), HOT_EMPTY, key) != HOT_EMPTY) slot = (slot + 1) & (HOT_SLOTS - 1);
    }
  }
  // ---- this work item's rows and output regions
  constexpr int64_t T = static_cast<int64_t>(B) * RPT;
  int item = blockIdx.x, seg = 0;
  int64_t begin, end, step;
  int nreg = 0;
  int64_t rfirst = 0, rstride = 0;
  if constexpr (SRC == RING_SRC_REGIONS) {
    // the level-1 regions (seg, w), w = item, item + slices, ... as one virtual row range; s_pre[j] = rows before listed region j
    seg     = blockIdx.x / a.slices;
    item    = blockIdx.x % a.slices;
    nreg    = (a.in_slices - item + a.slices - 1) / a.slices;
    rfirst  = (static_cast<int64_t>(seg) * a.in_slices + item) * a.in_region_cap;
    rstride = static_cast<int64_t>(a.slices) * a.in_region_cap;
    if (threadIdx.x == 0) {
      int32_t run = 0;
      for (int j = 0; j < nreg; ++j) {
        s_pre[j] = run;
        // (clamped: a level-1 workgroup that gave up left its counts unwritten)
        int64_t const c = a.in_region_count[static_cast<int64_t>(seg) * a.in_slices + item + static_cast<int64_t>(j) * a.slices];
        run += static_cast<int32_t>(min(max(c, int64_t{0}), a.in_region_cap));
      }
      s_pre[nreg] = run;
    }
    __syncthreads();
    begin = 0;
    end   = s_pre[nreg];
    step  = T;
  } else {
    // workgroup w takes the row tiles w, w + slices, ...: sorted or clustered keys spread over every workgroup's regions
    begin = static_cast<int64_t>(item) * T;
    end   = a.nrows;
    step  = static_cast<int64_t>(a.slices) * T;
    __syncthreads();
  }
  int64_t const region0     = static_cast<int64_t>(seg) * P * a.slices;
  uint32_t const region_cap = static_cast<uint32_t>(a.region_cap);
  uint64_t* const out_val   = a.out_val;
  TAG* const out_tag        = static_cast<TAG*>(a.out_tag);
  [[maybe_unused]] uint64_t const dense_lo = a.map.lo, dense_range = a.map.range;
  [[maybe_unused]] uint32_t const mult = a.map.mult, bmask = (1u << a.map.bits) - 1u;
  int const shift = a.shift;
  uint32_t const pmask = static_cast<uint32_t>(P - 1), tmask = (1u << shift) - 1u;
  [[maybe_unused]] uint64_t const* kbase = p.simple_base[0];
  [[maybe_unused]] uint64_t const* vbase = p.simple_base[1];

  // virtual row of the region list -> record index. A thread's rows ascend from tile to tile, so the listed region of its k-th
  // row only moves forward: a hint per k and a walk of usually zero steps (a binary search per row: five dependent LDS reads).
  [[maybe_unused]] int reg_hint[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) reg_hint[k] = 0;
  auto record_of = [&](int64_t v, int& reg) -> int64_t {
    while (reg + 1 < nreg && s_pre[reg + 1] <= v) ++reg;
    return rfirst + static_cast<int64_t>(reg) * rstride + (v - s_pre[reg]);
  };
  [[maybe_unused]] dense_local const L = COLS ? make_dense_local(p, a.map, a.ones) : dense_local{};
  auto issue = [&](int64_t tile, ring_tile<SRC, RPT, SHAPE>& r) {
    if constexpr (SRC == RING_SRC_COLS) {
      issue_dense_local<RPT>(L, tile, B, end, r.t);
    } else if constexpr (SRC == RING_SRC_COLS_STATIC) {
      issue_dense_static<SHAPE, RPT>(L, tile, B, end, r.t);
    } else {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        int64_t const row = tile + static_cast<int64_t>(k) * B + threadIdx.x;
        if (row < end) {
          if constexpr (SRC == RING_SRC_SIMPLE) {
            r.k[k] = gload(kbase + row);
            r.v[k] = gload(vbase + row);
          } else {
            int64_t const ri = record_of(row, reg_hint[k]);
            r.v[k]           = gload(a.in_val + ri);
            r.t[k]           = gload(a.in_tag + ri);
          }
        }
      }
    }
  };
  // flush every complete granule of this wave's partitions (common/ring_scatter.hpp); final: also the partial last granule
  auto flush = [&](bool final) {
    int const dmine = wave * PW + lane;
    auto const f    = cudf::detail::ring::plan_flush<G, GT>(tail, lane < PW, dmine, head, headt, CAP, region_cap, final, s_abort);
    auto const rbase_of = [&](int d) { return (region0 + static_cast<int64_t>(d) * a.slices + item) * a.region_cap; };
    cudf::detail::ring::flush_stream(rval, out_val, wave, PW, lane, f.nrec, head, f.ab,
                                     [&](uint32_t d, uint32_t pos) { return (d << capl) + (pos & cmask); }, rbase_of);
    cudf::detail::ring::flush_stream(rtag, out_tag, wave, PW, lane, f.nrect, headt, f.ab,
                                     [&](uint32_t d, uint32_t pos) { return (d << (capl + TL)) + (pos & tcmask); }, rbase_of);
    cudf::detail::ring::commit_flush(limit, lane < PW, dmine, head, headt, f, CAP);
  };

  ring_tile<SRC, RPT, SHAPE> pre[D];
#pragma unroll
  for (int j = 0; j < D; ++j) issue(begin + j * step, pre[j]);
  for (int64_t tile = begin; tile < end; tile += D * step) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      int64_t const t0 = tile + j * step;
      // ---- this tile's rows: partition digit, tag and value (waits for the tile's loads)
      bool keep[RPT];
      uint32_t d[RPT], tg[RPT];
      uint64_t val[RPT];
      if constexpr (COLS) {
        uint32_t idx32[RPT], valid[RPT];
        bool bad = false;
        if (t0 < end) {  // (uniform)
          // (tried: decode one half of the tile and at once issue the same half of the tile after next into the registers that
          // frees, so that loads stay in flight through the decode phase - C4's first level 11.9 ms against 9.4 ms)
          if constexpr (SRC == RING_SRC_COLS) {
            decode_dense_local<RPT>(p, a.map, L, t0, B, end, pre[j].t, keep, idx32, valid, val, bad);
            issue_dense_local<RPT>(L, t0 + D * step, B, end, pre[j].t);
          } else {
            decode_dense_static<SHAPE, RPT>(L, t0, B, end, pre[j].t, keep, idx32, valid, val, bad);
            issue_dense_static<SHAPE, RPT>(L, t0 + D * step, B, end, pre[j].t);
          }
          if (bad) atomicOr(a.overflow, 4);
#pragma unroll
          for (int k = 0; k < RPT; ++k) {
            uint32_t const x = (idx32[k] * mult) & bmask;
            d[k]  = (x >> shift) & pmask;
            tg[k] = (x & tmask) | (valid[k] << VBIT);
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          keep[k] = t0 + static_cast<int64_t>(k) * B + threadIdx.x < end;
          d[k]    = 0;
          tg[k]   = 0;
          val[k]  = pre[j].v[k];
          if constexpr (HOT) {  // a heavy hitter is accumulated here and leaves the scatter
            if (keep[k]) {
              uint64_t const key = pre[j].k[k];
              uint32_t slot      = hot_slot_of(key);
              for (;;) {
                uint64_t const kk = hkeys[slot];
                if (kk == key) {
                  if (hot_float) atomicAdd(reinterpret_cast<double*>(hsum + slot), __longlong_as_double(static_cast<long long>(val[k])));
                  else atomicAdd(reinterpret_cast<unsigned long long*>(hsum + slot), static_cast<unsigned long long>(val[k]));
                  atomicAdd(hcnt + slot, 1u);
                  keep[k] = false;
                  break;
                }
                if (kk == HOT_EMPTY) break;
                slot = (slot + 1) & (HOT_SLOTS - 1);
              }
            }
          }
          if (keep[k]) {
            if constexpr (SRC == RING_SRC_SIMPLE) {
              uint64_t idx = pre[j].k[k] - dense_lo;
              if (idx >= dense_range) {  // the sampled key range was wrong: this call is void (redone by hash)
                atomicOr(a.overflow, 4);
                idx = 0;
              }
              uint32_t const x = (static_cast<uint32_t>(idx) * mult) & bmask;
              d[k]  = (x >> shift) & pmask;
              tg[k] = (x & tmask) | (1u << VBIT);
            } else {
              uint32_t const x = pre[j].t[k];
              d[k]  = ((x & 0x7fffffffu) >> shift) & pmask;
              tg[k] = (x & tmask) | ((x >> 31) << VBIT);
            }
          }
        }
      }
      if constexpr (!COLS) issue(t0 + D * step, pre[j]);
      if (t0 >= end) break;  // (uniform)
      // ---- reserve ring positions, write the records, flush; rows whose position lies beyond the ring wait for the flush
      cudf::detail::ring::place_tile<RPT>(
        tail, limit, keep, d, s_pending, s_abort,
        [&](int k, uint32_t pos) {
          rval[(d[k] << capl) + (pos & cmask)]                        = val[k];
          rtag[(d[k] << (capl + TL)) + (pos & tcmask)] = static_cast<TAG>(tg[k]);
        },
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
  if (lane < PW) a.region_count[region0 + static_cast<int64_t>(wave * PW + lane) * a.slices + item] = static_cast<int32_t>(head);
  if constexpr (HOT) {  // this workgroup's heavy-hitter partials: [key | accumulators in plan order]
    int const PU  = 1 + p.NACC;
    uint64_t* out = a.hot_out + static_cast<int64_t>(blockIdx.x) * HOT_SLOTS * PU;
    for (int t = threadIdx.x; t < HOT_SLOTS; t += B) {
      if (hcnt[t] == 0) continue;
      uint32_t const at = atomicAdd(&s_hot_dumped, 1u);
      gstore(out + static_cast<int64_t>(at) * PU, hkeys[t]);
      for (int q = 0; q < p.NACC; ++q)
        gstore(out + static_cast<int64_t>(at) * PU + 1 + q, p.acc[q].src == SRC_VALUE ? hsum[t] : static_cast<uint64_t>(hcnt[t]));
    }
    __syncthreads();
    if (threadIdx.x == 0) a.hot_count[blockIdx.x] = static_cast<int32_t>(s_hot_dumped);
  }
}

template <int SRC, int RPT, int D, typename TAG, bool HOT = false, typename SHAPE = void>
void launch_ring_tag(dense_ring_args const& a, dense_ring_args const* d_args, hipStream_t stream)
{
  static std::once_flag attr_once;  // (the API is re-entrant across objects: two threads may launch this kernel first)
  std::call_once(attr_once, [] { allow_full_lds(reinterpret_cast<void const*>(&k_dense_ring_scatter<SRC, RPT, D, TAG, HOT, SHAPE>)); });
  // rings (values + tags), 2 KiB for the ring counters of up to 256 partitions, then the heavy-hitter table
  std::size_t const lds = static_cast<std::size_t>(DENSE_RING_SLOTS) * 16 + 2048 + (HOT ? static_cast<std::size_t>(HOT_SLOTS) * (8 + 8 + 4) : 0);
  int const items       = a.from_columns ? a.slices : a.nseg * a.slices;
  cudf::detail::prof::scope prof_{a.from_columns ? "partition_scatter" : "partition_scatter_level2", stream};
  hipLaunchKernelGGL((k_dense_ring_scatter<SRC, RPT, D, TAG, HOT, SHAPE>), dim3(items), dim3(1024), lds, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}
template <int SRC, int RPT, int D, typename SHAPE = void>
void launch_ring_t(dense_ring_args const& a, dense_ring_args const* d_args, hipStream_t stream)
{
  if (a.tag16) return launch_ring_tag<SRC, RPT, D, uint16_t, false, SHAPE>(a, d_args, stream);
  if constexpr (SRC != RING_SRC_REGIONS) return launch_ring_tag<SRC, RPT, D, uint32_t, false, SHAPE>(a, d_args, stream);
  CUDF_FAIL("ring scatter: the last level writes 16-bit tags");
}
// composite dense keys: the column shapes with a loader of their own (dense_loader.hpp dense_shape), else the run-time loader
using shape_c4          = dense_shape<8, false, 4, true, true>;   // BASELINE configs[3]: (int64, nullable int32) keys, nullable 8-byte value
using shape_k32         = dense_shape<4, false, 0, false, false>; // one 4-byte key, plain 8-byte value
using shape_k32_vnull   = dense_shape<4, false, 0, false, true>;
using shape_k64_vnull   = dense_shape<8, false, 0, false, true>;  // one 8-byte key, nullable 8-byte value
using shape_k64_k32     = dense_shape<8, false, 4, false, false>; // (int64, int32) keys and an 8-byte value, no nulls
template <typename SHAPE>
bool try_static_shape(dense_ring_args const& a, dense_ring_args const* d_args, hipStream_t stream)
{
  if (!dense_shape_matches<SHAPE>(a.plan, a.map)) return false;
  // Two tiles of loads in flight where the shape's registers allow it (one key column: 106-123 VGPRs, no scratch; 1B rows on 1M
  // groups, int32 key 7.9 -> 6.3 ms, int64 key with a nullable value 8.1 -> 7.2 ms). Two key columns with masks (C4's shape) spill
  // 32-48 bytes with two tiles and run no faster than the run-time loader (13.6 ms); with one tile C4's first level takes 6.06
  // instead of 7.2 ms (profiles/r3_c4_static_ab.txt). CUDF_AMD_GB_STATIC_SHAPES=1 forces one tile, 0 the run-time loader.
  constexpr bool WIDE = SHAPE::W1 != 0 && (SHAPE::M0 || SHAPE::M1);
  if (WIDE || a.static_shapes == 1) launch_ring_t<RING_SRC_COLS_STATIC, 4, 1, SHAPE>(a, d_args, stream);
  else launch_ring_t<RING_SRC_COLS_STATIC, 4, STATIC_TILES_AHEAD, SHAPE>(a, d_args, stream);
  return true;
}

}  // namespace

void store_args(dense_ring_args const& a, dense_ring_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<dense_ring_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_dense_ring_scatter(dense_ring_args const& a, dense_ring_args const* d_args, hipStream_t stream)
{
  CUDF_EXPECTS(a.hot_n == 0 || a.from_columns, "ring scatter: heavy hitters are taken out on the first level");
  CUDF_EXPECTS(a.P >= 16 && a.P <= 256 && (a.P & (a.P - 1)) == 0 && (a.P << a.capl) == DENSE_RING_SLOTS && a.capl >= 5,
               "ring scatter: fan-out 16 ... 256, rings of at least two granules");
  CUDF_EXPECTS(a.region_cap % 64 == 0 && a.shift >= 0 && a.shift < 31 && a.slices >= 1 && (!a.tag16 || a.shift <= 15), "ring scatter: region geometry");
  if (a.from_columns) {
    // (composite loader: 4 rows per thread, one tile ahead. Measured on C4's first level: two tiles ahead spill - 14.3 ms against
    // 9.4 - and 2 or 3 rows per thread with two or three tiles ahead take 11.3-13.1 ms: the per-tile barriers and flush do not amortise)
    if (a.map.nkeys > 0) {
      if (a.static_shapes &&
          (try_static_shape<shape_c4>(a, d_args, stream) || try_static_shape<shape_k32>(a, d_args, stream) ||
           try_static_shape<shape_k32_vnull>(a, d_args, stream) || try_static_shape<shape_k64_vnull>(a, d_args, stream) ||
           try_static_shape<shape_k64_k32>(a, d_args, stream)))
        return;
      return launch_ring_t<RING_SRC_COLS, 4, 1>(a, d_args, stream);
    }
    CUDF_EXPECTS(a.plan.simple && a.plan.KU == 1 && a.plan.NPAY == 1, "ring scatter: one plain key column and one plain value column");
    if (a.hot_n > 0) {  // (the first level of one or of two levels)
      CUDF_EXPECTS(a.hot_n <= HOT_MAX_KEYS, "ring scatter: heavy hitters");
      if (a.tag16) return launch_ring_tag<RING_SRC_SIMPLE, 4, 2, uint16_t, true>(a, d_args, stream);
      return launch_ring_tag<RING_SRC_SIMPLE, 4, 2, uint32_t, true>(a, d_args, stream);
    }
    return launch_ring_t<RING_SRC_SIMPLE, 4, 2>(a, d_args, stream);
  }
  CUDF_EXPECTS((a.in_slices + a.slices - 1) / a.slices <= MAX_REGION_LIST, "ring scatter: region list too long");
  return launch_ring_t<RING_SRC_REGIONS, 4, 2>(a, d_args, stream);
}

}  // namespace cudf::groupby::detail
