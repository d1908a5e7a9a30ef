// SPDX-License-Identifier: Apache-2.0
// gfx950 kernels of the DENSE-KEY hash-groupby path (engine.hpp dense_map / dense_agg_args): one plain 8-byte integer
// key column whose values span a small range is aggregated by direct addressing - the perfect-hash special case of the
// open-addressing tables of aggregate_kernels.hip: no hash, no probe, no key words, no state words; a row costs one
// 16-byte load, one 32-bit multiply and its ds_add / ds_min / ds_max atomics.
// Replaces, for such keys, the reference's cuco::static_set insert + global atomics
// (cpp/src/groupby/hash/compute_global_memory_aggs.cuh:74-187, single_pass_functors.cuh:86-157).
#include "device_common.hpp"

namespace cudf::groupby::detail {
namespace {

// ------------------------------------------------------------------ K_key_range (sample minimum / maximum)
__device__ __forceinline__ int64_t dense_sample_row(int64_t i, int64_t nrows, int64_t sample)
{
  // the strided sample of k_estimate (aggregate_kernels.hip sample_row): one row per stratum at a pseudo-random offset
  if (sample >= nrows) return i;
  int64_t const lo    = static_cast<int64_t>((static_cast<__int128>(i) * nrows) / sample);
  int64_t const hi    = static_cast<int64_t>((static_cast<__int128>(i + 1) * nrows) / sample);
  uint64_t const span = static_cast<uint64_t>(hi - lo);
  return span <= 1 ? lo : lo + static_cast<int64_t>(mix64(static_cast<uint64_t>(i) + 0x51ed270b35a3c1ull) % span);
}
template <bool SIGNED>
__global__ void __launch_bounds__(256) k_key_range(plan_dev const* __restrict__ pp, int64_t nrows, int64_t sample, uint64_t* out)
{
  plan_dev const& p = *pp;
  using T           = std::conditional_t<SIGNED, long long, unsigned long long>;
  __shared__ T s_lo[4], s_hi[4];
  T lo = SIGNED ? static_cast<T>(INT64_MAX) : static_cast<T>(UINT64_MAX), hi = SIGNED ? static_cast<T>(INT64_MIN) : T{0};
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < sample; i += stride) {
    T const k = static_cast<T>(gload(p.simple_base[0] + dense_sample_row(i, nrows, sample)));
    lo = k < lo ? k : lo;
    hi = k > hi ? k : hi;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    T const l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  // (same-address global atomics serialise at ~12 ns each: one pair per workgroup, not per wave)
  if ((threadIdx.x & 63) == 0) {
    s_lo[threadIdx.x >> 6] = lo;
    s_hi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      lo = s_lo[w] < lo ? s_lo[w] : lo;
      hi = s_hi[w] > hi ? s_hi[w] : hi;
    }
    atomicMin(reinterpret_cast<T*>(out), lo);
    atomicMax(reinterpret_cast<T*>(out) + 1, hi);
  }
}

// sample minimum / maximum of EVERY key column of a plan (any integer width, nullable): valid values only, as int64
__global__ void __launch_bounds__(256) k_key_ranges(plan_dev const* __restrict__ pp, int nkeycols, int64_t nrows, int64_t sample, long long* out)
{
  plan_dev const& p = *pp;
  __shared__ long long s_lo[4][MAX_KU], s_hi[4][MAX_KU];
  long long lo[MAX_KU], hi[MAX_KU];
#pragma unroll
  for (int c = 0; c < MAX_KU; ++c) {
    lo[c] = INT64_MAX;
    hi[c] = INT64_MIN;
  }
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < sample; i += stride) {
    int64_t const row = dense_sample_row(i, nrows, sample);
#pragma unroll
    for (int c = 0; c < MAX_KU; ++c) {
      if (c >= nkeycols) break;
      device_column const col = p.cols[c];
      if (!col_is_valid(col, row)) continue;
      uint64_t const raw = col_load_bits(col, row);
      int const sh       = 64 - 8 * col.width;
      long long const v  = col.cls == cudf::detail::CLS_SINT ? (static_cast<long long>(raw << sh) >> sh) : static_cast<long long>(raw);
      lo[c] = v < lo[c] ? v : lo[c];
      hi[c] = v > hi[c] ? v : hi[c];
    }
  }
#pragma unroll
  for (int c = 0; c < MAX_KU; ++c) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      long long const l2 = __shfl_xor(lo[c], o), h2 = __shfl_xor(hi[c], o);
      lo[c] = l2 < lo[c] ? l2 : lo[c];
      hi[c] = h2 > hi[c] ? h2 : hi[c];
    }
    if ((threadIdx.x & 63) == 0) {
      s_lo[threadIdx.x >> 6][c] = lo[c];
      s_hi[threadIdx.x >> 6][c] = hi[c];
    }
  }
  __syncthreads();
  if (static_cast<int>(threadIdx.x) < nkeycols) {
    int const c = threadIdx.x;
    long long l = s_lo[0][c], h = s_hi[0][c];
    for (int w = 1; w < 4; ++w) {
      l = s_lo[w][c] < l ? s_lo[w][c] : l;
      h = s_hi[w][c] > h ? s_hi[w][c] : h;
    }
    atomicMin(out + 2 * c, l);
    atomicMax(out + 2 * c + 1, h);
  }
}

// ------------------------------------------------------------------ K_aggregate_dense
// LDS image of a table: accumulator q as [slots] u64 (u32 for a COUNT), 8-byte arrays first; then the occupancy bitmap
// [slots / 32] u32 if no accumulator counts every row. The image is copied verbatim from / to HBM between chunks.
struct dense_layout {
  uint32_t off[MAX_ACC];
  uint32_t occ_off;
  uint32_t bytes;
};
__host__ __device__ inline dense_layout make_dense_layout(plan_dev const& p, int slots, int occ_acc)
{
  dense_layout L{};
  uint32_t o = 0;
  for (int q = 0; q < p.NACC; ++q)
    if (!acc_is_narrow(p.acc[q].op, p.acc[q].src)) {
      L.off[q] = o;
      o += static_cast<uint32_t>(slots) * 8u;
    }
  for (int q = 0; q < p.NACC; ++q)
    if (acc_is_narrow(p.acc[q].op, p.acc[q].src)) {
      L.off[q] = o;
      o += static_cast<uint32_t>(slots) * 4u;
    }
  L.occ_off = o;
  if (occ_acc < 0) o += static_cast<uint32_t>((slots + 31) / 32) * 4u;
  L.bytes = (o + 15u) & ~15u;
  return L;
}

// SIG: compile-time accumulator signature (device_common.hpp), 0 = descriptors read at run time.
// One workgroup per partition; wave w walks the regions [w * g, (w + 1) * g) of its partition as ONE virtual record range
// (a chunk's regions hold 50-400 records each: walked one by one they would be all tail).
// The key units of the partial record of dense index `idx` (what k_finalize reads): a single plain key is lo + idx; composite
// keys are rebuilt from the mixed-radix digits of the index and placed where the plan's key units hold them.
__device__ __forceinline__ void dense_store_key_units(dense_map const& m, int KU, uint32_t idx, uint64_t* o)
{
  if (m.nkeys == 0) {
    gstore(o, m.lo + idx);
    return;
  }
  uint64_t unit[MAX_KU] = {0, 0, 0, 0};
  for (int c = 0; c < m.nkeys; ++c) {
    dense_key const dk   = m.key[c];
    uint64_t const digit = (idx / dk.stride) % dk.range;
    uint64_t v           = dk.lo + digit;
    if (dk.width < 8) v &= (uint64_t{1} << (8 * dk.width)) - 1;  // key units hold the zero-extended raw bits
#pragma unroll
    for (int u = 0; u < MAX_KU; ++u)
      if (u == dk.unit) unit[u] |= dk.half == 1 ? (v << 32) : v;
  }
#pragma unroll
  for (int u = 0; u < MAX_KU; ++u)
    if (u < KU) gstore(o + u, unit[u]);
}

// SOA: 10-byte records of the ring scatter (value stream + 16-bit tag stream) instead of 16-byte {key | value} records.
template <uint64_t SIG, int NACCT, bool SOA>
__global__ void __launch_bounds__(1024, 8) k_aggregate_dense(dense_agg_args const* __restrict__ ap, int first_chunk, int last_chunk)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ uint32_t s_dump;
  dense_agg_args const& a = *ap;
  plan_dev const& p       = a.plan;
  constexpr bool STATIC_SIG = SIG != 0;
  int const NACC  = STATIC_SIG ? sig_n(SIG) : p.NACC;
  // work item = (partition, share h of its regions)
  int const nsplit = a.nsplit > 1 ? a.nsplit : 1;
  int const slots = a.slots, item = blockIdx.x / nsplit, B = blockDim.x;
  int const nsl = a.slices / nsplit, sl0 = (static_cast<int>(blockIdx.x) % nsplit) * nsl;
  int const occ_acc = a.occ_acc;
  int acc_op[NACCT], acc_src[NACCT], acc_vbit[NACCT];
  uint32_t acc_off[NACCT];
  bool acc_narrow[NACCT];
  uint32_t off8 = 0, off4 = 0;
#pragma unroll
  for (int j = 0; j < NACCT; ++j) {
    uint32_t const w = j < NACC ? reinterpret_cast<uint32_t const*>(p.acc)[j] : 0u;
    acc_op[j]        = STATIC_SIG ? sig_op(SIG, j) : static_cast<int8_t>(w);
    acc_src[j]       = STATIC_SIG ? sig_src(SIG, j) : static_cast<int8_t>(w >> 8);
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

  // ---- table: identities on the first chunk, the carried image afterwards
  uint64_t* image = a.tables + static_cast<int64_t>(blockIdx.x) * (a.image_bytes / 8);
  if (first_chunk) {
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
  } else {
    for (int i = threadIdx.x; i < a.image_bytes / 16; i += B)
      reinterpret_cast<u64x2*>(lds_raw)[i] = gload(reinterpret_cast<u64x2 const*>(image) + i);
  }
  if (threadIdx.x == 0) s_dump = 0;
  __syncthreads();

  uint64_t const lo = a.map.lo;
  uint32_t const mult = a.map.mult, smask = static_cast<uint32_t>(slots - 1);
  bool const composite = a.map.nkeys > 0;  // records hold {index | validity of the value, value}
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = B >> 6;
  // an upstream scatter gave up (region overflow / key outside the dense range): its counts are not valid
  bool const upstream_ok = *a.overflow == 0;
  // ---- the partition's regions as virtual record ranges. Many short regions (one per scatter workgroup): wave w walks the
  // regions [w * g, (w + 1) * g) as ONE range. A few long ones (the output of a second partition level): every wave sees all
  // of them as one range and the waves take its batches in turn.
  bool const shared = nsl <= 64;
  int const g       = shared ? nsl : (nsl + nwaves - 1) / nwaves;  // regions in this wave's range, <= 64
  int const r_base  = shared ? 0 : wave * g;
  int32_t cnt       = 0;
  if (upstream_ok && lane < g && r_base + lane < nsl)
    cnt = min(max(a.region_count[static_cast<int64_t>(item) * a.slices + sl0 + r_base + lane], 0), static_cast<int32_t>(a.region_cap));
  int32_t pend = cnt;  // inclusive prefix: end of region `lane` in the virtual range
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int32_t const t = __shfl_up(pend, o);
    if (lane >= o) pend += t;
  }
  int32_t const total = __builtin_amdgcn_readlane(pend, 63);
  int64_t const cap   = a.region_cap;
  int64_t const rec0  = (static_cast<int64_t>(item) * a.slices + sl0 + r_base) * cap;
  [[maybe_unused]] u64x2 const* recs       = reinterpret_cast<u64x2 const*>(a.records) + rec0;
  [[maybe_unused]] uint64_t const* rec_val = a.rec_val + rec0;
  [[maybe_unused]] uint16_t const* rec_tag = a.rec_tag + rec0;
  int rcur            = 0;  // wave-uniform: first region that may hold the next virtual record
  // record index (relative to `recs`) of virtual record v; v ascends from call to call
  auto locate = [&](int32_t v, bool active) -> int64_t {
    int32_t const v0 = __builtin_amdgcn_readfirstlane(v);
    while (rcur < g - 1 && __builtin_amdgcn_readlane(pend, rcur) <= v0) ++rcur;
    int reg       = rcur;
    int32_t start = rcur == 0 ? 0 : __builtin_amdgcn_readlane(pend, rcur - 1);
    for (int i = rcur; i < g - 1; ++i) {
      int32_t const e = __builtin_amdgcn_readlane(pend, i);
      bool const ge   = active && v >= e;
      if (__ballot(ge) == 0) break;
      reg += ge ? 1 : 0;
      start = ge ? e : start;
    }
    return static_cast<int64_t>(reg) * cap + (v - start);
  };
  auto accumulate = [&](uint64_t key, uint64_t value) {
    uint32_t s;
    bool val_valid;
    if constexpr (SOA) {  // key = the record's tag: slot | validity of the value << 15
      s         = static_cast<uint32_t>(key) & smask;
      val_valid = (static_cast<uint32_t>(key) >> 15) != 0;
    } else {
      uint32_t const idx = composite ? static_cast<uint32_t>(key) : static_cast<uint32_t>(key - lo);
      val_valid          = !composite || ((key >> 32) & 1u);  // (a single plain key column comes with a plain value column)
      s                  = (idx * mult) & smask;
    }
#pragma unroll
    for (int q = 0; q < NACCT; ++q) {
      if (q >= NACC) break;
      bool const counts_all = acc_src[q] == SRC_ONE;
      if (!counts_all && acc_vbit[q] >= 0 && !val_valid) continue;  // a NULL value reaches nothing but COUNT_ALL
      if (acc_narrow[q]) {
        atomicAdd(acc32(q) + s, 1u);
        continue;
      }
      uint64_t const v = acc_contribution(acc_src[q], acc_op[q], value);
      lds_merge(acc64(q) + s, acc_op[q], v);
    }
    if (occ_acc < 0) {
      uint32_t const bit = 1u << (s & 31);
      if (!(occ[s >> 5] & bit)) atomicOr(&occ[s >> 5], bit);
    }
  };
  // Batches of R x 64 records; the loads of the next batch are issued before the current one is accumulated (one 1024-thread
  // workgroup per CU is all a 96 KiB table allows: 16 waves x one batch in flight did not cover the memory latency - C2's
  // aggregate 2.8 ms against 2.0 ms with two workgroups per CU).
  constexpr int R = 4;
  int32_t const vstep = shared ? R * 64 * nwaves : R * 64;
  // (the record indices of a batch are resolved first - locate() is loops and cross-lane reads - and its loads then go out
  // back to back, unconditionally: a lane past the end reads record 0 of the range. With a load inside each row's branch the
  // compiler waited for the previous load before every new one.)
  auto load_batch = [&](int32_t v0, u64x2 (&rec)[R], bool (&act)[R]) {
    int64_t ri[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      int32_t const v = v0 + k * 64 + lane;
      act[k]          = v < total;
      ri[k]           = 0;
      if (v0 + k * 64 < total) {  // (wave-uniform)
        int64_t const r = locate(v, act[k]);
        ri[k]           = act[k] ? r : 0;
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if constexpr (SOA) {
        rec[k].y = gload(rec_val + ri[k]);
        rec[k].x = gload(rec_tag + ri[k]);
      } else {
        rec[k] = gload(recs + ri[k]);
      }
    }
  };
  auto accumulate_batch = [&](u64x2 const (&rec)[R], bool const (&act)[R]) {
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (act[k]) accumulate(rec[k].x, rec[k].y);
  };
  {
    u64x2 recA[R], recB[R];
    bool actA[R], actB[R];
    int32_t v0 = shared ? wave * R * 64 : 0;
    if (v0 < total) {
      load_batch(v0, recA, actA);
      for (;;) {
        load_batch(v0 + vstep, recB, actB);  // (nothing is loaded past `total`)
        accumulate_batch(recA, actA);
        v0 += vstep;
        if (v0 >= total) break;
        load_batch(v0 + vstep, recA, actA);
        accumulate_batch(recB, actB);
        v0 += vstep;
        if (v0 >= total) break;
      }
    }
  }
  __syncthreads();
  if (!last_chunk || nsplit > 1 || a.keep_images) {  // the image is carried to the next chunk, or merged with the partition's other shares / columns
    for (int i = threadIdx.x; i < a.image_bytes / 16; i += B)
      gstore(reinterpret_cast<u64x2*>(image) + i, reinterpret_cast<u64x2 const*>(lds_raw)[i]);
    return;
  }
  // ---- last chunk: occupied slots -> partial records [key units | accumulators] (k_finalize reads them)
  int const KU       = a.KU;
  int const PU       = KU + NACC;
  uint64_t* out      = a.out_records + static_cast<int64_t>(item) * slots * PU;
  uint32_t const hi  = static_cast<uint32_t>(item) << (a.map.bits - a.map.log2P);
  uint32_t const bmask = (1u << a.map.bits) - 1u;
  for (int s = threadIdx.x; s < slots; s += B) {
    bool occupied;
    if (occ_acc >= 0) {
      occupied = false;
#pragma unroll
      for (int q = 0; q < NACCT; ++q)
        if (q == occ_acc) occupied = acc32(q)[s] != 0;
    } else {
      occupied = (occ[s >> 5] >> (s & 31)) & 1u;
    }
    if (!occupied) continue;
    uint32_t const pos = atomicAdd(&s_dump, 1u);
    uint64_t* o        = out + static_cast<int64_t>(pos) * PU;
    uint32_t const idx = ((hi | static_cast<uint32_t>(s)) * a.map.mult_inv) & bmask;
    dense_store_key_units(a.map, KU, idx, o);
#pragma unroll
    for (int q = 0; q < NACCT; ++q) {
      if (q >= NACC) break;
      gstore(o + KU + q, acc_narrow[q] ? static_cast<uint64_t>(acc32(q)[s]) : acc64(q)[s]);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) a.out_count[item] = static_cast<int32_t>(s_dump);
}

// ------------------------------------------------------------------ K_dense_merge_dump
// nsplit table images per partition (left by k_aggregate_dense) -> partial records. Work item (d, j): slots [j, j + 1) * slots /
// dsplit of partition d; one thread per slot folds the nsplit images (coalesced reads) and appends the slot's record if occupied.
__global__ void __launch_bounds__(1024) k_dense_merge_dump(dense_agg_args const* __restrict__ ap, int dsplit)
{
  __shared__ uint32_t s_dump;
  dense_agg_args const& a = *ap;
  plan_dev const& p       = a.plan;
  int const NACC = p.NACC, slots = a.slots, nsplit = a.nsplit, B = blockDim.x;
  int const d = blockIdx.x / dsplit, per = slots / dsplit, s0 = (static_cast<int>(blockIdx.x) % dsplit) * per;
  dense_layout const L = make_dense_layout(p, slots, a.occ_acc);
  unsigned char const* images = reinterpret_cast<unsigned char const*>(a.tables) + static_cast<int64_t>(d) * nsplit * a.image_bytes;
  if (threadIdx.x == 0) s_dump = 0;
  __syncthreads();
  int const KU = a.KU, PU = KU + NACC;
  uint64_t* out       = a.out_records + static_cast<int64_t>(blockIdx.x) * per * PU;
  uint32_t const hi   = static_cast<uint32_t>(d) << (a.map.bits - a.map.log2P);
  uint32_t const bmask = (1u << a.map.bits) - 1u;
  for (int s = s0 + threadIdx.x; s < s0 + per; s += B) {
    uint64_t acc[MAX_ACC];
    bool occupied = false;
    for (int q = 0; q < NACC; ++q) {
      int const op = p.acc[q].op;
      bool const narrow = acc_is_narrow(op, p.acc[q].src);
      uint64_t v = narrow ? 0 : acc_identity(op);
      for (int h = 0; h < nsplit; ++h) {
        unsigned char const* img = images + static_cast<int64_t>(h) * a.image_bytes + L.off[q];
        if (narrow) v += gload(reinterpret_cast<uint32_t const*>(img) + s);
        else v = combine_values(op, v, gload(reinterpret_cast<uint64_t const*>(img) + s));
      }
      acc[q] = v;
      if (q == a.occ_acc) occupied = v != 0;
    }
    if (a.occ_acc < 0)
      for (int h = 0; h < nsplit; ++h)
        occupied = occupied || ((gload(reinterpret_cast<uint32_t const*>(images + static_cast<int64_t>(h) * a.image_bytes + L.occ_off) + (s >> 5)) >> (s & 31)) & 1u);
    if (!occupied) continue;
    uint32_t const pos = atomicAdd(&s_dump, 1u);
    uint64_t* o        = out + static_cast<int64_t>(pos) * PU;
    dense_store_key_units(a.map, KU, ((hi | static_cast<uint32_t>(s)) * a.map.mult_inv) & bmask, o);
    for (int q = 0; q < NACC; ++q) gstore(o + KU + q, acc[q]);
  }
  __syncthreads();
  if (threadIdx.x == 0) a.out_count[blockIdx.x] = static_cast<int32_t>(s_dump);
}

// The same for MANY images of one table (the one-table path: a.nsplit = hundreds of workgroups): work item b = slots [64 b, 64 b + 64);
// wave w of the workgroup folds the images w, w + 16, ... for its 64 slots (coalesced 512-byte reads), the sixteen partial
// results meet in LDS one accumulator at a time. (One thread per slot walking all images: 0.4 ms for 512 images.)
__global__ void __launch_bounds__(1024) k_dense_merge_dump_wide(dense_agg_args const* __restrict__ ap)
{
  __shared__ uint64_t s_part[16][64];
  __shared__ uint32_t s_dump;
  dense_agg_args const& a = *ap;
  plan_dev const& p       = a.plan;
  int const NACC = p.NACC, slots = a.slots, nimg = a.nsplit;
  int const ls = threadIdx.x & 63, hy = threadIdx.x >> 6;
  int const s  = static_cast<int>(blockIdx.x) * 64 + ls;
  bool const in = s < slots;
  dense_layout const L = make_dense_layout(p, slots, a.occ_acc);
  unsigned char const* images = reinterpret_cast<unsigned char const*>(a.tables);
  if (threadIdx.x == 0) s_dump = 0;
  uint64_t acc[MAX_ACC];
  bool occupied = false;
  for (int q = 0; q < NACC; ++q) {
    int const op      = p.acc[q].op;
    bool const narrow = acc_is_narrow(op, p.acc[q].src);
    uint64_t v        = narrow ? 0 : acc_identity(op);
    if (in)
      for (int h = hy; h < nimg; h += 16) {
        unsigned char const* img = images + static_cast<int64_t>(h) * a.image_bytes + L.off[q];
        if (narrow) v += gload(reinterpret_cast<uint32_t const*>(img) + s);
        else v = combine_values(op, v, gload(reinterpret_cast<uint64_t const*>(img) + s));
      }
    __syncthreads();  // (s_part of the previous accumulator has been read)
    s_part[hy][ls] = v;
    __syncthreads();
    if (hy == 0) {
      for (int w = 1; w < 16; ++w) v = narrow ? v + s_part[w][ls] : combine_values(op, v, s_part[w][ls]);
      acc[q] = v;
      if (q == a.occ_acc) occupied = v != 0;
    }
  }
  if (a.occ_acc < 0) {
    uint64_t bit = 0;
    if (in)
      for (int h = hy; h < nimg; h += 16)
        bit |= (gload(reinterpret_cast<uint32_t const*>(images + static_cast<int64_t>(h) * a.image_bytes + L.occ_off) + (s >> 5)) >> (s & 31)) & 1u;
    __syncthreads();
    s_part[hy][ls] = bit;
    __syncthreads();
    if (hy == 0)
      for (int w = 0; w < 16; ++w) occupied = occupied || s_part[w][ls] != 0;
  }
  __syncthreads();
  int const KU = a.KU, PU = KU + NACC;
  uint64_t* out        = a.out_records + static_cast<int64_t>(blockIdx.x) * 64 * PU;
  uint32_t const bmask = (1u << a.map.bits) - 1u;
  if (hy == 0 && in && occupied) {
    uint32_t const pos = atomicAdd(&s_dump, 1u);
    uint64_t* o        = out + static_cast<int64_t>(pos) * PU;
    dense_store_key_units(a.map, KU, (static_cast<uint32_t>(s) * a.map.mult_inv) & bmask, o);
    for (int q = 0; q < NACC; ++q) gstore(o + KU + q, acc[q]);
  }
  __syncthreads();
  if (threadIdx.x == 0) a.out_count[blockIdx.x] = static_cast<int32_t>(s_dump);
}

}  // namespace

int dense_occ_acc(plan_dev const& plan)
{
  for (int q = 0; q < plan.NACC; ++q)
    if (plan.acc[q].op == ADD_I64 && plan.acc[q].src == SRC_ONE) return q;
  return -1;
}
std::size_t dense_table_bytes(plan_dev const& plan, int slots) { return make_dense_layout(plan, slots, dense_occ_acc(plan)).bytes; }
uint32_t dense_acc_offset(plan_dev const& plan, int slots, int q) { return make_dense_layout(plan, slots, dense_occ_acc(plan)).off[q]; }
uint32_t dense_occ_offset(plan_dev const& plan, int slots) { return make_dense_layout(plan, slots, dense_occ_acc(plan)).occ_off; }

void store_args(dense_agg_args const& a, dense_agg_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<dense_agg_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

template <uint64_t SIG, int NACCT, bool SOA>
static void launch_dense_t(dense_agg_args const& a, dense_agg_args const* d_args, bool first_chunk, bool last_chunk, hipStream_t stream)
{
  static std::once_flag attr_once;  // (the API is re-entrant across objects: two threads may launch this kernel first)
  std::call_once(attr_once, [] { allow_full_lds(reinterpret_cast<void const*>(&k_aggregate_dense<SIG, NACCT, SOA>)); });
  cudf::detail::prof::scope prof_{"aggregate", stream};
  hipLaunchKernelGGL((k_aggregate_dense<SIG, NACCT, SOA>), dim3(a.nitems * std::max(a.nsplit, 1)), dim3(a.block), a.image_bytes, stream, d_args,
                     first_chunk ? 1 : 0, last_chunk ? 1 : 0);
  CUDF_HIP_TRY(hipGetLastError());
}
template <uint64_t SIG, int NACCT>
static void launch_dense_n(dense_agg_args const& a, dense_agg_args const* d_args, bool first_chunk, bool last_chunk, hipStream_t stream)
{
  if (a.rec_tag != nullptr) return launch_dense_t<SIG, NACCT, true>(a, d_args, first_chunk, last_chunk, stream);
  return launch_dense_t<SIG, NACCT, false>(a, d_args, first_chunk, last_chunk, stream);
}

void launch_dense_merge_dump_wide(dense_agg_args const& a, dense_agg_args const* d_args, hipStream_t stream)
{
  CUDF_EXPECTS(a.nitems == 1 && a.nsplit >= 1 && a.map.log2P == 0 && a.slots % 64 == 0, "dense keys: one table, its images");
  for (int q = 0; q < a.plan.NACC; ++q) CUDF_EXPECTS(a.plan.acc[q].op != ANY_U64, "dense keys: integer keys only");
  cudf::detail::prof::scope prof_{"aggregate_merge", stream};
  hipLaunchKernelGGL(k_dense_merge_dump_wide, dim3(a.slots / 64), dim3(1024), 0, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_dense_merge_dump(dense_agg_args const& a, dense_agg_args const* d_args, int dsplit, hipStream_t stream)
{
  CUDF_EXPECTS(a.nsplit >= 1 && dsplit >= 1 && a.slots % dsplit == 0, "dense keys: merge geometry");
  for (int q = 0; q < a.plan.NACC; ++q) CUDF_EXPECTS(a.plan.acc[q].op != ANY_U64, "dense keys: integer keys only");
  cudf::detail::prof::scope prof_{"aggregate_merge", stream};
  hipLaunchKernelGGL(k_dense_merge_dump, dim3(a.nitems * dsplit), dim3(1024), 0, stream, d_args, dsplit);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_aggregate_dense(dense_agg_args const& a, dense_agg_args const* d_args, bool first_chunk, bool last_chunk, hipStream_t stream)
{
  CUDF_EXPECTS(a.plan.narg == 0 && (a.map.nkeys > 0 || (a.plan.simple && a.plan.KU == 1 && a.plan.NPAY == 1)),
               "dense keys: one plain key column and one plain value column, or composite integer keys and one value column");
  CUDF_EXPECTS(a.nsplit <= 1 || a.slices % a.nsplit == 0, "dense keys: the regions of a partition divide evenly among its workgroups");
  CUDF_EXPECTS(a.block == 1024 && a.slices / std::max(a.nsplit, 1) <= 1024 && a.image_bytes <= 159 * 1024 && (a.slots & (a.slots - 1)) == 0 &&
                 a.occ_acc == dense_occ_acc(a.plan) && a.image_bytes == static_cast<int32_t>(dense_table_bytes(a.plan, a.slots)),
               "dense keys: table geometry");
  uint64_t const sig = plan_sig(a.plan);
  if (sig == SIG_SUMF_CNT) return launch_dense_n<SIG_SUMF_CNT, 2>(a, d_args, first_chunk, last_chunk, stream);
  if (sig == SIG_SUMI_CNT) return launch_dense_n<SIG_SUMI_CNT, 2>(a, d_args, first_chunk, last_chunk, stream);
  if (sig == SIG_SUMF) return launch_dense_n<SIG_SUMF, 2>(a, d_args, first_chunk, last_chunk, stream);
  if (sig == SIG_SUMI) return launch_dense_n<SIG_SUMI, 2>(a, d_args, first_chunk, last_chunk, stream);
  if (sig == SIG_CNT) return launch_dense_n<SIG_CNT, 2>(a, d_args, first_chunk, last_chunk, stream);
  if (sig == SIG_MEAN_MIN_MAX_F) return launch_dense_n<SIG_MEAN_MIN_MAX_F, 4>(a, d_args, first_chunk, last_chunk, stream);
  if (sig == SIG_MEAN_MIN_MAX_F_NULLS) return launch_dense_n<SIG_MEAN_MIN_MAX_F_NULLS, 4>(a, d_args, first_chunk, last_chunk, stream);
  if (sig == SIG_SUMF_CNT_NULLS) return launch_dense_n<SIG_SUMF_CNT_NULLS, 2>(a, d_args, first_chunk, last_chunk, stream);
  if (sig == SIG_SUMI_CNT_NULLS) return launch_dense_n<SIG_SUMI_CNT_NULLS, 2>(a, d_args, first_chunk, last_chunk, stream);
  if (a.plan.NACC <= 4) return launch_dense_n<0, 4>(a, d_args, first_chunk, last_chunk, stream);
  return launch_dense_n<0, MAX_ACC>(a, d_args, first_chunk, last_chunk, stream);
}

void launch_key_ranges(plan_dev const* d_plan, int nkeycols, int64_t nrows, int64_t sample, int64_t* out, hipStream_t stream)
{
  CUDF_EXPECTS(nkeycols >= 1 && nkeycols <= MAX_KU, "dense keys: 1 to 4 key columns");
  struct ranges_init { int64_t v[2 * MAX_KU]; };
  ranges_init init{};
  for (int c = 0; c < MAX_KU; ++c) {
    init.v[2 * c]     = INT64_MAX;
    init.v[2 * c + 1] = INT64_MIN;
  }
  hipLaunchKernelGGL(k_store_args<ranges_init>, dim3(1), dim3(1), 0, stream, init, reinterpret_cast<ranges_init*>(out));
  cudf::detail::prof::scope prof_{"estimate", stream};
  unsigned const grid = static_cast<unsigned>(std::clamp<int64_t>((sample + 1023) / 1024, 1, 1024));
  hipLaunchKernelGGL(k_key_ranges, dim3(grid), dim3(256), 0, stream, d_plan, nkeycols, nrows, sample, reinterpret_cast<long long*>(out));
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_key_range(plan_dev const* d_plan, int64_t nrows, int64_t sample, int is_signed, uint64_t* out, hipStream_t stream)
{
  struct range_init { uint64_t lo, hi; };
  range_init const init{is_signed ? static_cast<uint64_t>(INT64_MAX) : UINT64_MAX, is_signed ? static_cast<uint64_t>(INT64_MIN) : uint64_t{0}};
  hipLaunchKernelGGL(k_store_args<range_init>, dim3(1), dim3(1), 0, stream, init, reinterpret_cast<range_init*>(out));
  cudf::detail::prof::scope prof_{"estimate", stream};
  // (four sampled rows per thread: the stratum arithmetic is 128-bit division; one atomic pair per workgroup)
  unsigned const grid = static_cast<unsigned>(std::clamp<int64_t>((sample + 1023) / 1024, 1, 1024));
  if (is_signed) hipLaunchKernelGGL(k_key_range<true>, dim3(grid), dim3(256), 0, stream, d_plan, nrows, sample, out);
  else hipLaunchKernelGGL(k_key_range<false>, dim3(grid), dim3(256), 0, stream, d_plan, nrows, sample, out);
  CUDF_HIP_TRY(hipGetLastError());
}

}  // namespace cudf::groupby::detail
