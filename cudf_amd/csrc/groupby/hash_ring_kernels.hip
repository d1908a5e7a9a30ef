// SPDX-License-Identifier: Apache-2.0
// gfx950 aggregate kernel of the hash-groupby path for SPARSE single 8-byte keys behind the ring scatter (engine.hpp
// k64_agg_args; the scatter is dense_multi_kernels.hip launch_hash_ring_scatter: two 8-byte streams - key, value - partitioned on
// the top bits of the key hash into 256 partitions, whole aligned 128-byte granules only, two barriers per 2048-row tile).
//
// The open-addressing LDS table of aggregate_kernel.inl keeps a state word per slot (empty / locked / tag) and resolves a row in
// two dependent LDS round trips (state words of a bucket, then the key words of the matching slot); 24 bytes per group for
// SUM + COUNT, 6784 slots, so 1M groups need 512 partitions - more than the rings can feed. Here the slot's state IS its key word:
// one 8-byte key unit, an empty marker, claimed with one 64-bit LDS compare-and-swap. 20 bytes per group (8140 slots: 1M groups
// in 256 tables at load 0.48) and ONE round trip per probe: the row's slot is read, and it is the row's key, or empty, or
// another key (next slot). A row whose key equals the marker goes to a slot of its own behind the table.
// The walk over a partition's regions (batches of 4 x 64 records, indices resolved first, loads back to back, the next batch in
// flight while this one is accumulated) is the dense aggregate's (dense_kernels.hip k_aggregate_dense).
// Replaces the reference's cuco::static_set insert + global atomics (cpp/src/groupby/hash/compute_global_memory_aggs.cuh:74-187,
// compute_groupby.cu:65-78) for such keys.
// STATUS (round 3, profiles/r3_sparse_ring.txt): correct (tests/test_groupby_gpu.py test_sparse_keys_*) but NOT faster - 1B rows on
// 1M sparse keys: ring scatter 7.08 ms (write-combining scatter into 512 partitions: 7.45) and this aggregate 5.62 ms (tagged
// tables at load 0.36: 3.45): 13.0 against 11.2 ms per call. With 256 rings of 32 records the tile is 2048 rows (the dense path's
// 128 rings of 64 carry 4096), and at load 0.48 a batch of 256 rows almost always holds a row that needs a second and a third
// bucket. Off by default (CUDF_AMD_GB_HASH_RING=1 turns it on); kept because the pieces are the ones a better geometry would use.
#include "device_common.hpp"

namespace cudf::groupby::detail {
namespace {

constexpr uint64_t K64_EMPTY = 0x9e3779b97f4a7c15ull;  // (any value will do: the key that equals it has a slot of its own)

__host__ __device__ inline uint32_t k64_slot_bytes(plan_dev const& p)
{
  uint32_t b = 8;
  for (int q = 0; q < p.NACC; ++q) b += acc_is_narrow(p.acc[q].op, p.acc[q].src) ? 4u : 8u;
  return b;
}

template <uint64_t SIG, int NACCT>
__global__ void __launch_bounds__(1024, 4) k_aggregate_k64(k64_agg_args const* __restrict__ ap)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ uint32_t s_dump, s_nfilled, s_marker_key;
  __shared__ int32_t s_overflow;
  k64_agg_args const& a = *ap;
  plan_dev const& p     = a.plan;
  constexpr bool STATIC_SIG = SIG != 0;
  int const NACC = STATIC_SIG ? sig_n(SIG) : p.NACC;
  int const cap = a.cap, item = blockIdx.x, B = blockDim.x;
  uint32_t const nslots = static_cast<uint32_t>(cap) + 2u;  // slot `cap`: the key that equals the marker; one more keeps arrays 16-byte aligned
  uint64_t* keys = reinterpret_cast<uint64_t*>(lds_raw);
  int acc_op[NACCT], acc_src[NACCT], acc_vbit[NACCT];
  uint32_t acc_off[NACCT];
  bool acc_narrow[NACCT];
  uint32_t off8 = nslots * 8u, off4 = nslots * 8u;  // 8-byte arrays in accumulator order, then the 4-byte ones
#pragma unroll
  for (int j = 0; j < NACCT; ++j) {
    uint32_t const w = j < NACC ? reinterpret_cast<uint32_t const*>(p.acc)[j] : 0u;
    acc_op[j]        = STATIC_SIG ? sig_op(SIG, j) : static_cast<int8_t>(w);
    acc_src[j]       = STATIC_SIG ? sig_src(SIG, j) : static_cast<int8_t>(w >> 8);
    acc_vbit[j]      = STATIC_SIG ? sig_vbit(SIG, j) : static_cast<int8_t>(w >> 24);
    acc_narrow[j]    = j < NACC && acc_is_narrow(acc_op[j], acc_src[j]);
    if (j < NACC && !acc_narrow[j]) off4 += nslots * 8u;
  }
#pragma unroll
  for (int j = 0; j < NACCT; ++j) {
    if (j >= NACC) { acc_off[j] = 0; continue; }
    if (acc_narrow[j]) { acc_off[j] = off4; off4 += nslots * 4u; }
    else { acc_off[j] = off8; off8 += nslots * 8u; }
  }
  auto acc64 = [&](int q) { return reinterpret_cast<uint64_t*>(lds_raw + acc_off[q]); };
  auto acc32 = [&](int q) { return reinterpret_cast<uint32_t*>(lds_raw + acc_off[q]); };
  for (uint32_t s = threadIdx.x; s < nslots; s += B) keys[s] = K64_EMPTY;
#pragma unroll
  for (int q = 0; q < NACCT; ++q) {
    if (q >= NACC) break;
    if (acc_narrow[q]) {
      for (uint32_t s = threadIdx.x; s < nslots; s += B) acc32(q)[s] = 0;
    } else {
      uint64_t const id = acc_identity(acc_op[q]);
      for (uint32_t s = threadIdx.x; s < nslots; s += B) acc64(q)[s] = id;
    }
  }
  if (threadIdx.x == 0) {
    s_dump       = 0;
    s_nfilled    = 0;
    s_marker_key = 0;
    s_overflow   = 0;
  }
  __syncthreads();

  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = B >> 6;
  uint64_t const kmask0 = p.key_mask[0];
  int const fill_limit  = a.fill_limit;
  // an upstream scatter gave up (region overflow): its counts are not valid
  bool const upstream_ok = *a.overflow == 0;
  // ---- the partition's regions as one virtual record range per wave (dense_kernels.hip k_aggregate_dense)
  int const nsl     = a.slices;
  bool const shared = nsl <= 64;
  int const g       = shared ? nsl : (nsl + nwaves - 1) / nwaves;  // regions in this wave's range, <= 64
  int const r_base  = shared ? 0 : wave * g;
  int32_t cnt       = 0;
  if (upstream_ok && lane < g && r_base + lane < nsl)
    cnt = min(max(a.region_count[static_cast<int64_t>(item) * a.slices + r_base + lane], 0), static_cast<int32_t>(a.region_cap));
  int32_t pend = cnt;  // inclusive prefix: end of region `lane` in the virtual range
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int32_t const t = __shfl_up(pend, o);
    if (lane >= o) pend += t;
  }
  int32_t const total = __builtin_amdgcn_readlane(pend, 63);
  int64_t const rcap  = a.region_cap;
  int64_t const rec0  = (static_cast<int64_t>(item) * a.slices + r_base) * rcap;
  uint64_t const* rec_key = a.rec_key + rec0;
  uint64_t const* rec_val = a.rec_val + rec0;
  int rcur = 0;  // wave-uniform: first region that may hold the next virtual record
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
    return static_cast<int64_t>(reg) * rcap + (v - start);
  };
  constexpr int R = 4;
  int32_t const vstep = shared ? R * 64 * nwaves : R * 64;
  auto load_batch = [&](int32_t v0, uint64_t (&key)[R], uint64_t (&val)[R], bool (&act)[R]) {
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
      key[k] = gload(rec_key + ri[k]);
      val[k] = gload(rec_val + ri[k]);
    }
  };
  bool dead = false;  // this workgroup's table overflowed: the call will be redone, stop working on it
  // Slots of R rows together. The table is probed in aligned BUCKETS of four slots (32 bytes: two ds_read_b128 issued back to
  // back): the row's key is one of the four, or the bucket has an empty slot (claimed in order: occupied slots form a prefix, a
  // key moves on to the next bucket only when its bucket is full), or the bucket is full of other keys (5 % of the keys at load
  // 0.48). One LDS round trip resolves almost every row; walking slot by slot, a batch took as many dependent round trips as the
  // LONGEST displacement among the wave's 256 rows (the sparse-key C2 call: 16.6 ms instead of 11).
  int const nbkt = cap >> 2;
  auto find_slots = [&](uint64_t const (&key)[R], bool const (&act)[R], int (&sl)[R]) {
    uint32_t bkt[R];
    uint32_t pendm = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      uint64_t const kk = key[k] & kmask0;
      uint64_t const h  = mix64(0x9e3779b97f4a7c15ull ^ kk);
      bkt[k] = static_cast<uint32_t>((static_cast<uint64_t>(static_cast<uint32_t>(h)) * static_cast<uint32_t>(nbkt)) >> 32);
      sl[k]  = -1;
      if (act[k]) {
        if (kk == K64_EMPTY) {  // the key that equals the empty marker: its own slot behind the table
          sl[k]        = cap;
          s_marker_key = 1;
        } else {
          pendm |= 1u << k;
        }
      }
    }
    int guard = 0;
    while (pendm != 0) {
      asm volatile("" ::: "memory");  // the key words change under us: read them again every round
      u64x2 lo2[R], hi2[R];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if ((pendm >> k) & 1u) {
          lo2[k] = *reinterpret_cast<u64x2 const*>(keys + 4u * bkt[k]);
          hi2[k] = *reinterpret_cast<u64x2 const*>(keys + 4u * bkt[k] + 2u);
        }
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (!((pendm >> k) & 1u)) continue;
        uint64_t const kk = key[k] & kmask0;
        int const hit = lo2[k].x == kk ? 0 : (lo2[k].y == kk ? 1 : (hi2[k].x == kk ? 2 : (hi2[k].y == kk ? 3 : -1)));
        if (hit >= 0) {
          sl[k] = static_cast<int>(4u * bkt[k]) + hit;
          pendm &= ~(1u << k);
          continue;
        }
        int const e = lo2[k].x == K64_EMPTY ? 0 : (lo2[k].y == K64_EMPTY ? 1 : (hi2[k].x == K64_EMPTY ? 2 : (hi2[k].y == K64_EMPTY ? 3 : -1)));
        if (e < 0) {  // full of other keys
          bkt[k] = bkt[k] + 1 == static_cast<uint32_t>(nbkt) ? 0u : bkt[k] + 1;
          continue;
        }
        uint32_t const c = 4u * bkt[k] + static_cast<uint32_t>(e);
        uint64_t const old = atomicCAS(reinterpret_cast<unsigned long long*>(keys + c), static_cast<unsigned long long>(K64_EMPTY),
                                       static_cast<unsigned long long>(kk));
        if (old == K64_EMPTY) {
          uint32_t const nf = atomicAdd(&s_nfilled, 1u);
          if (static_cast<int>(nf) >= fill_limit) s_overflow = 1;
        }
        if (old == K64_EMPTY || old == kk) {
          sl[k] = static_cast<int>(c);
          pendm &= ~(1u << k);
        }
        // (lost the slot to another key - possibly to an earlier row of this lane: read the bucket again)
      }
      // the attempt is void once the table has overflowed: do not walk a saturated table to its end for every row
      if (++guard > 24) {
        if (__hip_atomic_load(&s_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0 || guard > nbkt + 64) {
          s_overflow = 1;
          pendm      = 0;
          dead       = true;
        }
      }
    }
  };
  auto accumulate = [&](int slot, uint64_t value) {
#pragma unroll
    for (int q = 0; q < NACCT; ++q) {
      if (q >= NACC) break;
      if (acc_narrow[q]) {  // (a plain plan has no NULL values: SRC_ONE and SRC_ONE_IF_VALID both count every row)
        atomicAdd(acc32(q) + slot, 1u);
        continue;
      }
      lds_merge(acc64(q) + slot, acc_op[q], acc_contribution(acc_src[q], acc_op[q], value));
    }
  };
  {
    uint64_t keyA[R], valA[R], keyB[R], valB[R];
    bool actA[R], actB[R];
    int sl[R];
    int32_t v0 = shared ? wave * R * 64 : 0;
    if (v0 < total) {
      load_batch(v0, keyA, valA, actA);
      for (;;) {
        load_batch(v0 + vstep, keyB, valB, actB);  // (nothing is loaded past `total`)
        find_slots(keyA, actA, sl);
        if (dead) break;
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (sl[k] >= 0) accumulate(sl[k], valA[k]);
        v0 += vstep;
        if (v0 >= total) break;
        load_batch(v0 + vstep, keyA, valA, actA);
        find_slots(keyB, actB, sl);
        if (dead) break;
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (sl[k] >= 0) accumulate(sl[k], valB[k]);
        v0 += vstep;
        if (v0 >= total) break;
      }
    }
  }
  __syncthreads();
  // ---- occupied slots -> partial records [key | accumulators] (k_finalize reads them)
  int const PU  = 1 + NACC;
  uint64_t* out = a.out_records + static_cast<int64_t>(item) * (cap + 1) * PU;
  for (int s = threadIdx.x; s <= cap; s += B) {
    uint64_t const k = keys[s];
    bool const occupied = s == cap ? s_marker_key != 0 : k != K64_EMPTY;
    if (!occupied) continue;
    uint32_t const pos = atomicAdd(&s_dump, 1u);
    uint64_t* o        = out + static_cast<int64_t>(pos) * PU;
    gstore(o, s == cap ? (K64_EMPTY & kmask0) : k);
#pragma unroll
    for (int q = 0; q < NACCT; ++q) {
      if (q >= NACC) break;
      gstore(o + 1 + q, acc_narrow[q] ? static_cast<uint64_t>(acc32(q)[s]) : acc64(q)[s]);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.out_count[item] = static_cast<int32_t>(s_dump);
    if (s_overflow) atomicOr(a.overflow, 2);  // bit 1: a table overflowed (bit 0: a region of the scatter)
  }
}

template <uint64_t SIG, int NACCT>
void launch_k64_t(k64_agg_args const& a, k64_agg_args const* d_args, std::size_t lds, hipStream_t stream)
{
  static std::once_flag attr_once;
  std::call_once(attr_once, [] { allow_full_lds(reinterpret_cast<void const*>(&k_aggregate_k64<SIG, NACCT>)); });
  cudf::detail::prof::scope prof_{"aggregate", stream};
  hipLaunchKernelGGL((k_aggregate_k64<SIG, NACCT>), dim3(a.nitems), dim3(1024), lds, stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

}  // namespace

int k64_table_slots(plan_dev const& plan, std::size_t lds_bytes)
{
  int64_t const slots = static_cast<int64_t>(lds_bytes) / k64_slot_bytes(plan) - 2;  // (+ the marker key's slot and one of padding)
  return static_cast<int>(std::clamp<int64_t>(slots & ~int64_t{3}, 0, 16384));  // (buckets of four slots)
}

void store_args(k64_agg_args const& a, k64_agg_args* d_args, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<k64_agg_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_aggregate_k64(k64_agg_args const& a, k64_agg_args const* d_args, hipStream_t stream)
{
  CUDF_EXPECTS(a.plan.simple && a.plan.KU == 1 && a.plan.NPAY == 1 && a.plan.narg == 0 && a.plan.flags_unit < 0,
               "k64 aggregate: one plain 8-byte key column and one plain 8-byte value column");
  std::size_t const lds = static_cast<std::size_t>(a.cap + 2) * k64_slot_bytes(a.plan);
  CUDF_EXPECTS(a.cap >= 64 && a.cap % 4 == 0 && lds + 64 <= 160 * 1024 && a.slices >= 1 && a.slices <= 1024 && a.fill_limit >= 1 && a.fill_limit <= a.cap,
               "k64 aggregate: table geometry");
  for (int q = 0; q < a.plan.NACC; ++q) CUDF_EXPECTS(a.plan.acc[q].op != ANY_U64, "k64 aggregate: integer keys only");
  uint64_t const sig = plan_sig(a.plan);
  if (sig == SIG_SUMF_CNT) return launch_k64_t<SIG_SUMF_CNT, 2>(a, d_args, lds, stream);
  if (sig == SIG_SUMI_CNT) return launch_k64_t<SIG_SUMI_CNT, 2>(a, d_args, lds, stream);
  if (sig == SIG_SUMF) return launch_k64_t<SIG_SUMF, 2>(a, d_args, lds, stream);
  if (sig == SIG_SUMI) return launch_k64_t<SIG_SUMI, 2>(a, d_args, lds, stream);
  if (sig == SIG_CNT) return launch_k64_t<SIG_CNT, 2>(a, d_args, lds, stream);
  if (sig == SIG_MEAN_MIN_MAX_F) return launch_k64_t<SIG_MEAN_MIN_MAX_F, 4>(a, d_args, lds, stream);
  if (a.plan.NACC <= 4) return launch_k64_t<0, 4>(a, d_args, lds, stream);
  return launch_k64_t<0, MAX_ACC>(a, d_args, lds, stream);
}

}  // namespace cudf::groupby::detail
