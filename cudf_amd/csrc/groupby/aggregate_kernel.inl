// SPDX-License-Identifier: Apache-2.0
// The LDS hash-table aggregate kernel of the hash-groupby engine and its launch helpers. Included by the three
// translation units that instantiate it for one input mode each (aggregate_columns.hip, aggregate_raw.hip,
// aggregate_partial.hip): one translation unit with every instantiation took five minutes to compile.
#pragma once
#include "device_common.hpp"

namespace cudf::groupby::detail {
namespace {
// ------------------------------------------------------------------ K_aggregate
// (acc_identity / lds_merge / combine_values and the accumulator signatures: device_common.hpp)
// The LDS table is an open-addressing table probed in aligned BUCKETS of four slots: the four state words of a
// bucket are one ds_read_b128, a tag match names the one slot whose key words are worth reading, and slots of a
// bucket are claimed in order (occupied slots form a prefix; a key moves on to the next bucket only when its bucket
// is full). Two dependent LDS round trips resolve a row at any load factor the planner uses; slot-at-a-time linear
// probing needed as many dependent round trips as the longest displacement among the wave's rows.
__device__ __forceinline__ int home_bucket(uint64_t h, int cap)
{
  return static_cast<int>((static_cast<uint64_t>(static_cast<uint32_t>(h)) * static_cast<uint32_t>(cap >> 2)) >> 32);
}

// Finds or claims the slot of `key`, one slot at a time (tail rows, tag collisions): walking the slots in order from
// the home bucket's first slot visits the same buckets and claims the same first empty slot as the bucketed probe.
// Returns -1 if the table is saturated.
template <int KUT>
__device__ __forceinline__ int lds_find_or_insert(int KU, uint64_t const (&kmask)[KUT], uint32_t* st, uint64_t* keys,
                                                  int cap, uint64_t const (&key)[KUT], uint64_t h, uint32_t* nfilled,
                                                  int fill_limit, int32_t* overflow_flag)
{
  uint32_t const tag = tag_of(h);
  int slot           = 4 * home_bucket(h, cap);
  for (int probes = 0; probes < cap; ++probes) {
    uint32_t s = __hip_atomic_load(&st[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (s == ST_EMPTY) {
      uint32_t const old = atomicCAS(&st[slot], ST_EMPTY, ST_LOCKED);
      if (old == ST_EMPTY) {
#pragma unroll
        for (int u = 0; u < KUT; ++u)
          if (u < KU) keys[static_cast<uint32_t>(u * cap + slot)] = key[u] & kmask[u];
        // publish: key words first, then the tag (LDS executes a wave's accesses in order; the release fence
        // keeps the compiler from reordering and waits for the key stores)
        __hip_atomic_store(&st[slot], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t const n = atomicAdd(nfilled, 1u);
        if (static_cast<int>(n) >= fill_limit) *overflow_flag = 1;
        return slot;
      }
      s = old;
    }
    if (s == ST_LOCKED) {
      --probes;  // owner is publishing: re-read the same slot
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    if (s == tag) {
      bool eq = true;
#pragma unroll
      for (int u = 0; u < KUT; ++u)
        if (u < KU) eq = eq && (keys[static_cast<uint32_t>(u * cap + slot)] == (key[u] & kmask[u]));
      if (eq) return slot;
    }
    slot = slot + 1 == cap ? 0 : slot + 1;
    // the attempt is void once the table has overflowed: do not walk a saturated table to its end for every row
    if ((probes & 31) == 31 && __hip_atomic_load(overflow_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return -1;
  }
  *overflow_flag = 1;
  return -1;
}

// Slot of a key that is known to be in the table (second sweep of ARGMIN / ARGMAX); -1 if the table was saturated.
template <int KUT>
__device__ __forceinline__ int lds_lookup(int KU, uint64_t const (&kmask)[KUT], uint32_t const* st, uint64_t const* keys, int cap,
                                          uint64_t const (&key)[KUT], uint64_t h)
{
  uint32_t const tag = tag_of(h);
  int slot           = 4 * home_bucket(h, cap);
  for (int probes = 0; probes < cap; ++probes) {
    uint32_t const s = st[slot];
    if (s == ST_EMPTY) return -1;
    if (s == tag) {
      bool eq = true;
#pragma unroll
      for (int u = 0; u < KUT; ++u)
        if (u < KU) eq = eq && (keys[static_cast<uint32_t>(u * cap + slot)] == (key[u] & kmask[u]));
      if (eq) return slot;
    }
    slot = slot + 1 == cap ? 0 : slot + 1;
  }
  return -1;
}

// INPUT: agg_input. KUT: key units held in registers. PAYT: payload units of a RECORD prefetched into
// registers together with the key (0 = payload fetched lazily per accumulator: column input, wide records).
// NACCT: compile-time bound of the accumulator loop (descriptors sit in registers, statically indexed).
// EXACT: the input record has exactly KUT + PAYT units (a 16-byte record is one global_load_dwordx4).
// SIG: compile-time accumulator signature (0 = read the descriptors at run time). For the hot shapes every
// descriptor test folds away and the accumulate step is straight-line ds_* atomics: the generic form spends
// ~200 scalar instructions per 64 rows on descriptor branches and is bound by the CU's single scalar ALU.
// 12 bits per accumulator: op(4) | src(2) | pay+1 (3) | vbit+1 (3); accumulator count in bits 60..63.
template <int INPUT, int KUT, int PAYT, int NACCT, bool SIMPLE, bool EXACT, uint64_t SIG = 0>
__global__ void __launch_bounds__(1024, 4) k_aggregate(agg_args const* __restrict__ ap)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  agg_args const& a = *ap;
  plan_dev const& p = a.plan;
  int const cap     = a.geom.cap;
  constexpr bool STATIC_SIG = SIG != 0;
  int const KU = EXACT ? KUT : p.KU, NACC = STATIC_SIG ? sig_n(SIG) : p.NACC;
  uint64_t* keys = reinterpret_cast<uint64_t*>(lds_raw);                 // [KU][cap]
  __shared__ uint32_t s_nfilled, s_dump;
  __shared__ int32_t s_overflow;

  if (threadIdx.x == 0) {
    s_nfilled  = 0;
    s_dump     = 0;
    s_overflow = 0;
  }
  uint64_t kmask[KUT];
#pragma unroll
  for (int u = 0; u < KUT; ++u) kmask[u] = u < KU ? p.key_mask[u] : 0;
  // accumulator descriptors live in (scalar) registers for the whole kernel: no memory access per row
  int acc_op[NACCT], acc_src[NACCT], acc_pay[NACCT], acc_vbit[NACCT];
  // LDS layout: keys [KU][cap] u64 | accumulator q [cap] u64, or u32 for a COUNT | state words [cap] u32
  uint32_t acc_off[NACCT];
  bool acc_narrow[NACCT];
  uint32_t lds_off = static_cast<uint32_t>(KU) * static_cast<uint32_t>(cap) * 8u;
#pragma unroll
  for (int j = 0; j < NACCT; ++j) {
    uint32_t const w = j < NACC ? reinterpret_cast<uint32_t const*>(p.acc)[j] : 0u;
    acc_op[j]        = STATIC_SIG ? sig_op(SIG, j) : static_cast<int8_t>(w);
    acc_src[j]       = STATIC_SIG ? sig_src(SIG, j) : static_cast<int8_t>(w >> 8);
    acc_pay[j]       = STATIC_SIG ? sig_pay(SIG, j) : static_cast<int8_t>(w >> 16);
    acc_vbit[j]      = STATIC_SIG ? sig_vbit(SIG, j) : static_cast<int8_t>(w >> 24);
    acc_narrow[j]    = j < NACC && acc_is_narrow(acc_op[j], acc_src[j]);
    acc_off[j]       = lds_off;
    if (j < NACC) lds_off += static_cast<uint32_t>(cap) * (acc_narrow[j] ? 4u : 8u);
  }
  uint32_t* st = reinterpret_cast<uint32_t*>(lds_raw + lds_off);  // [cap], 16-byte aligned (cap is a multiple of 4)
  auto acc_off_rt = [&](int q) {  // offset of accumulator q for a run-time q (ARGMIN / ARGMAX sweep)
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < NACCT; ++j)
      if (j == q) o = acc_off[j];
    return o;
  };
  auto acc64 = [&](int q) { return reinterpret_cast<uint64_t*>(lds_raw + acc_off[q]); };
  auto acc32 = [&](int q) { return reinterpret_cast<uint32_t*>(lds_raw + acc_off[q]); };
  for (int s = threadIdx.x; s < cap; s += blockDim.x) st[s] = ST_EMPTY;
#pragma unroll
  for (int q = 0; q < NACCT; ++q) {
    if (q >= NACC) break;
    if (acc_narrow[q]) {
      for (int s = threadIdx.x; s < cap; s += blockDim.x) acc32(q)[s] = 0;
    } else {
      uint64_t const id = acc_identity(acc_op[q]);
      for (int s = threadIdx.x; s < cap; s += blockDim.x) acc64(q)[s] = id;
    }
  }
  __syncthreads();

  int const item  = blockIdx.x;
  int const RU    = KU + p.NPAY;  // raw record units
  int const PU    = KU + NACC;    // partial record units
  int const U     = EXACT ? (KUT + PAYT) : (INPUT == IN_RAW_RECORDS ? RU : PU);
  int const fill_limit = a.geom.fill_limit;
  int const flags_unit = p.flags_unit, flags_hi = p.flags_hi;
  uint64_t const* records = a.records;

  // payload unit of accumulator q for row r (prefetched record units, a plain / generic column, or the record in HBM)
  auto payload_of = [&](int q, int64_t r, uint64_t const (&pay)[PAYT > 0 ? PAYT : 1]) -> uint64_t {
    uint64_t value = 0;
    if constexpr (PAYT > 0) {
#pragma unroll
      for (int w = 0; w < PAYT; ++w)
        if (acc_pay[q] == w) value = pay[w];
    } else if constexpr (INPUT == IN_COLUMNS) {
      if constexpr (SIMPLE) value = gload(p.simple_base[KU + acc_pay[q]] + r);
      else value = col_load_acc_bits(p.cols[p.nkeycols + acc_pay[q]], r);
    } else {
      value = gload(records + r * U + KU + acc_pay[q]);
    }
    return value;
  };
  // accumulators of one row whose LDS slot is known
  auto accumulate = [&](int64_t r, int slot, uint64_t const (&pay)[PAYT > 0 ? PAYT : 1], uint32_t valvalid) {
    int last_pay   = -1;
    uint64_t value = 0;
#pragma unroll
    for (int q = 0; q < NACCT; ++q) {
      if (q >= NACC) break;
      uint64_t* tgt = acc64(q) + slot;  // (COUNT accumulators: acc32(q) + slot)
      if ((acc_src[q] == SRC_ARG_IDX || acc_src[q] == SRC_ARG_IDX_OF_MAX)) continue;  // filled by the second sweep
      if constexpr (INPUT == IN_PARTIAL_RECORDS) {
        uint64_t v;
        if constexpr (PAYT > 0) v = q < PAYT ? pay[q < PAYT ? q : 0] : 0;
        else v = gload(records + r * U + KU + q);
        if (acc_narrow[q]) atomicAdd(acc32(q) + slot, static_cast<uint32_t>(v));
        else lds_merge(tgt, acc_op[q], v);
      } else {
        if (acc_src[q] == SRC_ONE) {
          atomicAdd(acc32(q) + slot, 1u);
          continue;
        }
        bool const valid = acc_vbit[q] < 0 || ((valvalid >> acc_vbit[q]) & 1u);
        if (!valid) continue;
        if (acc_src[q] == SRC_ONE_IF_VALID) {
          atomicAdd(acc32(q) + slot, 1u);
          continue;
        }
        if (acc_pay[q] != last_pay) {
          value    = payload_of(q, r, pay);
          last_pay = acc_pay[q];
        }
        lds_merge(tgt, acc_op[q], acc_contribution(acc_src[q], acc_op[q], value));
      }
    }
  };
  // Wave-combined accumulate (raw rows): the lanes in `mine` all found the SAME slot. Same-address LDS atomics serialise
  // (a key with percents of the rows kept one workgroup busy for 17 ms per 12M rows); here each accumulator is reduced
  // across the wave (6 exchange steps) and lane `leader` issues one atomic.
  auto accumulate_wave = [&](int64_t r, int slot, bool mine, bool leader, uint64_t const (&pay)[PAYT > 0 ? PAYT : 1], uint32_t valvalid) {
#pragma unroll
    for (int q = 0; q < NACCT; ++q) {
      if (q >= NACC) break;
      if ((acc_src[q] == SRC_ARG_IDX || acc_src[q] == SRC_ARG_IDX_OF_MAX)) continue;
      bool const valid = mine && (acc_src[q] == SRC_ONE || acc_vbit[q] < 0 || ((valvalid >> acc_vbit[q]) & 1u));
      unsigned long long const vm = __ballot(valid);
      if (vm == 0) continue;  // (wave-uniform)
      if (acc_narrow[q]) {
        if (leader) atomicAdd(acc32(q) + slot, static_cast<uint32_t>(__popcll(vm)));
        continue;
      }
      uint64_t v = acc_identity(acc_op[q]);
      if (valid) {
        uint64_t const value = payload_of(q, r, pay);
        v                    = acc_contribution(acc_src[q], acc_op[q], value);
      }
      if (acc_op[q] == ANY_U64) {  // any contributing row will do: the first valid lane's
        v = static_cast<uint64_t>(__shfl(static_cast<unsigned long long>(v), __ffsll(static_cast<long long>(vm)) - 1));
      } else {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
          v = combine_values(acc_op[q], v, static_cast<uint64_t>(__shfl_xor(static_cast<unsigned long long>(v), off)));
      }
      if (leader) lds_merge(acc64(q) + slot, acc_op[q], v);
    }
  };
  auto hash_of = [&](uint64_t const (&key)[KUT]) {
    if constexpr (KUT == 1) {
      // one key unit: ONE 64-bit multiply, halves swapped (the table's bucket comes from the low 32 bits of the result = the high
      // half of the product, which every key bit reaches; the tag from its middle). The partition a row is in was chosen from
      // mix64's top bits, a different function, so the buckets of a partition's keys stay spread.
      uint64_t const x = (key[0] & kmask[0]) * 0x9e3779b97f4a7c15ull;
      return ((x >> 32) | (x << 32)) ^ (x >> 15);
    } else {
      uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
      for (int u = 0; u < KUT; ++u)
        if (u < KU) h = mix64(h ^ (key[u] & kmask[u]));
      return h;
    }
  };
  // one row, unbatched (tail rows)
  auto process = [&](int64_t r, uint64_t const (&key)[KUT], uint64_t const (&pay)[PAYT > 0 ? PAYT : 1], uint32_t valvalid) {
    uint64_t const h = hash_of(key);
    int const slot   = lds_find_or_insert<KUT>(KU, kmask, st, keys, cap, key, h, &s_nfilled, fill_limit, &s_overflow);
    if (slot >= 0) accumulate(r, slot, pay, valvalid);
  };
  // Generic columns (narrow types, nulls) whose record shape is known at compile time: records of a whole batch are built
  // column-at-a-time (descriptors decoded once per batch, typed loads issued back to back) - row-at-a-time record
  // building with lazily loaded payloads ran the single-pass path at 13.5 ms per 1B rows against 3.1 ms for plain columns.
  constexpr bool BATCH_COLS = INPUT == IN_COLUMNS && !SIMPLE && EXACT && PAYT > 0;
  units_local<BATCH_COLS ? KUT + PAYT : 1> L{};
  if constexpr (BATCH_COLS) L.load(p, KUT + PAYT);
  // loads one row; false if the row is dropped
  auto load_row = [&](int64_t r, uint64_t (&key)[KUT], uint64_t (&pay)[PAYT > 0 ? PAYT : 1], uint32_t& valvalid) -> bool {
    valvalid = 0xffffffffu;
    if constexpr (INPUT == IN_COLUMNS) {
      // the payload is loaded WITH the key (a lazy load at accumulate time is a second exposed HBM round trip per batch)
      if constexpr (SIMPLE && PAYT > 0) {
#pragma unroll
        for (int v = 0; v < PAYT; ++v) pay[v] = gload(p.simple_base[KU + v] + r);
      }
      if constexpr (BATCH_COLS) {  // a single row through the batched record builder (tail rows)
        int64_t row[1] = {r};
        bool live[1]   = {true};
        uint64_t rec[1][KUT + PAYT];
        uint32_t vv[1];
        batch_units_local<1, KUT + PAYT>(L, KUT + PAYT, row, live, rec, vv);
#pragma unroll
        for (int u = 0; u < KUT; ++u) key[u] = rec[0][u];
#pragma unroll
        for (int v = 0; v < PAYT; ++v) pay[v] = rec[0][KUT + v];
        valvalid = vv[0];
        return live[0];
      } else {
        return build_key_units<KUT, SIMPLE>(p, r, key, valvalid);
      }
    } else if constexpr (EXACT && KUT == 1 && PAYT == 1) {
      u64x2 const v = gload(reinterpret_cast<u64x2 const*>(records) + r);
      key[0]        = v.x;
      pay[0]        = v.y;
      // (a key of at most 4 bytes carries the validity flags of a nullable value in its spare half)
      if (INPUT == IN_RAW_RECORDS && flags_unit >= 0) {
        uint64_t const w = flags_unit == 0 ? v.x : v.y;
        valvalid         = static_cast<uint32_t>(flags_hi ? (w >> 32) : w);
      }
      return true;
    } else {
#pragma unroll
      for (int u = 0; u < KUT; ++u) key[u] = (u < KU) ? gload(records + r * U + u) : 0;
      if constexpr (PAYT > 0) {
#pragma unroll
        for (int v = 0; v < PAYT; ++v) pay[v] = gload(records + r * U + KU + v);
      }
      if (INPUT == IN_RAW_RECORDS && flags_unit >= 0)
        valvalid = gload(reinterpret_cast<uint32_t const*>(records + r * U + flags_unit) + flags_hi);
      return true;
    }
  };

  int nsrc = 1, src0 = item;
  if (a.seg == SEG_STRIDED) {
    src0 = item * a.fan;
    nsrc = min(a.fan, a.nsrc - src0);
    // an upstream kernel (optimistic partition, previous merge round) gave up: its counts are not valid
    if (*a.overflow != 0) nsrc = 0;
  }
  constexpr int R = (KUT + PAYT <= 2) ? 4 : 2;  // rows in flight per thread
  // Work is dealt to WAVES in batches of W = R*64 consecutive records. One big segment (a partition, a row chunk):
  // the waves interleave batches. Many short segments (the per-slice regions of an optimistic partition, the
  // partial tables of a merge round): each wave takes whole segments, so short segments still run batched.
  int64_t const B     = blockDim.x;
  constexpr int64_t W = R * 64;
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  // Bucketed probe of R rows per lane together: per round, the four state words of each pending row's bucket (one
  // ds_read_b128 each, issued back to back), then the key words of the slot whose tag matches, then the verdict.
  bool dead = false;  // this workgroup's table overflowed: the call will be redone, stop working on it
  auto probe_batch = [&](uint64_t const (&key)[R][KUT], uint64_t const (&h)[R], int (&bkt)[R], int (&sl)[R], uint32_t pend) {
    int const nbkt = cap >> 2;
    int guard      = 0;
    while (pend != 0) {
      asm volatile("" ::: "memory");  // the state words change under us: read them again every round
      u32x4 sw[R];
      int cand[R];
      uint64_t kc[R][KUT];
#pragma unroll
      for (int k = 0; k < R; ++k)
        if ((pend >> k) & 1u) sw[k] = *reinterpret_cast<u32x4 const*>(st + 4 * bkt[k]);
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (!((pend >> k) & 1u)) continue;
        uint32_t const tag = tag_of(h[k]);
        cand[k] = sw[k].x == tag ? 0 : (sw[k].y == tag ? 1 : (sw[k].z == tag ? 2 : (sw[k].w == tag ? 3 : -1)));
        if (cand[k] >= 0) {
#pragma unroll
          for (int u = 0; u < KUT; ++u) kc[k][u] = u < KU ? keys[static_cast<uint32_t>(u * cap + 4 * bkt[k] + cand[k])] : 0;
        }
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (!((pend >> k) & 1u)) continue;
        uint32_t const tag = tag_of(h[k]);
        if (cand[k] >= 0) {
          bool eq = true;
#pragma unroll
          for (int u = 0; u < KUT; ++u)
            if (u < KU) eq = eq && (kc[k][u] == (key[k][u] & kmask[u]));
          // a different key with the same tag (2^-30 per occupied slot): resolve this row slot by slot
          sl[k] = eq ? 4 * bkt[k] + cand[k]
                     : lds_find_or_insert<KUT>(KU, kmask, st, keys, cap, key[k], h[k], &s_nfilled, fill_limit, &s_overflow);
          pend &= ~(1u << k);
          continue;
        }
        bool const locked = sw[k].x == ST_LOCKED || sw[k].y == ST_LOCKED || sw[k].z == ST_LOCKED || sw[k].w == ST_LOCKED;
        if (locked) continue;  // a slot of this bucket is being published (it may be this key): read again
        int const e = sw[k].x == ST_EMPTY ? 0 : (sw[k].y == ST_EMPTY ? 1 : (sw[k].z == ST_EMPTY ? 2 : (sw[k].w == ST_EMPTY ? 3 : -1)));
        if (e < 0) {  // full of other keys
          bkt[k] = bkt[k] + 1 == nbkt ? 0 : bkt[k] + 1;
          continue;
        }
        int const c = 4 * bkt[k] + e;
        if (atomicCAS(&st[c], ST_EMPTY, ST_LOCKED) == ST_EMPTY) {
#pragma unroll
          for (int u = 0; u < KUT; ++u)
            if (u < KU) keys[static_cast<uint32_t>(u * cap + c)] = key[k][u] & kmask[u];
          __hip_atomic_store(&st[c], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          uint32_t const nf = atomicAdd(&s_nfilled, 1u);
          if (static_cast<int>(nf) >= fill_limit) s_overflow = 1;
          sl[k] = c;
          pend &= ~(1u << k);
        }
        // lost the race (possibly to an earlier row of this lane): read the bucket again
      }
      // A saturated table would cost every batch `cap` rounds (seconds per call when a skewed sample under-sized the
      // tables): once the overflow flag is up the attempt is void, so the workgroup stops probing altogether.
      if (++guard > 24) {
        if (__hip_atomic_load(&s_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
          pend = 0;
          dead = true;
        } else if (guard > cap + 64) {  // saturated table
          s_overflow = 1;
          pend       = 0;
          dead       = true;
        }
      }
    }
  };
  // few long sources (the regions of a second partition level): every wave works on every source
  bool const multi = a.seg == SEG_STRIDED && nsrc >= nwaves;
  for (int sidx = multi ? wave : 0; sidx < nsrc; sidx += multi ? nwaves : 1) {
    if (dead || __hip_atomic_load(&s_overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;
    int64_t begin, end;
    if (a.seg == SEG_ROW_CHUNKS) {
      begin = static_cast<int64_t>(item) * a.chunk;
      end   = min(a.nrows, begin + a.chunk);
    } else if (a.seg == SEG_OFFSETS) {
      begin = a.offsets[item];
      end   = a.offsets[item + 1];
    } else {
      begin = static_cast<int64_t>(src0 + sidx) * a.src_stride;
      end   = begin + min<int64_t>(max(a.src_count[src0 + sidx], 0), a.src_stride);
    }
    int64_t const nbatches = (end - begin) / W;
    // main loop: R full rows per lane, all loads issued before the LDS work
    // (narrow records: the NEXT batch's loads are issued before the LDS work of this one - a wave walks its
    // batches one after the other and 4 waves per SIMD do not hide a full HBM round trip per batch)
    constexpr bool PREFETCH = (KUT + PAYT <= 2) && (INPUT != IN_COLUMNS || (SIMPLE && PAYT > 0));
    int64_t const bstep     = multi ? 1 : nwaves;
    uint64_t nkey[R][KUT];
    uint64_t npay[R][PAYT > 0 ? PAYT : 1];
    uint32_t nvalvalid[R];
    bool nkeep[R];
    if constexpr (PREFETCH) {
      int64_t const b0 = multi ? 0 : wave;
      if (b0 < nbatches) {
#pragma unroll
        for (int k = 0; k < R; ++k) nkeep[k] = load_row(begin + b0 * W + k * 64 + lane, nkey[k], npay[k], nvalvalid[k]);
      }
    }
    for (int64_t b = multi ? 0 : wave; b < nbatches; b += bstep) {
      int64_t const base = begin + b * W;
      uint64_t key[R][KUT];
      uint64_t pay[R][PAYT > 0 ? PAYT : 1];
      uint32_t valvalid[R];
      bool keep[R];
      if constexpr (PREFETCH) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
#pragma unroll
          for (int u = 0; u < KUT; ++u) key[k][u] = nkey[k][u];
#pragma unroll
          for (int u = 0; u < (PAYT > 0 ? PAYT : 1); ++u) pay[k][u] = npay[k][u];
          valvalid[k] = nvalvalid[k];
          keep[k]     = nkeep[k];
        }
        if (b + bstep < nbatches) {
#pragma unroll
          for (int k = 0; k < R; ++k)
            nkeep[k] = load_row(begin + (b + bstep) * W + k * 64 + lane, nkey[k], npay[k], nvalvalid[k]);
        }
      } else if constexpr (BATCH_COLS) {
        // (no software pipeline here: a second register set for the next batch spills - measured slower)
        int64_t row[R];
        uint64_t rec[R][KUT + PAYT];
#pragma unroll
        for (int k = 0; k < R; ++k) {
          row[k]  = base + k * 64 + lane;
          keep[k] = true;
        }
        batch_units_local<R, KUT + PAYT>(L, KUT + PAYT, row, keep, rec, valvalid);
#pragma unroll
        for (int k = 0; k < R; ++k) {
#pragma unroll
          for (int u = 0; u < KUT; ++u) key[k][u] = rec[k][u];
#pragma unroll
          for (int v = 0; v < PAYT; ++v) pay[k][v] = rec[k][KUT + v];
        }
      } else {
#pragma unroll
        for (int k = 0; k < R; ++k) keep[k] = load_row(base + k * 64 + lane, key[k], pay[k], valvalid[k]);
      }
      // Bucketed probe of all R rows together: per round, the four state words of each pending row's bucket (one
      // ds_read_b128 each, issued back to back), then the key words of the slot whose tag matches, then the verdict.
      uint64_t h[R];
      int bkt[R], sl[R];
      uint32_t pend   = 0;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        h[k]   = hash_of(key[k]);
        bkt[k] = home_bucket(h[k], cap);
        sl[k]  = -1;
        if (keep[k]) pend |= 1u << k;
      }
      probe_batch(key, h, bkt, sl, pend);
      if (dead) break;
      // Most of the wave on ONE slot (a heavy key, sorted or clustered rows, very few groups): reduce across the wave
      // and issue one atomic per accumulator. Ordinary data pays one readfirstlane + ballot per 256 rows: the batch's
      // first row set decides whether the other sets are looked at.
      // (testing only every fourth batch, with a sticky flag, measured SLOWER: 3.61 vs 3.51 ms on C2's aggregate)
      bool crowded = false;
      if constexpr (INPUT != IN_PARTIAL_RECORDS) {
        // (from 8 lanes on the first row's slot: runs of 64 equal keys that straddle a row set leave lane 0's key as few as
        // a handful of lanes - with the old threshold of 32 such batches went row by row into two slots: 1B rows in runs of 64
        // took 12.6 ms in this pass against 5.3 ms for runs of 1000)
        int const first = __builtin_amdgcn_readfirstlane(sl[0]);
        crowded         = __popcll(__ballot(keep[0] && sl[0] == first && first >= 0)) >= 8;
      }
      bool combined[R];  // (one inlined copy of accumulate: a second one cost the generic shapes 8-10 %)
#pragma unroll
      for (int k = 0; k < R; ++k) combined[k] = false;
      if (crowded) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
          // up to three slots per row set, most crowded lanes first in row order (clustered rows: a row set of 64 consecutive rows
          // holds two or three runs); what is left goes row by row
          for (int round = 0; round < 3; ++round) {
            bool const act                = keep[k] && sl[k] >= 0 && !combined[k];
            unsigned long long const am   = __ballot(act);
            if (am == 0) break;
            int const lead_slot           = __shfl(sl[k], __ffsll(static_cast<long long>(am)) - 1);
            bool const mine               = act && sl[k] == lead_slot;
            unsigned long long const same = __ballot(mine);
            if (__popcll(same) < 8) break;
            accumulate_wave(base + k * 64 + lane, lead_slot, mine, lane == __ffsll(static_cast<long long>(same)) - 1, pay[k], valvalid[k]);
            combined[k] = combined[k] || mine;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < R; ++k)
        if (keep[k] && sl[k] >= 0 && !combined[k]) accumulate(base + k * 64 + lane, sl[k], pay[k], valvalid[k]);
    }
    if (dead) break;
    // tail: the < W records after the last full batch (a masked partial batch instead measured 25-30 % slower)
    int64_t const tail_begin = begin + nbatches * W;
    for (int64_t r = tail_begin + (multi ? lane : static_cast<int>(threadIdx.x)); r < end; r += multi ? 64 : B) {
      uint64_t key[KUT];
      uint64_t pay[PAYT > 0 ? PAYT : 1];
      uint32_t valvalid;
      if (load_row(r, key, pay, valvalid)) process(r, key, pay, valvalid);
    }
  }
  __syncthreads();
  // ---- ARGMIN / ARGMAX: every group's extreme value is final; a second sweep over the same rows takes the smallest
  // row index among the rows that attain it
  if (p.narg > 0 && s_overflow == 0) {
    for (int sidx = 0; sidx < nsrc; ++sidx) {
      int64_t begin, end;
      if (a.seg == SEG_ROW_CHUNKS) {
        begin = static_cast<int64_t>(item) * a.chunk;
        end   = min(a.nrows, begin + a.chunk);
      } else if (a.seg == SEG_OFFSETS) {
        begin = a.offsets[item];
        end   = a.offsets[item + 1];
      } else {
        begin = static_cast<int64_t>(src0 + sidx) * a.src_stride;
        end   = begin + min<int64_t>(max(a.src_count[src0 + sidx], 0), a.src_stride);
      }
      // four rows per thread in flight: the loads of all four are issued before the first lookup (one row per iteration left the
      // sweep waiting for a memory round trip per row: 29 ms of a 41 ms ARGMIN + ARGMAX call at 1B rows, profiles/r4_exotic_shapes.txt)
      constexpr int SW = 4;
      for (int64_t r0 = begin + threadIdx.x; r0 < end; r0 += static_cast<int64_t>(SW) * blockDim.x) {
        uint64_t key[SW][KUT];
        uint64_t pay[SW][PAYT > 0 ? PAYT : 1];
        uint32_t valvalid[SW];
        bool live[SW];
#pragma unroll
        for (int j = 0; j < SW; ++j) {
          int64_t const r = r0 + static_cast<int64_t>(j) * blockDim.x;
          live[j]         = r < end && load_row(r, key[j], pay[j], valvalid[j]);
        }
#pragma unroll
        for (int j = 0; j < SW; ++j) {
          if (!live[j]) continue;
          int64_t const r = r0 + static_cast<int64_t>(j) * blockDim.x;
          int const slot  = lds_lookup<KUT>(KU, kmask, st, keys, cap, key[j], hash_of(key[j]));
          if (slot < 0) continue;
          for (int i = 0; i < p.narg; ++i) {
            int const qv = p.arg[i].valacc, qi = p.arg[i].idxacc;
            uint64_t v, rowid = 0;
            if constexpr (INPUT == IN_PARTIAL_RECORDS) {
              v     = gload(records + r * U + KU + qv);
              rowid = gload(records + r * U + KU + qi);
              if (rowid == static_cast<uint64_t>(INT64_MAX)) continue;  // that partial saw no valid row
            } else {
              int const vb = p.acc[qi].valid_bit, pw = p.acc[qi].pay;
              if (vb >= 0 && !((valvalid[j] >> vb) & 1u)) continue;
              if constexpr (INPUT == IN_COLUMNS) {
                v     = col_load_acc_bits(p.cols[p.nkeycols + pw], r);
                rowid = static_cast<uint64_t>(r);
              } else {
                v = gload(records + r * U + KU + pw);  // (the line load_row just read)
              }
            }
            uint64_t const best = reinterpret_cast<uint64_t const*>(lds_raw + acc_off_rt(qv))[slot];
            bool const same     = p.arg[i].is_float
                                    ? __longlong_as_double(static_cast<long long>(v)) == __longlong_as_double(static_cast<long long>(best))
                                    : v == best;
            if (same) {
              // (records: the row id is read only for the few rows that attain the extreme)
              if constexpr (INPUT != IN_PARTIAL_RECORDS && INPUT != IN_COLUMNS) rowid = gload(records + r * U + p.rowid_unit) & 0xffffffffull;
              atomicMin(reinterpret_cast<long long*>(lds_raw + acc_off_rt(qi)) + slot, static_cast<long long>(rowid));
            }
          }
        }
      }
    }
    __syncthreads();
  }
  // dump the table as compact partial records
  uint64_t* out = a.out_records + static_cast<int64_t>(item) * cap * PU;
  for (int s = threadIdx.x; s < cap; s += blockDim.x) {
    if (st[s] >= 2) {
      uint32_t const pos = atomicAdd(&s_dump, 1u);
      uint64_t* o        = out + static_cast<int64_t>(pos) * PU;
      for (int u = 0; u < KU; ++u) gstore(o + u, keys[static_cast<size_t>(u) * cap + s]);
#pragma unroll
      for (int q = 0; q < NACCT; ++q) {
        if (q >= NACC) break;
        gstore(o + KU + q, acc_narrow[q] ? static_cast<uint64_t>(acc32(q)[s]) : acc64(q)[s]);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.out_count[item] = static_cast<int32_t>(s_dump);
    if (s_overflow) atomicOr(a.overflow, 2);  // bit 1: a table overflowed (bit 0: a region of an optimistic partition)
  }
}

}  // namespace

template <int INPUT, int KUT, int PAYT, int NACCT, bool SIMPLE, bool EXACT, uint64_t SIG = 0>
static void launch_aggregate_n(agg_args const& a, agg_args* d_args, hipStream_t stream)
{
  auto const lds = aggregate_lds_bytes(a.plan, a.geom);
  static std::once_flag attr_once;  // (the API is re-entrant across objects: two threads may launch this kernel first)
  std::call_once(attr_once, [] { allow_full_lds(reinterpret_cast<void const*>(&k_aggregate<INPUT, KUT, PAYT, NACCT, SIMPLE, EXACT, SIG>)); });
  hipLaunchKernelGGL(k_store_args<agg_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"aggregate", stream};
  hipLaunchKernelGGL((k_aggregate<INPUT, KUT, PAYT, NACCT, SIMPLE, EXACT, SIG>), dim3(a.nitems), dim3(a.geom.block), lds,
                     stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}


template <int INPUT, int KUT, int PAYT, bool SIMPLE, bool EXACT>
static void launch_aggregate_t(agg_args const& a, agg_args* d_args, hipStream_t stream)
{
  if (a.plan.NACC <= 2) return launch_aggregate_n<INPUT, KUT, PAYT, 2, SIMPLE, EXACT>(a, d_args, stream);
  if (a.plan.NACC <= 4) return launch_aggregate_n<INPUT, KUT, PAYT, 4, SIMPLE, EXACT>(a, d_args, stream);
  return launch_aggregate_n<INPUT, KUT, PAYT, MAX_ACC, SIMPLE, EXACT>(a, d_args, stream);
}


template <int INPUT>
static void launch_aggregate_records(agg_args const& a, agg_args* d_args, hipStream_t stream)
{
  int const KU   = a.plan.KU;
  int const npay = INPUT == IN_RAW_RECORDS ? a.plan.NPAY : a.plan.NACC;
  // hot signatures: descriptors folded at compile time
  if constexpr (INPUT == IN_RAW_RECORDS) {
    uint64_t const sig = plan_sig(a.plan);
    if (KU == 1 && npay == 1 && a.plan.flags_unit < 0) {
      if (sig == SIG_SUMF_CNT) return launch_aggregate_n<INPUT, 1, 1, 2, false, true, SIG_SUMF_CNT>(a, d_args, stream);
      if (sig == SIG_SUMI_CNT) return launch_aggregate_n<INPUT, 1, 1, 2, false, true, SIG_SUMI_CNT>(a, d_args, stream);
      if (sig == SIG_SUMF) return launch_aggregate_n<INPUT, 1, 1, 2, false, true, SIG_SUMF>(a, d_args, stream);
      if (sig == SIG_SUMI) return launch_aggregate_n<INPUT, 1, 1, 2, false, true, SIG_SUMI>(a, d_args, stream);
      if (sig == SIG_CNT) return launch_aggregate_n<INPUT, 1, 1, 2, false, true, SIG_CNT>(a, d_args, stream);
    }
    if (KU == 1 && (npay == 1 || npay == 2)) {  // nullable value: the validity flags ride in the key's spare half or in their own unit
      if (sig == SIG_SUMF_CNT_NULLS)
        return npay == 1 ? launch_aggregate_n<INPUT, 1, 1, 2, false, true, SIG_SUMF_CNT_NULLS>(a, d_args, stream)
                         : launch_aggregate_n<INPUT, 1, 2, 2, false, true, SIG_SUMF_CNT_NULLS>(a, d_args, stream);
      if (sig == SIG_SUMI_CNT_NULLS)
        return npay == 1 ? launch_aggregate_n<INPUT, 1, 1, 2, false, true, SIG_SUMI_CNT_NULLS>(a, d_args, stream)
                         : launch_aggregate_n<INPUT, 1, 2, 2, false, true, SIG_SUMI_CNT_NULLS>(a, d_args, stream);
    }
    if (KU == 2 && npay == 1 && sig == SIG_MEAN_MIN_MAX_F_NULLS)
      return launch_aggregate_n<INPUT, 2, 1, 4, false, true, SIG_MEAN_MIN_MAX_F_NULLS>(a, d_args, stream);
    if (KU == 2 && npay == 1 && a.plan.flags_unit < 0 && sig == SIG_MEAN_MIN_MAX_F)
      return launch_aggregate_n<INPUT, 2, 1, 4, false, true, SIG_MEAN_MIN_MAX_F>(a, d_args, stream);
    if (KU == 1 && npay == 1 && a.plan.flags_unit < 0 && sig == SIG_MEAN_MIN_MAX_F)
      return launch_aggregate_n<INPUT, 1, 1, 4, false, true, SIG_MEAN_MIN_MAX_F>(a, d_args, stream);
  }
  // exact shapes get the payload prefetched with the key; everything else fetches it lazily
  if (KU == 1 && npay == 1) return launch_aggregate_t<INPUT, 1, 1, false, true>(a, d_args, stream);
  if (KU == 1 && npay == 2) return launch_aggregate_t<INPUT, 1, 2, false, true>(a, d_args, stream);
  if (KU == 2 && npay == 1) return launch_aggregate_t<INPUT, 2, 1, false, true>(a, d_args, stream);
  if (KU == 2 && npay == 4) return launch_aggregate_t<INPUT, 2, 4, false, true>(a, d_args, stream);
  if (KU <= 1) return launch_aggregate_t<INPUT, 1, 0, false, false>(a, d_args, stream);
  if (KU <= 2) return launch_aggregate_t<INPUT, 2, 0, false, false>(a, d_args, stream);
  return launch_aggregate_t<INPUT, 4, 0, false, false>(a, d_args, stream);
}


}  // namespace cudf::groupby::detail
