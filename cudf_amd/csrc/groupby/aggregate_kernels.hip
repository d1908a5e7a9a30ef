// SPDX-License-Identifier: Apache-2.0
// gfx950 aggregation kernels of the hash-groupby engine: LDS hash-table aggregate, finalize, cardinality estimate.
#include "device_common.hpp"

namespace cudf::groupby::detail {
namespace {
// ------------------------------------------------------------------ K_aggregate
__device__ __forceinline__ uint64_t acc_identity(int op)
{
  switch (op) {
    case MIN_I64: return static_cast<uint64_t>(INT64_MAX);
    case MIN_U64: return UINT64_MAX;
    case MIN_F64: return 0x7ff0000000000000ull;  // +inf
    case MAX_I64: return static_cast<uint64_t>(INT64_MIN);
    case MAX_U64: return 0;
    case MAX_F64: return 0xfff0000000000000ull;  // -inf
    case MUL_I64: return 1;
    case MUL_F64: return 0x3ff0000000000000ull;  // 1.0
    default: return 0;                            // ADD_I64 / ADD_F64
  }
}

__device__ __forceinline__ void lds_merge(uint64_t* slot, int op, uint64_t v)
{
  switch (op) {
    case ADD_I64: atomicAdd(reinterpret_cast<unsigned long long*>(slot), static_cast<unsigned long long>(v)); break;
    case ADD_F64: atomicAdd(reinterpret_cast<double*>(slot), __longlong_as_double(static_cast<long long>(v))); break;
    case MIN_I64: atomicMin(reinterpret_cast<long long*>(slot), static_cast<long long>(v)); break;
    case MIN_U64: atomicMin(reinterpret_cast<unsigned long long*>(slot), static_cast<unsigned long long>(v)); break;
    case MAX_I64: atomicMax(reinterpret_cast<long long*>(slot), static_cast<long long>(v)); break;
    case MAX_U64: atomicMax(reinterpret_cast<unsigned long long*>(slot), static_cast<unsigned long long>(v)); break;
    case MIN_F64:
      __hip_atomic_fetch_min(reinterpret_cast<double*>(slot), __longlong_as_double(static_cast<long long>(v)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    case MAX_F64:
      __hip_atomic_fetch_max(reinterpret_cast<double*>(slot), __longlong_as_double(static_cast<long long>(v)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    case ANY_U64: *slot = v; break;  // any contributing row will do: a plain store
    case MUL_I64:
    case MUL_F64: {
      // no multiply atomic: compare-and-swap loop (ds_cmpst_rtn_b64), as the reference's product (device_atomics.cuh:226-337)
      unsigned long long* p = reinterpret_cast<unsigned long long*>(slot);
      unsigned long long old = *p, seen;
      do {
        seen = old;
        unsigned long long const next =
          op == MUL_I64 ? seen * static_cast<unsigned long long>(v)
                        : static_cast<unsigned long long>(__double_as_longlong(__longlong_as_double(static_cast<long long>(seen)) *
                                                                                __longlong_as_double(static_cast<long long>(v))));
        old = atomicCAS(p, seen, next);
      } while (old != seen);
      break;
    }
  }
}

// a (op) b on accumulator bit patterns - the wave-level counterpart of lds_merge (float min / max: a NaN never beats a
// number, as ds_min_f64 / ds_max_f64)
__device__ __forceinline__ uint64_t combine_values(int op, uint64_t a, uint64_t b)
{
  auto f = [](uint64_t x) { return __longlong_as_double(static_cast<long long>(x)); };
  auto u = [](double x) { return static_cast<uint64_t>(__double_as_longlong(x)); };
  switch (op) {
    case ADD_I64: return a + b;
    case ADD_F64: return u(f(a) + f(b));
    case MIN_I64: return static_cast<uint64_t>(min(static_cast<long long>(a), static_cast<long long>(b)));
    case MIN_U64: return min(a, b);
    case MAX_I64: return static_cast<uint64_t>(max(static_cast<long long>(a), static_cast<long long>(b)));
    case MAX_U64: return max(a, b);
    case MIN_F64: return u(fmin(f(a), f(b)));
    case MAX_F64: return u(fmax(f(a), f(b)));
    case MUL_I64: return a * b;
    case MUL_F64: return u(f(a) * f(b));
    default: return a;  // ANY_U64
  }
}

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
constexpr uint64_t sig_acc(int op, int src, int pay, int vbit)
{
  return static_cast<uint64_t>(op) | (static_cast<uint64_t>(src) << 4) | (static_cast<uint64_t>(pay + 1) << 6) |
         (static_cast<uint64_t>(vbit + 1) << 9);
}
constexpr uint64_t make_sig(int n, uint64_t a0 = 0, uint64_t a1 = 0, uint64_t a2 = 0, uint64_t a3 = 0)
{
  return (static_cast<uint64_t>(n) << 60) | a0 | (a1 << 12) | (a2 << 24) | (a3 << 36);
}
constexpr int sig_n(uint64_t s) { return static_cast<int>(s >> 60); }
constexpr int sig_op(uint64_t s, int q) { return static_cast<int>((s >> (12 * q)) & 0xf); }
constexpr int sig_src(uint64_t s, int q) { return static_cast<int>((s >> (12 * q + 4)) & 0x3); }
constexpr int sig_pay(uint64_t s, int q) { return static_cast<int>((s >> (12 * q + 6)) & 0x7) - 1; }
constexpr int sig_vbit(uint64_t s, int q) { return static_cast<int>((s >> (12 * q + 9)) & 0x7) - 1; }

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
  auto squared = [](int op, uint64_t v) -> uint64_t {
    if (op == ADD_F64) {
      double const x = __longlong_as_double(static_cast<long long>(v));
      return static_cast<uint64_t>(__double_as_longlong(x * x));
    }
    return v * v;
  };
  // accumulators of one row whose LDS slot is known
  auto accumulate = [&](int64_t r, int slot, uint64_t const (&pay)[PAYT > 0 ? PAYT : 1], uint32_t valvalid) {
    int last_pay   = -1;
    uint64_t value = 0;
#pragma unroll
    for (int q = 0; q < NACCT; ++q) {
      if (q >= NACC) break;
      uint64_t* tgt = acc64(q) + slot;  // (COUNT accumulators: acc32(q) + slot)
      if (acc_src[q] >= SRC_ARG_IDX) continue;  // filled by the second sweep
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
        lds_merge(tgt, acc_op[q], acc_src[q] == SRC_SQUARE ? squared(acc_op[q], value) : value);
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
      if (acc_src[q] >= SRC_ARG_IDX) continue;
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
        v                    = acc_src[q] == SRC_SQUARE ? squared(acc_op[q], value) : value;
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
    uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
    for (int u = 0; u < KUT; ++u)
      if (u < KU) h = mix64(h ^ (key[u] & kmask[u]));
    return h;
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
        int const first = __builtin_amdgcn_readfirstlane(sl[0]);
        crowded         = __popcll(__ballot(keep[0] && sl[0] == first && first >= 0)) >= 32;
      }
      bool combined[R];  // (one inlined copy of accumulate: a second one cost the generic shapes 8-10 %)
#pragma unroll
      for (int k = 0; k < R; ++k) combined[k] = false;
      if (crowded) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
          bool const act                = keep[k] && sl[k] >= 0;
          unsigned long long const am   = __ballot(act);
          if (am == 0) continue;
          int const lead_slot           = __shfl(sl[k], __ffsll(static_cast<long long>(am)) - 1);
          bool const mine               = act && sl[k] == lead_slot;
          unsigned long long const same = __ballot(mine);
          if (__popcll(same) < 16) continue;
          accumulate_wave(base + k * 64 + lane, lead_slot, mine, lane == __ffsll(static_cast<long long>(same)) - 1, pay[k], valvalid[k]);
          combined[k] = mine;
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
      for (int64_t r = begin + threadIdx.x; r < end; r += blockDim.x) {
        uint64_t key[KUT];
        uint64_t pay[PAYT > 0 ? PAYT : 1];
        uint32_t valvalid;
        if (!load_row(r, key, pay, valvalid)) continue;
        int const slot = lds_lookup<KUT>(KU, kmask, st, keys, cap, key, hash_of(key));
        if (slot < 0) continue;
        for (int i = 0; i < p.narg; ++i) {
          int const qv = p.arg[i].valacc, qi = p.arg[i].idxacc;
          uint64_t v, rowid;
          if constexpr (INPUT == IN_PARTIAL_RECORDS) {
            v     = gload(records + r * U + KU + qv);
            rowid = gload(records + r * U + KU + qi);
            if (rowid == static_cast<uint64_t>(INT64_MAX)) continue;  // that partial saw no valid row
          } else {
            int const vb = p.acc[qi].valid_bit, pw = p.acc[qi].pay;
            if (vb >= 0 && !((valvalid >> vb) & 1u)) continue;
            if constexpr (INPUT == IN_COLUMNS) {
              v     = col_load_acc_bits(p.cols[p.nkeycols + pw], r);
              rowid = static_cast<uint64_t>(r);
            } else {
              v     = gload(records + r * U + KU + pw);
              rowid = gload(records + r * U + p.rowid_unit) & 0xffffffffull;
            }
          }
          uint64_t const best = reinterpret_cast<uint64_t const*>(lds_raw + acc_off_rt(qv))[slot];
          bool const same     = p.arg[i].is_float
                                  ? __longlong_as_double(static_cast<long long>(v)) == __longlong_as_double(static_cast<long long>(best))
                                  : v == best;
          if (same)
            atomicMin(reinterpret_cast<long long*>(lds_raw + acc_off_rt(qi)) + slot, static_cast<long long>(rowid));
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

// ------------------------------------------------------------------ K_finalize
__device__ __forceinline__ void store_elem(void* base, int64_t i, int width, uint64_t bits)
{
  switch (width) {
    case 1: static_cast<uint8_t*>(base)[i] = static_cast<uint8_t>(bits); break;
    case 2: static_cast<uint16_t*>(base)[i] = static_cast<uint16_t>(bits); break;
    case 4: static_cast<uint32_t*>(base)[i] = static_cast<uint32_t>(bits); break;
    default: static_cast<uint64_t*>(base)[i] = bits;
  }
}

__global__ void __launch_bounds__(256) k_finalize(finalize_args const* __restrict__ fa, uint64_t const* records,
                                                  int64_t cap, int64_t const* prefix, int32_t nitems, int64_t total)
{
  plan_dev const& p     = fa->plan;
  finalize_dev const& f = fa->fin;
  int64_t const o = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  bool const live = o < total;
  int const PU    = p.KU + p.NACC;
  uint64_t const* rec = nullptr;
  if (live) {
    // largest item with prefix[item] <= o
    int lo = 0, hi = nitems - 1;
    while (lo < hi) {
      int const mid = (lo + hi + 1) >> 1;
      if (prefix[mid] <= o) lo = mid; else hi = mid - 1;
    }
    rec = records + (static_cast<int64_t>(lo) * cap + (o - prefix[lo])) * PU;
  }
  int const lane = threadIdx.x & 63;
  for (int c = 0; c < f.nout; ++c) {
    out_desc const d = f.out[c];
    bool valid       = live;
    uint64_t bits    = 0;
    if (live) {
      if (d.kind == OUT_KEY) {
        uint64_t const unit = rec[d.key_unit];
        bits                = d.key_full ? unit : (d.key_hi ? (unit >> 32) : (unit & 0xffffffffull));
        if (d.key_acc >= 0) {  // float key: the bits of a representative input row (carried as float64)
          uint64_t const raw = rec[p.KU + d.key_acc];
          bits = d.width == 4 ? __float_as_uint(static_cast<float>(__longlong_as_double(static_cast<long long>(raw)))) : raw;
        }
        if (d.key_null_bit >= 0) {
          uint64_t const kn = rec[d.keynulls_unit];
          uint32_t const w  = d.keynulls_hi ? static_cast<uint32_t>(kn >> 32) : static_cast<uint32_t>(kn);
          valid             = !((w >> d.key_null_bit) & 1u);
        }
      } else {
        uint64_t const a0 = rec[p.KU + d.a0];
        if (d.valid_acc >= 0) valid = static_cast<int64_t>(rec[p.KU + d.valid_acc]) > 0;
        if (d.kind == OUT_COUNT) {
          bits = a0;
        } else if (d.kind == OUT_M2 || d.kind == OUT_VAR || d.kind == OUT_STD) {
          // reference groupby/common/m2_var_std.cu:48-60,152-187
          auto as_double = [&](uint64_t v) {
            return d.cls == cudf::detail::CLS_F64 ? __longlong_as_double(static_cast<long long>(v))
                                                  : static_cast<double>(static_cast<int64_t>(v));
          };
          int64_t const cnt = static_cast<int64_t>(rec[p.KU + d.a2]);
          double const ssq  = as_double(a0);
          double const sm   = as_double(rec[p.KU + d.a1]);
          double const m2   = cnt > 0 ? ssq - sm * sm / static_cast<double>(cnt) : 0.0;
          double out        = m2;
          if (d.kind != OUT_M2) {
            int64_t const df = cnt - d.ddof;
            valid            = cnt > 0 && df > 0;
            out              = valid ? m2 / static_cast<double>(df) : 0.0;
            if (d.kind == OUT_STD) out = sqrt(out);
          } else {
            valid = true;
          }
          bits = static_cast<uint64_t>(__double_as_longlong(out));
        } else if (d.kind == OUT_MEAN) {
          // MEAN = double(SUM) / COUNT_VALID as FLOAT64 (reference hash_compound_agg_finalizer.cu:92-133)
          double const s = d.cls == cudf::detail::CLS_F64 ? __longlong_as_double(static_cast<long long>(a0))
                           : d.cls == cudf::detail::CLS_UINT ? static_cast<double>(a0)
                                                             : static_cast<double>(static_cast<int64_t>(a0));
          double const n = static_cast<double>(static_cast<int64_t>(rec[p.KU + d.a1]));
          bits           = static_cast<uint64_t>(__double_as_longlong(valid ? s / n : 0.0));
        } else {  // OUT_ACC: accumulator class -> output type
          if (d.cls == cudf::detail::CLS_F64 && d.out_cls == cudf::detail::CLS_F32) {
            bits = __float_as_uint(static_cast<float>(__longlong_as_double(static_cast<long long>(a0))));
          } else {
            bits = a0;  // integers truncate to the output width; FLOAT64 passes through
          }
          if (!valid) bits = 0;
        }
      }
      store_elem(d.data, o, d.width, bits);
    }
    if (d.mask != nullptr) {
      unsigned long long const ballot      = __ballot(valid);  // valid implies live
      unsigned long long const live_ballot = __ballot(live);
      if (live && (lane & 31) == 0) d.mask[o >> 5] = static_cast<uint32_t>(ballot >> (lane & 32));
      int const nulls = __popcll(live_ballot & ~ballot);
      if (lane == 0 && nulls) atomicAdd(d.null_count, nulls);
    }
  }
}

// ------------------------------------------------------------------ K_estimate (linear counting on a sample)
// Row of sample i: one row out of each of `sample` equal strata, at a pseudo-random offset inside its stratum. (A fixed
// stride aliases with periodic keys: `row % 10M` sampled every 200th row shows 50,000 distinct keys, the tables were
// planned 200x too small and the call took seconds.)
__device__ __forceinline__ int64_t sample_row(int64_t i, int64_t nrows, int64_t sample)
{
  if (sample >= nrows) return i;
  int64_t const lo = static_cast<int64_t>((static_cast<__int128>(i) * nrows) / sample);
  int64_t const hi = static_cast<int64_t>((static_cast<__int128>(i + 1) * nrows) / sample);
  uint64_t const span = static_cast<uint64_t>(hi - lo);
  return span <= 1 ? lo : lo + static_cast<int64_t>(mix64(static_cast<uint64_t>(i) + 0x51ed270b35a3c1ull) % span);
}
__global__ void __launch_bounds__(256) k_estimate(plan_dev const* __restrict__ pp, int64_t nrows, int64_t sample,
                                                  uint32_t* bitmap, int32_t bits_log2, uint32_t* hot_buckets)
{
  plan_dev const& p = *pp;
  int64_t const i   = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= sample) return;
  int64_t const row = sample_row(i, nrows, sample);
  uint64_t key[MAX_KU];
  uint32_t vv;
  if (!build_key_units<MAX_KU, false>(p, row, key, vv)) return;
  uint64_t const h   = hash_key_units<MAX_KU>(p, key);
  uint32_t const bit = static_cast<uint32_t>(h >> (64 - bits_log2));
  // (same-address global atomics serialise - a constant key column made this pass take 16 ms: set a bit only if it is
  // not seen set, and add to a bucket counter once per distinct bucket of the wave)
  if (!((gload(bitmap + (bit >> 5)) >> (bit & 31)) & 1u)) atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
  // heavy-hitter search: rows per hash bucket over every fourth sampled row (scattered global atomics run at 24 G/s:
  // a quarter of the sample keeps this at ~10 us)
  bool pending = hot_buckets != nullptr && (i & 3) == 0;
  uint32_t const bucket = static_cast<uint32_t>(h >> 48);
  for (int round = 0; round < 8; ++round) {  // wave-aggregated: one atomic per distinct bucket, for the first 8 of them
    unsigned long long const todo = __ballot(pending);
    if (todo == 0) break;
    int const lead            = __ffsll(static_cast<long long>(todo)) - 1;
    uint32_t const lead_bkt   = __shfl(bucket, lead);
    unsigned long long const same = __ballot(pending && bucket == lead_bkt);
    if ((threadIdx.x & 63) == lead) atomicAdd(&hot_buckets[lead_bkt], static_cast<uint32_t>(__popcll(same)));
    if (bucket == lead_bkt) pending = false;
  }
  if (pending) atomicAdd(&hot_buckets[bucket], 1u);
}
// ------------------------------------------------------------------ K_distinct (HyperLogLog over ALL rows)
// Run after a table overflowed, i.e. when the sample misjudged the group count (skewed key frequencies: a sample of a
// Zipf-distributed column shows a twentieth of its keys). 2^14 registers in LDS per workgroup (ds_max_u32), merged into
// the global registers at the end: the pass streams the key columns once (1.3 ms per 1B int64 keys), standard error 0.8 %.
static_assert(HLL_REGISTERS == (1 << 14));
constexpr int HLL_LOG2 = 14;
__global__ void __launch_bounds__(1024) k_distinct(plan_dev const* __restrict__ pp, int64_t nrows, uint32_t* regs)
{
  __shared__ uint32_t s_regs[1 << HLL_LOG2];
  plan_dev const& p = *pp;
  for (int i = threadIdx.x; i < (1 << HLL_LOG2); i += blockDim.x) s_regs[i] = 0;
  __syncthreads();
  int64_t const stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t row = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; row < nrows; row += stride) {
    uint64_t key[MAX_KU];
    uint32_t vv;
    if (!build_key_units<MAX_KU, false>(p, row, key, vv)) continue;
    // (a second mix: the partition and table index bits of the engine's hash stay independent of the register choice)
    uint64_t const h   = mix64(hash_key_units<MAX_KU>(p, key) ^ 0xa0761d6478bd642full);
    uint32_t const idx = static_cast<uint32_t>(h >> (64 - HLL_LOG2));
    uint64_t const w   = (h << HLL_LOG2) | (uint64_t{1} << (HLL_LOG2 - 1));  // rank is at most 64 - HLL_LOG2 + 1
    uint32_t const rho = static_cast<uint32_t>(__clzll(static_cast<long long>(w))) + 1;
    if (s_regs[idx] < rho) atomicMax(&s_regs[idx], rho);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (1 << HLL_LOG2); i += blockDim.x) {
    uint32_t const r = s_regs[i];
    if (r != 0 && gload(regs + i) < r) atomicMax(&regs[i], r);
  }
}

// ---- heavy hitters in the sample (plain 8-byte key): bucket counts, exact counts of the keys of crowded buckets, selection
constexpr uint64_t HOT_SENTINEL = ~uint64_t{0};
__device__ __forceinline__ bool hot_sample_key(plan_dev const& p, int64_t nrows, int64_t sample, uint64_t& key, uint64_t& h)
{
  int64_t const i = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) * 4;  // the rows k_estimate counted
  if (i >= sample) return false;
  int64_t const row = sample_row(i, nrows, sample);
  key = gload(p.simple_base[0] + row) & p.key_mask[0];
  h   = mix64(0x9e3779b97f4a7c15ull ^ key);
  return true;
}
__global__ void __launch_bounds__(256) k_hot_any(uint32_t const* buckets, uint32_t min_count, uint32_t* crowded)
{
  int const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < HOT_BUCKETS && buckets[i] >= min_count) *crowded = 1;
}
__global__ void __launch_bounds__(256) k_hot_collect(plan_dev const* __restrict__ pp, int64_t nrows, int64_t sample,
                                                     uint32_t const* buckets, uint32_t const* crowded, uint32_t min_count,
                                                     uint64_t* tkeys, uint32_t* tcounts)
{
  if (*crowded == 0) return;  // no bucket reached the threshold: nothing to collect
  uint64_t key = 0, h = 0;
  bool pending = hot_sample_key(*pp, nrows, sample, key, h) && buckets[h >> 48] >= min_count && key != HOT_SENTINEL;
  uint32_t mine = 1;  // rows this lane inserts for (wave-aggregated: the first lane of each distinct key of the wave)
  for (int round = 0; round < 8; ++round) {
    unsigned long long const todo = __ballot(pending && mine == 1);
    if (todo == 0) break;
    int const lead          = __ffsll(static_cast<long long>(todo)) - 1;
    uint64_t const lead_key = __shfl(static_cast<unsigned long long>(key), lead);
    unsigned long long const same = __ballot(pending && mine == 1 && key == lead_key);
    if (pending && mine == 1 && key == lead_key) {
      if (static_cast<int>(threadIdx.x & 63) == lead) mine = static_cast<uint32_t>(__popcll(same)) + 1;  // +1: marks "leader"
      else pending = false;
    }
  }
  if (!pending) return;
  uint32_t const add = mine > 1 ? mine - 1 : 1;
  uint32_t slot      = static_cast<uint32_t>(h >> 20) & (HOT_TABLE - 1);
  for (int probe = 0; probe < 64; ++probe) {
    unsigned long long const cur = atomicCAS(reinterpret_cast<unsigned long long*>(tkeys + slot), HOT_SENTINEL, key);
    if (cur == HOT_SENTINEL || cur == key) {
      atomicAdd(&tcounts[slot], add);
      return;
    }
    slot = (slot + 1) & (HOT_TABLE - 1);
  }
}

__global__ void __launch_bounds__(256) k_popcount(uint32_t const* bitmap, int64_t nwords, uint32_t* out)
{
  int64_t i     = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  uint32_t acc  = 0;
  for (; i < nwords; i += static_cast<int64_t>(gridDim.x) * blockDim.x) acc += __popc(bitmap[i]);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

}  // namespace

int aggregate_slot_bytes(plan_dev const& plan)
{
  int b = 8 * plan.KU + 4;
  for (int q = 0; q < plan.NACC; ++q) b += acc_is_narrow(plan.acc[q].op, plan.acc[q].src) ? 4 : 8;
  return b;
}
std::size_t aggregate_lds_bytes(plan_dev const& plan, agg_geom const& g)
{
  return static_cast<std::size_t>(g.cap) * static_cast<std::size_t>(aggregate_slot_bytes(plan));
}

template <int INPUT, int KUT, int PAYT, int NACCT, bool SIMPLE, bool EXACT, uint64_t SIG = 0>
static void launch_aggregate_n(agg_args const& a, agg_args* d_args, hipStream_t stream)
{
  auto const lds = aggregate_lds_bytes(a.plan, a.geom);
  static bool attr_set = false;
  if (!attr_set) {
    allow_full_lds(reinterpret_cast<void const*>(&k_aggregate<INPUT, KUT, PAYT, NACCT, SIMPLE, EXACT, SIG>));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_store_args<agg_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"aggregate", stream};
  hipLaunchKernelGGL((k_aggregate<INPUT, KUT, PAYT, NACCT, SIMPLE, EXACT, SIG>), dim3(a.nitems), dim3(a.geom.block), lds,
                     stream, d_args);
  CUDF_HIP_TRY(hipGetLastError());
}

// Signature of a plan's accumulators (0 if it does not fit the static encoding).
static uint64_t plan_sig(plan_dev const& p)
{
  if (p.NACC < 1 || p.NACC > 4) return 0;
  uint64_t a[4] = {0, 0, 0, 0};
  for (int q = 0; q < p.NACC; ++q) {
    auto const& d = p.acc[q];
    if (d.pay > 5 || d.valid_bit > 5 || d.src > 3) return 0;
    a[q] = sig_acc(d.op, d.src, d.pay, d.valid_bit);
  }
  return make_sig(p.NACC, a[0], a[1], a[2], a[3]);
}
// hot signatures with their own instantiation
constexpr uint64_t SIG_SUMF_CNT = make_sig(2, sig_acc(ADD_F64, SRC_VALUE, 0, -1), sig_acc(ADD_I64, SRC_ONE, -1, -1));
constexpr uint64_t SIG_SUMI_CNT = make_sig(2, sig_acc(ADD_I64, SRC_VALUE, 0, -1), sig_acc(ADD_I64, SRC_ONE, -1, -1));
constexpr uint64_t SIG_SUMF     = make_sig(1, sig_acc(ADD_F64, SRC_VALUE, 0, -1));
constexpr uint64_t SIG_SUMI     = make_sig(1, sig_acc(ADD_I64, SRC_VALUE, 0, -1));
constexpr uint64_t SIG_CNT      = make_sig(1, sig_acc(ADD_I64, SRC_ONE, -1, -1));
// SUM + COUNT_VALID (and MEAN) of a NULLABLE column: both accumulators test the row's validity bit
constexpr uint64_t SIG_SUMF_CNT_NULLS = make_sig(2, sig_acc(ADD_F64, SRC_VALUE, 0, 0), sig_acc(ADD_I64, SRC_ONE_IF_VALID, 0, 0));
constexpr uint64_t SIG_SUMI_CNT_NULLS = make_sig(2, sig_acc(ADD_I64, SRC_VALUE, 0, 0), sig_acc(ADD_I64, SRC_ONE_IF_VALID, 0, 0));
// C4: MEAN + MIN + MAX of a nullable float64 column -> SUM, COUNT_VALID, MIN, MAX
constexpr uint64_t SIG_MEAN_MIN_MAX_F_NULLS =
  make_sig(4, sig_acc(ADD_F64, SRC_VALUE, 0, 0), sig_acc(ADD_I64, SRC_ONE_IF_VALID, 0, 0), sig_acc(MIN_F64, SRC_VALUE, 0, 0),
           sig_acc(MAX_F64, SRC_VALUE, 0, 0));

// the same without nulls: SUM, COUNT_ALL (as the count of MEAN), MIN, MAX
constexpr uint64_t SIG_MEAN_MIN_MAX_F =
  make_sig(4, sig_acc(ADD_F64, SRC_VALUE, 0, -1), sig_acc(ADD_I64, SRC_ONE, -1, -1), sig_acc(MIN_F64, SRC_VALUE, 0, -1),
           sig_acc(MAX_F64, SRC_VALUE, 0, -1));

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

void launch_aggregate(agg_args const& a, agg_args* d_args, hipStream_t stream)
{
  if (a.nitems == 0) return;
  int const KU = a.plan.KU;
  if (a.input == IN_RAW_RECORDS) return launch_aggregate_records<IN_RAW_RECORDS>(a, d_args, stream);
  if (a.input == IN_PARTIAL_RECORDS) return launch_aggregate_records<IN_PARTIAL_RECORDS>(a, d_args, stream);
  bool const simple = a.plan.simple;
  if (simple && KU == 1 && a.plan.NPAY == 1) {  // one plain key column, one plain value column
    uint64_t const sig = plan_sig(a.plan);
    if (sig == SIG_SUMF_CNT) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, true, true, SIG_SUMF_CNT>(a, d_args, stream);
    if (sig == SIG_SUMI_CNT) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, true, true, SIG_SUMI_CNT>(a, d_args, stream);
    if (sig == SIG_SUMF) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, true, true, SIG_SUMF>(a, d_args, stream);
    if (sig == SIG_SUMI) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, true, true, SIG_SUMI>(a, d_args, stream);
    return launch_aggregate_t<IN_COLUMNS, 1, 1, true, true>(a, d_args, stream);
  }
  if (!simple && a.plan.NPAY >= 1 && a.plan.NPAY <= 2 && KU >= 1 && KU <= 2 && a.plan.ncols <= MAX_LOCAL_COLS) {  // generic columns, known record shape
    uint64_t const sig = plan_sig(a.plan);
    if (KU == 1 && a.plan.NPAY == 1) {
      if (sig == SIG_SUMF_CNT) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, false, true, SIG_SUMF_CNT>(a, d_args, stream);
      if (sig == SIG_SUMI_CNT) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, false, true, SIG_SUMI_CNT>(a, d_args, stream);
      if (sig == SIG_SUMF_CNT_NULLS) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, false, true, SIG_SUMF_CNT_NULLS>(a, d_args, stream);
      if (sig == SIG_SUMI_CNT_NULLS) return launch_aggregate_n<IN_COLUMNS, 1, 1, 2, false, true, SIG_SUMI_CNT_NULLS>(a, d_args, stream);
      return launch_aggregate_t<IN_COLUMNS, 1, 1, false, true>(a, d_args, stream);
    }
    if (KU == 1 && a.plan.NPAY == 2) {
      if (sig == SIG_SUMF_CNT_NULLS) return launch_aggregate_n<IN_COLUMNS, 1, 2, 2, false, true, SIG_SUMF_CNT_NULLS>(a, d_args, stream);
      if (sig == SIG_SUMI_CNT_NULLS) return launch_aggregate_n<IN_COLUMNS, 1, 2, 2, false, true, SIG_SUMI_CNT_NULLS>(a, d_args, stream);
      return launch_aggregate_t<IN_COLUMNS, 1, 2, false, true>(a, d_args, stream);
    }
    if (KU == 2 && a.plan.NPAY == 1) return launch_aggregate_t<IN_COLUMNS, 2, 1, false, true>(a, d_args, stream);
    return launch_aggregate_t<IN_COLUMNS, 2, 2, false, true>(a, d_args, stream);
  }
  if (KU <= 1) simple ? launch_aggregate_t<IN_COLUMNS, 1, 0, true, false>(a, d_args, stream) : launch_aggregate_t<IN_COLUMNS, 1, 0, false, false>(a, d_args, stream);
  else if (KU <= 2) simple ? launch_aggregate_t<IN_COLUMNS, 2, 0, true, false>(a, d_args, stream) : launch_aggregate_t<IN_COLUMNS, 2, 0, false, false>(a, d_args, stream);
  else simple ? launch_aggregate_t<IN_COLUMNS, 4, 0, true, false>(a, d_args, stream) : launch_aggregate_t<IN_COLUMNS, 4, 0, false, false>(a, d_args, stream);
}

void launch_finalize(finalize_args const& a, finalize_args* d_args, uint64_t const* records, int64_t cap,
                     int64_t const* prefix, int32_t nitems, int64_t total, hipStream_t stream)
{
  if (total == 0) return;
  int const block = 256;
  int64_t const grid = (total + block - 1) / block;
  hipLaunchKernelGGL(k_store_args<finalize_args>, dim3(1), dim3(1), 0, stream, a, d_args);
  cudf::detail::prof::scope prof_{"finalize", stream};
  hipLaunchKernelGGL(k_finalize, dim3(static_cast<unsigned>(grid)), dim3(block), 0, stream, d_args, records, cap, prefix,
                     nitems, total);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_hot_keys(plan_dev const* d_plan, int64_t nrows, int64_t sample, uint32_t min_count, uint32_t* buckets,
                     uint64_t* table_keys, uint32_t* table_counts, hipStream_t stream)
{
  // table_counts holds HOT_TABLE counters followed by the one-word "some bucket is crowded" flag
  CUDF_HIP_TRY(hipMemsetAsync(table_counts, 0, (HOT_TABLE + 1) * sizeof(uint32_t), stream));
  CUDF_HIP_TRY(hipMemsetAsync(table_keys, 0xff, HOT_TABLE * sizeof(uint64_t), stream));
  uint32_t* crowded   = table_counts + HOT_TABLE;
  unsigned const grid = static_cast<unsigned>(((sample + 3) / 4 + 255) / 256);
  cudf::detail::prof::scope prof_{"estimate", stream};
  hipLaunchKernelGGL(k_hot_any, dim3(HOT_BUCKETS / 256), dim3(256), 0, stream, buckets, min_count, crowded);
  // a bucket holds sample / 4 / 65536 counted keys on average (4 of a 1M-row sample): only crowded ones can hide a
  // heavy hitter, and their keys are counted exactly
  hipLaunchKernelGGL(k_hot_collect, dim3(grid), dim3(256), 0, stream, d_plan, nrows, sample, buckets, crowded, min_count, table_keys,
                     table_counts);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_distinct_count(plan_dev const& plan, plan_dev* d_plan, int64_t nrows, uint32_t* regs, hipStream_t stream)
{
  CUDF_HIP_TRY(hipMemsetAsync(regs, 0, HLL_REGISTERS * sizeof(uint32_t), stream));
  hipLaunchKernelGGL(k_store_args<plan_dev>, dim3(1), dim3(1), 0, stream, plan, d_plan);
  cudf::detail::prof::scope prof_{"distinct_count", stream};
  unsigned const grid = static_cast<unsigned>(std::clamp<int64_t>((nrows + 4095) / 4096, 1, 512));
  hipLaunchKernelGGL(k_distinct, dim3(grid), dim3(1024), 0, stream, d_plan, nrows, regs);
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_estimate(plan_dev const& plan, plan_dev* d_plan, int64_t nrows, int64_t sample, uint32_t* bitmap,
                     int32_t bitmap_bits_log2, uint32_t* d_bits_set, uint32_t* hot_buckets, hipStream_t stream)
{
  if (hot_buckets != nullptr) CUDF_HIP_TRY(hipMemsetAsync(hot_buckets, 0, HOT_BUCKETS * sizeof(uint32_t), stream));
  int64_t const nwords = (int64_t{1} << bitmap_bits_log2) / 32;
  CUDF_HIP_TRY(hipMemsetAsync(bitmap, 0, nwords * 4, stream));
  CUDF_HIP_TRY(hipMemsetAsync(d_bits_set, 0, 4, stream));
  int const block = 256;
  hipLaunchKernelGGL(k_store_args<plan_dev>, dim3(1), dim3(1), 0, stream, plan, d_plan);
  cudf::detail::prof::scope prof_{"estimate", stream};
  hipLaunchKernelGGL(k_estimate, dim3(static_cast<unsigned>((sample + block - 1) / block)), dim3(block), 0, stream, d_plan,
                     nrows, sample, bitmap, bitmap_bits_log2, hot_buckets);
  hipLaunchKernelGGL(k_popcount, dim3(256), dim3(block), 0, stream, bitmap, nwords, d_bits_set);
  CUDF_HIP_TRY(hipGetLastError());
}

}  // namespace cudf::groupby::detail
