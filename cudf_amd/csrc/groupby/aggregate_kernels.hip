// SPDX-License-Identifier: Apache-2.0
// gfx950 kernels of the hash-groupby engine around the LDS hash-table aggregate (aggregate_kernel.inl): finalize,
// cardinality estimate, HyperLogLog recount, heavy-hitter search; and the aggregate's dispatcher.
#include "device_common.hpp"

namespace cudf::groupby::detail {
namespace {
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
        } else if (d.kind == OUT_MEAN_INT) {
          // MEAN of a duration / decimal column: the SUM in the source type (wrapped to its width, as the reference's
          // accumulator is) divided by COUNT_VALID with C++ integer division (reference hash_compound_agg_finalizer.cu:92-133:
          // binary DIV with the source type as output; KATs mean_tests.cpp:152-198)
          int const sh      = 64 - 8 * d.width;
          int64_t const sum = static_cast<int64_t>(a0 << sh) >> sh;
          int64_t const n   = static_cast<int64_t>(rec[p.KU + d.a1]);
          bits              = static_cast<uint64_t>(valid && n > 0 ? sum / n : 0);
        } else if (d.kind == OUT_SUMOV_SUM || d.kind == OUT_SUMOV_FLAG) {
          // SUM_OVERFLOW (reference device_aggregators.cuh:136-160): sum in the source type, overflow = the exact sum does not
          // fit it. The reference's flag depends on the arrival order of its atomics; "every order overflows" is the part of it
          // that does not, and it is what its tests pin (sum_overflow_tests.cpp:260-378).
          int const w = d.key_unit;  // source width in bytes (field reused)
          __int128 exact;
          if (d.a2 >= 0) exact = (static_cast<__int128>(static_cast<int64_t>(a0)) << 32) + static_cast<__int128>(static_cast<int64_t>(rec[p.KU + d.a2]));
          else exact = static_cast<__int128>(static_cast<int64_t>(a0));
          int const sh        = 64 - 8 * w;
          int64_t const wrap  = static_cast<int64_t>(static_cast<uint64_t>(exact) << sh) >> sh;
          bool const overflow = exact != static_cast<__int128>(wrap);
          bits                = d.kind == OUT_SUMOV_SUM ? static_cast<uint64_t>(wrap) : static_cast<uint64_t>(overflow);
          if (!valid) bits = 0;
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
// RANGE: 1 / 2 = the key is one plain 8-byte integer column (signed / unsigned): the pass also takes the minimum and maximum of
// the sampled keys - per workgroup into blk_range[2 b], [2 b + 1] (k_popcount folds them: same-address global atomics
// serialise at ~12 ns each); a pass of its own over the same sample took as long as this one.
template <int RANGE>
__global__ void __launch_bounds__(256) k_estimate(plan_dev const* __restrict__ pp, int64_t nrows, int64_t sample,
                                                  uint32_t* bitmap, int32_t bits_log2, uint32_t* hot_buckets, uint64_t* blk_range,
                                                  uint32_t* blk_adj)
{
  plan_dev const& p = *pp;
  int64_t const i   = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  uint64_t key[MAX_KU];
  bool live = i < sample;
  bool pair = false, pair_equal = false;
  if (live) {
    uint32_t vv;
    int64_t const row = sample_row(i, nrows, sample);
    live = build_key_units<MAX_KU, false>(p, row, key, vv);
    // clustering: does the NEXT row carry the same key? (sorted / clustered inputs: the planner aggregates row chunks locally
    // first - blk_adj[2 b], [2 b + 1] = pairs looked at, pairs with equal keys, per workgroup; k_popcount folds them)
    if (blk_adj != nullptr && live && row + 1 < nrows) {
      uint64_t key2[MAX_KU];
      if (build_key_units<MAX_KU, false>(p, row + 1, key2, vv)) {
        pair       = true;
        pair_equal = true;
#pragma unroll
        for (int u = 0; u < MAX_KU; ++u)
          if (u < p.KU) pair_equal = pair_equal && key[u] == key2[u];
      }
    }
  }
  if (blk_adj != nullptr) {
    __shared__ uint32_t s_pairs[4], s_equal[4];
    uint32_t const np = static_cast<uint32_t>(__popcll(__ballot(pair))), ne = static_cast<uint32_t>(__popcll(__ballot(pair && pair_equal)));
    if ((threadIdx.x & 63) == 0) {
      s_pairs[threadIdx.x >> 6] = np;
      s_equal[threadIdx.x >> 6] = ne;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      blk_adj[2 * blockIdx.x]     = s_pairs[0] + s_pairs[1] + s_pairs[2] + s_pairs[3];
      blk_adj[2 * blockIdx.x + 1] = s_equal[0] + s_equal[1] + s_equal[2] + s_equal[3];
    }
  }
  if constexpr (RANGE != 0) {
    using T = std::conditional_t<RANGE == 1, long long, unsigned long long>;
    __shared__ T s_lo[4], s_hi[4];
    T lo = live ? static_cast<T>(key[0]) : std::numeric_limits<T>::max(), hi = live ? static_cast<T>(key[0]) : std::numeric_limits<T>::lowest();
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      T const l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
    }
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
      blk_range[2 * blockIdx.x]     = static_cast<uint64_t>(lo);
      blk_range[2 * blockIdx.x + 1] = static_cast<uint64_t>(hi);
    }
  }
  uint64_t const h   = live ? hash_key_units<MAX_KU>(p, key) : 0;
  uint32_t const bit = static_cast<uint32_t>(h >> (64 - bits_log2));
  // (same-address global atomics serialise - a constant key column made this pass take 16 ms: set a bit only if it is
  // not seen set, and add to a bucket counter once per distinct bucket of the wave)
  if (live && !((gload(bitmap + (bit >> 5)) >> (bit & 31)) & 1u)) atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
  // heavy-hitter search: rows per hash bucket over every fourth sampled row (scattered global atomics run at 24 G/s:
  // a quarter of the sample keeps this at ~10 us)
  bool pending = live && hot_buckets != nullptr && (i & 3) == 0;
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
// The threshold a key's sample count must reach: min_count, and at least 8x the mean count of a key of the sample (bits_set =
// distinct keys of the sample, left by k_popcount): in a column of few keys EVERY key is frequent (5000 groups: all of them
// passed min_count, the collect pass pushed the whole sample through its 4096-entry table - 0.34 ms - and the planner got 256
// ordinary keys as heavy hitters).
__host__ __device__ inline uint32_t hot_threshold(uint32_t min_count, int64_t counted_rows, uint32_t bits_set)
{
  uint64_t const mean8 = 8ull * static_cast<uint64_t>(counted_rows) / (bits_set > 0 ? bits_set : 1u);
  return static_cast<uint32_t>(mean8 > min_count ? (mean8 > 0xffffffffull ? 0xffffffffull : mean8) : min_count);
}
// (also: the sum of the squared bucket counts -> *sumsq. (sumsq - S) / S^2 - 1 / HOT_BUCKETS estimates the sum over the keys of
// their squared row shares - the chance that two rows carry the same key - from which the planner sizes regions for skewed keys)
__global__ void __launch_bounds__(256) k_hot_any(uint32_t const* buckets, uint32_t min_count, int64_t counted_rows, uint32_t const* bits_set,
                                                 uint32_t* crowded, unsigned long long* sumsq)
{
  int const i      = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t const c = i < HOT_BUCKETS ? buckets[i] : 0u;
  if (c >= hot_threshold(min_count, counted_rows, *bits_set) && i < HOT_BUCKETS) *crowded = 1;
  unsigned long long sq = static_cast<unsigned long long>(c) * c;
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o);
  if ((threadIdx.x & 63) == 0 && sq) atomicAdd(sumsq, sq);
}
__global__ void __launch_bounds__(256) k_hot_collect(plan_dev const* __restrict__ pp, int64_t nrows, int64_t sample,
                                                     uint32_t const* buckets, uint32_t const* crowded, uint32_t min_count_in,
                                                     uint32_t const* bits_set, uint64_t* tkeys, uint32_t* tcounts)
{
  if (*crowded == 0) return;  // no bucket reached the threshold: nothing to collect
  uint32_t const min_count = hot_threshold(min_count_in, (sample + 3) / 4, *bits_set);
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

// exclusive prefix of the per-item group counts (one workgroup: thread t sums a contiguous run of items, the runs are scanned
// through LDS): prefix[i] = counts[0] + ... + counts[i - 1], prefix[nitems] = total. Replaces a host-side scan + upload.
__global__ void __launch_bounds__(1024) k_count_prefix(int32_t const* __restrict__ counts, int32_t nitems, int64_t* __restrict__ prefix)
{
  __shared__ int64_t s_run[1024];
  int const per = (nitems + 1023) / 1024, b = threadIdx.x * per, e = min(b + per, nitems);
  int64_t sum = 0;
  for (int i = b; i < e; ++i) sum += counts[i];
  s_run[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
    int64_t const add = static_cast<int>(threadIdx.x) >= o ? s_run[threadIdx.x - o] : 0;
    __syncthreads();
    s_run[threadIdx.x] += add;
    __syncthreads();
  }
  int64_t run = s_run[threadIdx.x] - sum;
  for (int i = b; i < e; ++i) {
    prefix[i] = run;
    run += counts[i];
  }
  if (threadIdx.x == 1023) prefix[nitems] = s_run[1023];
}

// item i's records [i * cap, i * cap + (prefix[i + 1] - prefix[i])) of `units` words each -> out[prefix[i] ...]: one workgroup per item
__global__ void __launch_bounds__(256) k_compact_records(uint64_t const* __restrict__ records, int64_t cap, int64_t const* __restrict__ prefix,
                                                         int units, uint64_t* __restrict__ out)
{
  int64_t const b = prefix[blockIdx.x], words = (prefix[blockIdx.x + 1] - b) * units;
  uint64_t const* src = records + static_cast<int64_t>(blockIdx.x) * cap * units;
  uint64_t* dst       = out + b * units;
  for (int64_t i = threadIdx.x; i < words; i += blockDim.x) gstore(dst + i, gload(src + i));
}
struct i64x2_pod { int64_t a, b; };

// range_mode 1 / 2: workgroup 0 also folds the per-workgroup key ranges of k_estimate<RANGE> into range_out[0] = min, [1] = max
// blk_adj != nullptr: workgroup 1 folds the per-workgroup adjacent-pair counts into adj_out[0] = pairs, [1] = equal pairs
__global__ void __launch_bounds__(256) k_popcount(uint32_t const* bitmap, int64_t nwords, uint32_t* out, uint64_t const* blk_range,
                                                  int nblk, int range_mode, uint64_t* range_out, uint32_t const* blk_adj, uint32_t* adj_out)
{
  int64_t i     = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  uint32_t acc  = 0;
  for (; i < nwords; i += static_cast<int64_t>(gridDim.x) * blockDim.x) acc += __popc(bitmap[i]);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
  if (blk_adj != nullptr && blockIdx.x == 1) {
    __shared__ uint32_t s_p[4], s_e[4];
    uint32_t np = 0, ne = 0;
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
      np += blk_adj[2 * b];
      ne += blk_adj[2 * b + 1];
    }
    for (int o = 32; o > 0; o >>= 1) {
      np += __shfl_down(np, o);
      ne += __shfl_down(ne, o);
    }
    if ((threadIdx.x & 63) == 0) {
      s_p[threadIdx.x >> 6] = np;
      s_e[threadIdx.x >> 6] = ne;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      adj_out[0] = s_p[0] + s_p[1] + s_p[2] + s_p[3];
      adj_out[1] = s_e[0] + s_e[1] + s_e[2] + s_e[3];
    }
    return;
  }
  if (range_mode == 0 || blockIdx.x != 0) return;
  __shared__ uint64_t s_lo[4], s_hi[4];
  bool const sg = range_mode == 1;
  auto less = [sg](uint64_t a, uint64_t b) { return sg ? static_cast<long long>(a) < static_cast<long long>(b) : a < b; };
  uint64_t lo = sg ? static_cast<uint64_t>(INT64_MAX) : UINT64_MAX, hi = sg ? static_cast<uint64_t>(INT64_MIN) : uint64_t{0};
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
    uint64_t const l = blk_range[2 * b], h = blk_range[2 * b + 1];
    lo = less(l, lo) ? l : lo;
    hi = less(hi, h) ? h : hi;
  }
  for (int o = 32; o >= 1; o >>= 1) {
    uint64_t const l2 = __shfl_xor(static_cast<unsigned long long>(lo), o), h2 = __shfl_xor(static_cast<unsigned long long>(hi), o);
    lo = less(l2, lo) ? l2 : lo;
    hi = less(hi, h2) ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) {
    s_lo[threadIdx.x >> 6] = lo;
    s_hi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      lo = less(s_lo[w], lo) ? s_lo[w] : lo;
      hi = less(hi, s_hi[w]) ? s_hi[w] : hi;
    }
    range_out[0] = lo;
    range_out[1] = hi;
  }
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

void launch_aggregate_raw(agg_args const& a, agg_args* d_args, hipStream_t stream);
void launch_aggregate_partial(agg_args const& a, agg_args* d_args, hipStream_t stream);
void launch_aggregate_columns(agg_args const& a, agg_args* d_args, hipStream_t stream);
void launch_aggregate(agg_args const& a, agg_args* d_args, hipStream_t stream)
{
  if (a.nitems == 0) return;
  if (a.input == IN_RAW_RECORDS) return launch_aggregate_raw(a, d_args, stream);
  if (a.input == IN_PARTIAL_RECORDS) return launch_aggregate_partial(a, d_args, stream);
  launch_aggregate_columns(a, d_args, stream);
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

void launch_compact_records(uint64_t const* records, int64_t cap, int64_t const* prefix, int32_t nitems, int units, uint64_t* out, hipStream_t stream)
{
  if (nitems == 0) return;
  hipLaunchKernelGGL(k_compact_records, dim3(nitems), dim3(256), 0, stream, records, cap, prefix, units, out);
  CUDF_HIP_TRY(hipGetLastError());
}
void launch_store_i64x2(int64_t a, int64_t b, int64_t* dst, hipStream_t stream)
{
  hipLaunchKernelGGL(k_store_args<i64x2_pod>, dim3(1), dim3(1), 0, stream, i64x2_pod{a, b}, reinterpret_cast<i64x2_pod*>(dst));
  CUDF_HIP_TRY(hipGetLastError());
}

void launch_count_prefix(int32_t const* counts, int32_t nitems, int64_t* prefix, hipStream_t stream)
{
  hipLaunchKernelGGL(k_count_prefix, dim3(1), dim3(1024), 0, stream, counts, nitems, prefix);
  CUDF_HIP_TRY(hipGetLastError());
}

uint32_t hot_keys_threshold(uint32_t min_count, int64_t sample, uint32_t bits_set) { return hot_threshold(min_count, (sample + 3) / 4, bits_set); }

void launch_hot_keys(plan_dev const* d_plan, int64_t nrows, int64_t sample, uint32_t min_count, uint32_t* buckets, uint32_t const* d_bits_set,
                     uint64_t* table_keys, uint32_t* table_counts, hipStream_t stream)
{
  // table_counts holds HOT_TABLE counters, the one-word "some bucket is crowded" flag, a pad word and the 8-byte sum of the
  // squared bucket counts (HOT_TABLE + 4 words)
  CUDF_HIP_TRY(hipMemsetAsync(table_counts, 0, (HOT_TABLE + 4) * sizeof(uint32_t), stream));
  CUDF_HIP_TRY(hipMemsetAsync(table_keys, 0xff, HOT_TABLE * sizeof(uint64_t), stream));
  uint32_t* crowded   = table_counts + HOT_TABLE;
  unsigned const grid = static_cast<unsigned>(((sample + 3) / 4 + 255) / 256);
  cudf::detail::prof::scope prof_{"estimate", stream};
  hipLaunchKernelGGL(k_hot_any, dim3(HOT_BUCKETS / 256), dim3(256), 0, stream, buckets, min_count, (sample + 3) / 4, d_bits_set, crowded,
                     reinterpret_cast<unsigned long long*>(table_counts + HOT_TABLE + 2));
  // a bucket holds sample / 4 / 65536 counted keys on average (4 of a 1M-row sample): only crowded ones can hide a
  // heavy hitter, and their keys are counted exactly
  hipLaunchKernelGGL(k_hot_collect, dim3(grid), dim3(256), 0, stream, d_plan, nrows, sample, buckets, crowded, min_count, d_bits_set, table_keys,
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
                     int32_t bitmap_bits_log2, uint32_t* d_bits_set, uint32_t* hot_buckets, hipStream_t stream, int range_mode,
                     uint64_t* blk_range, uint64_t* range_out, uint32_t* blk_adj, uint32_t* adj_out)
{
  if (hot_buckets != nullptr) CUDF_HIP_TRY(hipMemsetAsync(hot_buckets, 0, HOT_BUCKETS * sizeof(uint32_t), stream));
  int64_t const nwords = (int64_t{1} << bitmap_bits_log2) / 32;
  CUDF_HIP_TRY(hipMemsetAsync(bitmap, 0, nwords * 4, stream));
  CUDF_HIP_TRY(hipMemsetAsync(d_bits_set, 0, 4, stream));
  int const block = 256;
  hipLaunchKernelGGL(k_store_args<plan_dev>, dim3(1), dim3(1), 0, stream, plan, d_plan);
  cudf::detail::prof::scope prof_{"estimate", stream};
  unsigned const grid = static_cast<unsigned>((sample + block - 1) / block);
  CUDF_EXPECTS(range_mode == 0 || (plan.simple && plan.KU == 1 && blk_range != nullptr && range_out != nullptr), "estimate: key range of one plain key column");
  CUDF_EXPECTS((blk_adj == nullptr) == (adj_out == nullptr), "estimate: adjacent-pair counts need their scratch and their output");
  if (range_mode == 1) hipLaunchKernelGGL(k_estimate<1>, dim3(grid), dim3(block), 0, stream, d_plan, nrows, sample, bitmap, bitmap_bits_log2, hot_buckets, blk_range, blk_adj);
  else if (range_mode == 2) hipLaunchKernelGGL(k_estimate<2>, dim3(grid), dim3(block), 0, stream, d_plan, nrows, sample, bitmap, bitmap_bits_log2, hot_buckets, blk_range, blk_adj);
  else hipLaunchKernelGGL(k_estimate<0>, dim3(grid), dim3(block), 0, stream, d_plan, nrows, sample, bitmap, bitmap_bits_log2, hot_buckets, blk_range, blk_adj);
  hipLaunchKernelGGL(k_popcount, dim3(256), dim3(block), 0, stream, bitmap, nwords, d_bits_set, blk_range, static_cast<int>(grid), range_mode, range_out,
                     blk_adj, adj_out);
  CUDF_HIP_TRY(hipGetLastError());
}

}  // namespace cudf::groupby::detail
