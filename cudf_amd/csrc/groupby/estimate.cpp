// SPDX-License-Identifier: Apache-2.0
// aggregate_call: geometry, the sample pass that plans a call (group estimate, key ranges, heavy hitters, share of rows that
// repeat their predecessor's key), the attempt loop and what happens when an attempt's tables overflow.
// Reference counterpart of the strategy choice: cpp/src/groupby/hash/compute_single_pass_aggs.cuh:32-164 (try the shared-memory
// path, fall back to the global one); here the choice is made from a sample before any row is moved.
#include "call.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace cudf::groupby::detail {

aggregate_call::aggregate_call(table_view const& keys, null_policy policy, std::span<aggregation_request const> requests, hipStream_t stream)
  : env{planner_env::load()},
    trace{env.trace},
    hp{build_plan(keys, policy, requests, env)},
    p{hp.dev},
    n{keys.num_rows()},
    s{stream},
    RU{hp.dev.KU + hp.dev.NPAY},
    PU{hp.dev.KU + hp.dev.NACC},
    sc{stream, cudf::get_current_device_resource_ref(), {}}
{
  trace.mark("plan");
  // ---- geometry of the LDS hash tables
  int64_t const lds_budget = env.lds_kb * 1024;
  int const slot_bytes     = aggregate_slot_bytes(p);  // key units + accumulators (COUNTs take 4 bytes) + state word
  // a multiple of 4: the table is probed in aligned buckets of four slots (one ds_read_b128 of state words)
  ag.cap        = static_cast<int32_t>(std::min<int64_t>(lds_budget / slot_bytes, 16384)) & ~3;
  ag.block      = static_cast<int32_t>(env.agg_block);
  ag.fill_limit = static_cast<int32_t>(ag.cap * 0.6);
  CUDF_EXPECTS(ag.cap >= 64, "Aggregation state per group too large for an LDS table.");
  d_overflow = sc.alloc<int32_t>(1);
  // Dense-key candidate: one plain 8-byte integer key column, one plain 8-byte value column (or two / three of them: one value
  // stream per column, paths_dense.cpp try_dense_ring_multi), no ARGMIN / ARGMAX
  dense_signed    = p.cols[0].cls == cudf::detail::CLS_SINT;
  // (the dense-key, heavy-hitter and pre-aggregation paths are for big inputs; CUDF_AMD_GB_BIG_MIN_ROWS lets the fuzz tests walk
  // them at sizes a CPU checker can follow)
  dense_candidate = p.simple && p.KU == 1 && p.narg == 0 && n >= env.one_table_min_rows && env.dense &&
                    (p.NPAY == 1 || (p.NPAY >= 2 && p.NPAY <= RING_MAX_VALUES && env.dense_multi));
  // Composite dense keys: 1-4 integer key columns of any width (rows with a NULL key are dropped: no nullable key under
  // null_policy::INCLUDE), exactly one value column, no ARGMIN / ARGMAX
  dense_composite = !dense_candidate && p.nkeycols <= DENSE_MAX_KEYS && hp.value_cols.size() == 1 && p.narg == 0 &&
                    hp.keynulls_unit < 0 && n >= env.one_table_min_rows && env.dense && env.dense_composite;
  for (int c = 0; c < p.nkeycols && dense_composite; ++c)
    dense_composite = p.cols[c].cls == cudf::detail::CLS_SINT || p.cols[c].cls == cudf::detail::CLS_UINT;
  allow_dense      = dense_candidate || dense_composite;
  allow_optimistic = env.optimistic;
}

// ---- the sample pass
void aggregate_call::estimate()
{
  // ---- distinct-count estimate on a strided sample (skipped when n already fits one table)
  est_groups = static_cast<double>(n);
  // Small inputs skip the estimate (three memsets, three kernels and a stream synchronisation: about a third of a 10K-row call):
  // they are planned for one table per 16K-row chunk, and a table that overflows sends the call through escalate() - which
  // counts the keys over all rows - like any other misjudged cardinality.
  bool const skip_estimate = n < env.estimate_min_rows;
  if (skip_estimate && n > ag.fill_limit) est_groups = static_cast<double>(ag.fill_limit) / 1.3 - 1.0;
  if (n > ag.fill_limit && !skip_estimate) {
    // 1M sampled rows for big inputs; small inputs sample 1/16 of their rows (at least 64K): the estimate only picks the
    // strategy, and a 1M-row sample costs more than the aggregation of a 1M-row input
    // (round 4 tried n / 64: the 10M-row call 363 -> 340 us, but with 94K sampled rows of 6M a column with 30 % of its rows on 12 keys shows
    // too few of its cold keys, looks sparse and leaves the dense path - CUDF_AMD_GB_SAMPLE_DIV stays 16)
    int64_t const sample = std::min<int64_t>(n, std::clamp<int64_t>(n / std::max<int64_t>(env.sample_div, 1), int64_t{1} << 16, int64_t{1} << 20));
    // (linear counting wants the bitmap at a sixteenth of its load: 2^24 bits for the 1M-row sample of a big input, 2^20 for the
    // 64K-row sample of a 1M-row input - whose memset and population count were a 2 MB pass each for nothing)
    int bits_log2 = 16;
    while (bits_log2 < 24 && (int64_t{1} << bits_log2) < 16 * sample) ++bits_log2;
    uint32_t* bitmap     = sc.alloc<uint32_t>((size_t{1} << bits_log2) / 32);
    uint32_t* d_set      = sc.alloc<uint32_t>(1);
    plan_dev* d_plan = sc.alloc<plan_dev>(1);
    bool const hot_eligible = p.simple && RU == 2 && p.KU == 1 && hot_plan_ok(p) && n >= env.big_min_rows &&
                              env.hot;
    uint32_t* hot_buckets = hot_eligible ? sc.alloc<uint32_t>(HOT_BUCKETS) : nullptr;
    // (one plain integer key column: the same pass takes the minimum and maximum of the sampled keys for the dense-key test)
    uint64_t* d_range = dense_candidate ? sc.alloc<uint64_t>(2) : nullptr;
    uint64_t* d_blk_range = dense_candidate ? sc.alloc<uint64_t>(2 * static_cast<std::size_t>((sample + 255) / 256)) : nullptr;
    // (and how often a row's successor carries the same key: sorted / clustered inputs are aggregated in row chunks first)
    bool const want_adj = n >= env.big_min_rows && p.narg == 0 && env.preagg;
    uint32_t* d_blk_adj = want_adj ? sc.alloc<uint32_t>(2 * static_cast<std::size_t>((sample + 255) / 256)) : nullptr;
    uint32_t* d_adj     = want_adj ? sc.alloc<uint32_t>(2) : nullptr;
    launch_estimate(p, d_plan, n, sample, bitmap, bits_log2, d_set, hot_buckets, s, dense_candidate ? (dense_signed ? 1 : 2) : 0, d_blk_range, d_range,
                    d_blk_adj, d_adj);
    // Dense integer keys ("Dense keys"): minimum and maximum of the key column over the same sample
    // (every read-back of this pass lands in page-locked memory: a copy into pageable memory blocks the host until it is done)
    unsigned char* const pin = pinned_bytes(64 + 16 + sizeof(h_ranges) + HOT_TABLE * (sizeof(uint64_t) + sizeof(uint32_t)) + 16);
    uint32_t* const pin_set    = reinterpret_cast<uint32_t*>(pin);
    uint64_t* const pin_range  = reinterpret_cast<uint64_t*>(pin + 64);
    int64_t* const pin_ranges  = reinterpret_cast<int64_t*>(pin + 64 + 16);
    uint64_t* const pin_tkeys  = reinterpret_cast<uint64_t*>(pin + 64 + 16 + sizeof(h_ranges));
    uint32_t* const pin_tcounts = reinterpret_cast<uint32_t*>(pin_tkeys + HOT_TABLE);
    if (dense_candidate) {
      CUDF_HIP_TRY(hipMemcpyAsync(pin_range, d_range, 16, hipMemcpyDeviceToHost, s));
    } else if (dense_composite) {
      int64_t* d_ranges = sc.alloc<int64_t>(2 * MAX_KU);
      launch_key_ranges(d_plan, p.nkeycols, n, sample, d_ranges, s);
      CUDF_HIP_TRY(hipMemcpyAsync(pin_ranges, d_ranges, sizeof(h_ranges), hipMemcpyDeviceToHost, s));
    }
    // Heavy hitters (plain int64 key + one plain value, SUM / COUNT): a key above ~0.05 % of the rows overflows its
    // regions of the optimistic partition, and a key with percents of the rows leaves one workgroup aggregating its
    // partition alone. Keys seen min_count times in the sample are aggregated inside the scatter workgroups instead.
    std::vector<uint64_t> h_tkeys;
    std::vector<uint32_t> h_tcounts;
    // (a key overflows its regions from about 0.24 / P of the rows: 0.023 % at P = 1024; the threshold is half of that)
    uint32_t const hot_min_count = static_cast<uint32_t>(std::max<int64_t>(16, sample / 4 / 8192));  // of every 4th sampled row
    if (hot_eligible) {
      uint32_t* buckets = hot_buckets;
      uint64_t* tkeys   = sc.alloc<uint64_t>(HOT_TABLE);
      uint32_t* tcounts = sc.alloc<uint32_t>(HOT_TABLE + 4);
      launch_hot_keys(d_plan, n, sample, hot_min_count, buckets, d_set, tkeys, tcounts, s);
      CUDF_HIP_TRY(hipMemcpyAsync(pin_tkeys, tkeys, HOT_TABLE * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
      CUDF_HIP_TRY(hipMemcpyAsync(pin_tcounts, tcounts, (HOT_TABLE + 4) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    }
    CUDF_HIP_TRY(hipMemcpyAsync(pin_set, d_set, 4, hipMemcpyDeviceToHost, s));
    if (want_adj) CUDF_HIP_TRY(hipMemcpyAsync(pin_set + 2, d_adj, 8, hipMemcpyDeviceToHost, s));
    trace.mark("estimate queued");
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    trace.mark("estimate back");
    uint32_t const h_set = *pin_set;
    if (want_adj && pin_set[2] >= 1024) adjacent_equal = static_cast<double>(pin_set[3]) / static_cast<double>(pin_set[2]);
    if (dense_candidate) std::memcpy(h_range, pin_range, 16);
    if (dense_composite) std::memcpy(h_ranges, pin_ranges, sizeof(h_ranges));
    ranges_known = dense_candidate || dense_composite;
    if (hot_eligible) {
      h_tkeys.assign(pin_tkeys, pin_tkeys + HOT_TABLE);
      h_tcounts.assign(pin_tcounts, pin_tcounts + HOT_TABLE);
    }
    if (hot_eligible) {  // the most frequent keys first, at most HOT_MAX_KEYS of them
      std::vector<std::pair<uint32_t, uint64_t>> cand;
      for (int i = 0; i < HOT_TABLE; ++i)
        if (h_tkeys[i] != ~uint64_t{0} && h_tcounts[i] >= hot_keys_threshold(hot_min_count, sample, h_set)) cand.emplace_back(h_tcounts[i], h_tkeys[i]);
      std::sort(cand.begin(), cand.end(), [](auto const& a, auto const& b) { return a.first > b.first; });
      // Skew of the keys that stay in the scatter: the sum of their squared row shares (two rows meet in a key with that
      // probability) from the hashed counters, less the heavy hitters' own; a partition's share of the rows then varies by
      // sqrt(skew_m2 * partitions) of its mean on top of the sampling noise (region_cap_for in paths_dense.cpp)
      double const counted = static_cast<double>((sample + 3) / 4);
      unsigned long long sumsq = 0;
      std::memcpy(&sumsq, pin_tcounts + HOT_TABLE + 2, sizeof(sumsq));
      double m2 = std::max(0.0, (static_cast<double>(sumsq) - counted) / (counted * counted) - 1.0 / HOT_BUCKETS);
      if (cand.size() > HOT_MAX_KEYS) cand.resize(HOT_MAX_KEYS);
      hot_mass = 0.0;
      for (auto const& c : cand) {
        hot_keys.push_back(c.second);
        double const share = static_cast<double>(c.first) / counted;
        hot_mass += share;
        m2 -= share * share;
      }
      skew_m2 = std::max(0.0, m2);
      if (env.debug)
        fprintf(stderr, "[cudf_amd] heavy hitters: %zu keys (most frequent: %u of %ld counted rows), %.3f of the rows; squared shares of the other keys sum to %.3g\n",
                hot_keys.size(), cand.empty() ? 0u : cand[0].first, (long)(sample / 4), hot_mass, skew_m2);
    }
    double const m  = std::ldexp(1.0, bits_log2);
    double const ds = h_set >= m ? m * 20 : -m * std::log(1.0 - h_set / m);  // distinct keys in the sample
    // population estimate under uniform frequencies: solve G (1 - exp(-S/G)) = ds
    double const S = static_cast<double>(sample);
    // (few duplicates in the sample still carry information: distinct ~ S - S^2 / 2G; only a sample without any
    // duplicate leaves G unbounded)
    if (ds >= S - 0.5 || sample == n) {
      est_groups = sample == n ? ds : static_cast<double>(n);
    } else {
      double lo = ds, hi = static_cast<double>(n);
      for (int it = 0; it < 60; ++it) {
        double const g = 0.5 * (lo + hi);
        if (g * (1.0 - std::exp(-S / g)) < ds) lo = g; else hi = g;
      }
      est_groups = std::min<double>(hi, static_cast<double>(n));
    }
    est_groups = std::max(est_groups, 1.0);
  }
}

int32_t aggregate_call::overflow_and_counts()
{
    int32_t* const pin = pinned_ints(static_cast<std::size_t>(nitems) + 1);
    CUDF_HIP_TRY(hipMemcpyAsync(pin, d_overflow, 4, hipMemcpyDeviceToHost, s));
    CUDF_HIP_TRY(hipMemcpyAsync(pin + 1, d_count, sizeof(int32_t) * static_cast<std::size_t>(nitems), hipMemcpyDeviceToHost, s));
    trace.mark("attempt queued");
    CUDF_HIP_TRY(hipStreamSynchronize(s));
    trace.mark("attempt back");
    h_count.assign(pin + 1, pin + 1 + nitems);
    return pin[0];
  }

void aggregate_call::fresh_scratch()
{
  sc.bufs.clear();
  d_overflow = sc.alloc<int32_t>(1);
}

void aggregate_call::escalate()
{
    // A table overflowed: the estimate was too low (skewed sample). First time: count the distinct key rows over ALL rows (HyperLogLog, one
    // streaming pass over the key columns) and plan from that; after that, ask for 8x more tables and redo.
    sc.bufs.clear();
    if (!counted_all) {
      counted_all       = true;
      uint32_t* regs    = sc.alloc<uint32_t>(HLL_REGISTERS);
      plan_dev* d_plan2 = sc.alloc<plan_dev>(1);
      launch_distinct_count(p, d_plan2, n, regs, s);
      std::vector<uint32_t> h_regs(HLL_REGISTERS);
      CUDF_HIP_TRY(hipMemcpyAsync(h_regs.data(), regs, HLL_REGISTERS * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      CUDF_HIP_TRY(hipStreamSynchronize(s));
      double const counted = hyperloglog_estimate(h_regs);
      if (env.debug)
        fprintf(stderr, "[cudf_amd] table overflow: sample estimate %.0f groups, HyperLogLog over all rows %.0f\n", est_groups, counted);
      // (1.05: three standard errors of the 2^14-register estimate; never plan for fewer groups than the failed attempt)
      est_groups = std::min<double>(static_cast<double>(n), std::max(est_groups * 1.5, counted * 1.05));
      sc.bufs.clear();
    } else {
      // (an estimate already at the row count cannot grow: the planned load drops instead - tables of a few hundred slots,
      // CUDF_AMD_GB_LDS_KB=8 in the fuzz tests, overflowed at 4 sigma of an all-distinct key column and the same plan was redone)
      if (est_groups >= static_cast<double>(n)) safety *= 2.0;
      est_groups = std::min<double>(static_cast<double>(n), std::max(est_groups, static_cast<double>(ag.fill_limit)) * 8);
    }
    d_overflow = sc.alloc<int32_t>(1);
  }

// The dense map of the key columns from the sampled ranges (h_range / h_ranges): lo, range, the mixed-radix digits of composite
// keys. False if the keys are not dense-eligible or span too much.
bool aggregate_call::dense_map_from_sample(dense_map& dm, bool tight) const
{
    bool dense_ok = false;
    if (!ranges_known) return false;
    if (dense_candidate) {
      // one plain 8-byte key: range from the sample, widened by a margin (the sample's extremes of a dense column miss the
      // true ones by about range / sample); a key outside [lo, lo + range) voids the attempt (overflow bit 2): redone by hash
      uint64_t const width  = h_range[1] - h_range[0];  // exact in two's complement for either ordering
      // (tight: a handful of groups, every one of them sampled hundreds of times - the one-table path wants a small table)
      uint64_t const margin = tight ? std::max<uint64_t>(width / 32, 64) : std::clamp<uint64_t>(width / 64, 4096, uint64_t{1} << 26);
      uint64_t lo, hi;
      if (dense_signed) {
        int64_t const l = static_cast<int64_t>(h_range[0]), h = static_cast<int64_t>(h_range[1]);
        lo = static_cast<uint64_t>(l < INT64_MIN + static_cast<int64_t>(margin) ? INT64_MIN : l - static_cast<int64_t>(margin));
        hi = static_cast<uint64_t>(h > INT64_MAX - static_cast<int64_t>(margin) ? INT64_MAX : h + static_cast<int64_t>(margin));
      } else {
        lo = h_range[0] < margin ? 0 : h_range[0] - margin;
        hi = h_range[1] > UINT64_MAX - margin ? UINT64_MAX : h_range[1] + margin;
      }
      dm.lo    = lo;
      dm.range = hi - lo + 1;  // (0 if the keys span the whole type: fails the test below)
      dense_ok = width <= (uint64_t{1} << 30) && dm.range != 0;
    } else {
      // composite: every key column contributes the digit (value - lo_c) of a mixed-radix index, last column fastest
      double total = 1.0;
      dense_ok     = true;
      for (int c = 0; c < p.nkeycols; ++c) {
        int64_t const l = h_ranges[2 * c], h = h_ranges[2 * c + 1];
        if (l > h || static_cast<double>(h) - static_cast<double>(l) > 1e9) { dense_ok = false; break; }  // (no valid sampled value / wide)
        int64_t const width  = h - l;
        int64_t const margin = width / 64 + (width >= 64 ? 2 : 0);
        dense_key& dk = dm.key[c];
        dk.lo        = static_cast<uint64_t>(l - margin);
        dk.range     = static_cast<uint32_t>(width + 2 * margin + 1);
        dk.col       = static_cast<int8_t>(c);
        dk.unit      = static_cast<int8_t>(hp.key_unit[c]);
        dk.half      = static_cast<int8_t>(hp.key_half[c]);
        dk.is_signed = p.cols[c].cls == cudf::detail::CLS_SINT;
        dk.width     = p.cols[c].width;
        total *= static_cast<double>(dk.range);
      }
      dense_ok = dense_ok && total <= static_cast<double>(uint64_t{1} << 30);
      if (dense_ok) {
        uint64_t stride = 1;
        for (int c = p.nkeycols - 1; c >= 0; --c) {
          dm.key[c].stride = static_cast<uint32_t>(stride);
          stride *= dm.key[c].range;
        }
        dm.range          = stride;
        dm.nkeys          = p.nkeycols;
        dm.value_col      = p.nkeycols;
        dm.value_nullable = p.cols[p.nkeycols].mask != nullptr;
      }
    }
    return dense_ok;
  }

aggregate_call::attempt_plan aggregate_call::plan_attempt() const
{
  attempt_plan ap{};
  // tables needed; the group count can never exceed the row count
  // Planned table load. Bucketed probing resolves a row in two LDS round trips up to ~0.4; a lighter table means
  // more partitions. 16-byte records (write-combining scatter): halving the fan-out saves more in the scatter
  // (C2: 8.3 -> 7.1 ms) than the fuller tables cost the aggregate (3.1 -> 3.7 ms), so plan for 0.45/safety = 0.35.
  bool const wc_eligible = (RU == 2 || RU == 3) && p.KU == 1 && env.wc;
  double const plan_fill = std::max(1.0, ag.cap * 0.01 * static_cast<double>(env.plan_load_pct >= 0 ? env.plan_load_pct : (wc_eligible ? 45 : 25)));
  // (groups <= rows, and the safety factor rides on top of that bound: more than `safety` x n / plan_fill tables are never needed
  // for the tables' MEAN load, but without it an all-distinct key column was planned at the full load with no slack)
  ap.need = std::min(est_groups, static_cast<double>(n)) * (est_groups >= static_cast<double>(n) ? safety : std::min(safety, static_cast<double>(n) / est_groups)) / plan_fill;
  ap.aa.plan     = p;
  ap.aa.geom     = ag;
  ap.aa.overflow = d_overflow;
  ap.fits_one_table = std::min(est_groups * safety, static_cast<double>(n)) <= ag.fill_limit;
  return ap;
}

// One attempt: the paths in their order of preference; the first that applies takes it.
outcome aggregate_call::run_attempt(int attempt)
{
  CUDF_HIP_TRY(hipMemsetAsync(d_overflow, 0, 4, s));
  attempt_plan ap = plan_attempt();
  outcome o = try_dense_one_table();                       // T: the key range fits ONE direct-address table
  if (o != outcome::skip) return o;
  if (ap.fits_one_table && env.forced_p == 0) {
    o = run_single_pass(ap);                               // S: the groups fit one hash table
  } else {
    final_cap = 0;
    o = try_preaggregate(ap);                              // A: sorted / clustered rows
    if (o == outcome::skip) o = try_dense_ring();          // D: dense keys, ring scatter
    if (o == outcome::skip) o = try_dense_wc();            // D: dense keys, write-combining scatter
    if (o == outcome::skip) o = run_partitioned(ap);       // P: radix partition on hash bits
  }
  if (o == outcome::retry_counted && env.debug)
    fprintf(stderr, "[cudf_amd] attempt %d: a table overflowed with %d tables of %d slots (fill limit %d) for an estimate of %.0f groups in %ld rows\n",
            attempt, nitems, ag.cap, ag.fill_limit, est_groups, (long)n);
  return o;
}

void aggregate_call::run()
{
  // (Tried in round 3 and removed: for 64K <= n < 32M rows, path S first without the sample pass. 1M rows on 1000 groups 232 ->
  // 179 us, but a failed try is not cheap - 1M rows on 500K groups 214 -> 365 us, 10M rows on 1M groups 346 -> 724 us:
  // profiles/r3_small_call_latency.txt.)
  estimate();
  for (int attempt = 0;;) {
    CUDF_EXPECTS(attempt < 4, "hash groupby: could not fit the groups into LDS tables (pathological key distribution).");
    outcome const o = run_attempt(attempt);
    if (o == outcome::done) break;
    CUDF_EXPECTS(o != outcome::skip, "hash groupby: no path took the call");
    if (o == outcome::retry_counted) ++attempt;
  }
}

}  // namespace cudf::groupby::detail
