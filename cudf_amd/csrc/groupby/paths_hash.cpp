// SPDX-License-Identifier: Apache-2.0
// aggregate_call, the open-addressing LDS hash tables: S - the groups fit one table, every workgroup aggregates a row chunk and
// the partial tables are merged; A - sorted / clustered rows are aggregated locally first; P - records are radix-partitioned on
// the top bits of a 64-bit key hash (optimistic one / two levels without a histogram pass, or exact offsets) so that each
// partition's groups fit one table. Replaces the reference's cuco::static_set + global atomics
// (cpp/src/groupby/hash/compute_groupby.cu:51-155, compute_global_memory_aggs.cuh:74-187, compute_shared_memory_aggs.cu:260-353).
#include "call.hpp"
#include "../common/wc_scatter.hpp"

#include <algorithm>
#include <cmath>

namespace cudf::groupby::detail {

// Heavy hitters stay in the (first-level) scatter workgroups: `pa` gets the key list and the per-workgroup partial
// buffers; merge_hot() folds those partials into one more work item behind the tables' items.
bool aggregate_call::setup_hot(part_args& pa, int64_t P)
{
    // (first the cheap tests: a scatter without write-combining has granule 0, and wc_scatter_lds_bytes divides by it - 1B rows with
    // two value columns on 1M groups, 24-byte records at 1024 partitions, died of SIGFPE here)
    if (hot_keys.empty() || pa.wc_granule == 0 || !p.simple || RU != 2) return false;
    auto const wc_lds = cudf::detail::wc_scatter_lds_bytes(5 * 1024, static_cast<std::size_t>(P), pa.wc_granule, 2);
    if (wc_lds + partition_hot_lds_bytes() + 1200 > 160 * 1024) return false;  // (the LDS table needs room next to the tile)
    size_t const wgs = static_cast<size_t>(pa.geom.slices);
    uint64_t* d_hot  = sc.alloc<uint64_t>(HOT_MAX_KEYS);
    CUDF_HIP_TRY(hipMemcpyAsync(d_hot, hot_keys.data(), hot_keys.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    pa.hot_n          = static_cast<int32_t>(hot_keys.size());
    pa.hot_keys       = d_hot;
    pa.hot_lds_offset = static_cast<int32_t>(wc_lds);
    pa.hot_out        = sc.alloc<uint64_t>(wgs * HOT_SLOTS * PU);
    pa.hot_count      = sc.alloc<int32_t>(wgs);
    return true;
  }

void aggregate_call::merge_hot(part_args const& pa, agg_args const& aa)
{  // partial / d_count hold room for item `nitems`  // partial / d_count hold room for item `nitems`
    agg_args hm    = aa;
    hm.input       = IN_PARTIAL_RECORDS;
    hm.seg         = SEG_STRIDED;
    hm.records     = pa.hot_out;
    hm.src_count   = pa.hot_count;
    hm.src_stride  = HOT_SLOTS;
    hm.fan         = pa.geom.slices;
    hm.nsrc        = pa.geom.slices;
    hm.out_records = partial + static_cast<size_t>(nitems) * ag.cap * PU;
    hm.out_count   = d_count + nitems;
    hm.nitems      = 1;
    launch_aggregate(hm, sc.alloc<agg_args>(1), s);
    nitems += 1;
  }

// Exact pipeline (histogram, scan, scatter with exact offsets, one or two levels, then one table per partition) over the rows
// `pa` describes: the plan's columns, or `nrec` records of `units` 8-byte units each (raw records, or - in_mode ==
// IN_PARTIAL_RECORDS - partial records of a local pre-aggregation; the partition kernels only hash the key units, `pplan` is
// the plan as they shall see it). Leaves partial / d_count / nitems for the finalize step.
void aggregate_call::exact_pipeline(part_args& pa, part_args* d_pa, plan_dev const& pplan, int64_t nrec, int units, agg_input in_mode, int64_t P1,
                                    int64_t P2, int log2P1, int log2P2, agg_args& aa)
{
    size_t const items1 = static_cast<size_t>(pa.geom.nseg) * static_cast<size_t>(pa.geom.slices);
    pa.counts       = sc.alloc<uint32_t>(items1 * P1);
    pa.item_base    = sc.alloc<int64_t>(items1 * P1);
    pa.out_offsets  = sc.alloc<int64_t>(P1 + 1);
    uint64_t* recA  = sc.alloc<uint64_t>(static_cast<size_t>(nrec) * units);
    pa.out_records  = recA;
    store_args(pa, d_pa, s);
    launch_partition_hist(pa, d_pa, s);
    launch_partition_scan(pa, d_pa, s);
    launch_partition_scatter(pa, d_pa, s);
    int64_t const* offsets = pa.out_offsets;
    uint64_t const* recs   = recA;
    int64_t nparts         = P1;
    if (P2 > 1) {
      part_args pb{};
      pb.plan         = pplan;
      pb.geom.nseg    = static_cast<int32_t>(P1);
      pb.geom.slices  = static_cast<int32_t>(std::max<int64_t>(1, 1024 / P1));
      pb.geom.P       = static_cast<int32_t>(P2);
      pb.geom.shift   = 64 - log2P1 - log2P2;
      pb.geom.block   = 1024;
      pb.from_columns = 0;
      pb.in_records   = recA;
      pb.seg_offsets  = pa.out_offsets;
      size_t const items2 = static_cast<size_t>(pb.geom.nseg) * pb.geom.slices;
      pb.counts       = sc.alloc<uint32_t>(items2 * P2);
      pb.item_base    = sc.alloc<int64_t>(items2 * P2);
      pb.out_offsets  = sc.alloc<int64_t>(P1 * P2 + 1);
      uint64_t* recB  = sc.alloc<uint64_t>(static_cast<size_t>(nrec) * units);
      pb.out_records  = recB;
      part_args* d_pb = sc.alloc<part_args>(1);
      store_args(pb, d_pb, s);
      launch_partition_hist(pb, d_pb, s);
      launch_partition_scan(pb, d_pb, s);
      launch_partition_scatter(pb, d_pb, s);
      offsets = pb.out_offsets;
      recs    = recB;
      nparts  = P1 * P2;
    }
    nitems         = static_cast<int32_t>(nparts);
    partial        = sc.alloc<uint64_t>(static_cast<size_t>(nitems) * ag.cap * PU);
    d_count        = sc.alloc<int32_t>(nitems);
    aa.input       = in_mode;
    aa.seg         = SEG_OFFSETS;
    aa.offsets     = offsets;
    aa.records     = recs;
    aa.out_records = partial;
    aa.out_count   = d_count;
    aa.nitems      = nitems;
    launch_aggregate(aa, sc.alloc<agg_args>(1), s);
  }

// ---------------- path S: every workgroup aggregates a row chunk in LDS, then partials are merged
outcome aggregate_call::run_single_pass(attempt_plan& ap)
{
  agg_args& aa = ap.aa;
      // ---------------- path S: every workgroup aggregates a row chunk in LDS, then partials are merged
      path          = hash_path::LDS_SINGLE_PASS;
      int64_t const items = std::clamp<int64_t>(n / 16384, 1, env.s_items);
      nitems              = static_cast<int32_t>(items);
      partial             = sc.alloc<uint64_t>(static_cast<size_t>(nitems) * ag.cap * PU);
      d_count             = sc.alloc<int32_t>(nitems);
      aa.input            = IN_COLUMNS;
      aa.seg              = SEG_ROW_CHUNKS;
      aa.nrows            = n;
      aa.chunk            = (n + items - 1) / items;
      aa.out_records      = partial;
      aa.out_count        = d_count;
      aa.nitems           = nitems;
      launch_aggregate(aa, sc.alloc<agg_args>(1), s);
      int const fan = 16;
      while (nitems > 1) {
        int32_t const next = (nitems + fan - 1) / fan;
        uint64_t* out      = sc.alloc<uint64_t>(static_cast<size_t>(next) * ag.cap * PU);
        int32_t* cnt       = sc.alloc<int32_t>(next);
        agg_args m         = aa;
        m.input            = IN_PARTIAL_RECORDS;
        m.seg              = SEG_STRIDED;
        m.records          = partial;
        m.src_count        = d_count;
        m.src_stride       = ag.cap;
        m.fan              = fan;
        m.nsrc             = nitems;
        m.out_records      = out;
        m.out_count        = cnt;
        m.nitems           = next;
        launch_aggregate(m, sc.alloc<agg_args>(1), s);
        partial = out;
        d_count = cnt;
        nitems  = next;
      }
  if (overflow_and_counts() == 0) return outcome::done;
  escalate();
  return outcome::retry_counted;
}

// ---------------- path A: sorted / clustered keys (most rows are followed by a row of the same key). Every partition
// scheme here gives a workgroup whole keys instead of a share of every key - regions overflow, rings stall, a wave's 64 rows
// meet in one table slot (1B sorted rows on 1M groups: 74 ms against 7 ms uniform). Row chunks small enough to hold few
// distinct keys are aggregated straight from the columns into one LDS table each (the single-pass kernel, wave-combined
// accumulate); their partial records - about one per run of equal keys - then take the exact partition pipeline and are
// merged. Reference: none (its global hash set does not care about row order).
outcome aggregate_call::try_preaggregate(attempt_plan& ap)
{
  agg_args& aa = ap.aa;
      if (adjacent_equal >= 0.01 * static_cast<double>(env.preagg_min_pct) && !pre_failed && env.forced_p == 0 && p.narg == 0) {
        double const run_starts = std::max(1.0 - adjacent_equal, 1e-6);  // distinct keys of a chunk <= its run starts
        // (and at least ~2048 chunks: long runs would otherwise leave most CUs without a chunk - 200M sorted rows on 100K groups
        // ran on 48 workgroups, 5.1 ms)
        // plain shapes: the run-collapsing front end (collapse_runs.hip: no table, one record per run and row set); the rest: row
        // chunks small enough for one LDS table each
        bool const collapse = env.collapse_runs && collapse_runs_applies(p);
        int64_t const chunk_rows =
          collapse ? std::clamp<int64_t>(n / 4096 / 1024 * 1024, int64_t{1} << 16, int64_t{1} << 19)
                   : std::clamp<int64_t>(static_cast<int64_t>(static_cast<double>(ag.fill_limit) / 1.5 / run_starts), int64_t{1} << 14,
                                         std::max<int64_t>(int64_t{1} << 14, n / 2048));
        int64_t const items = (n + chunk_rows - 1) / chunk_rows;
        // (records per chunk: a table's worth, or - collapsing - every run start plus one per row set, doubled)
        int64_t const stride1 = collapse ? static_cast<int64_t>(static_cast<double>(chunk_rows) * std::min(1.0, 2.0 * (run_starts + 1.0 / 64.0))) + 256 : ag.cap;
        if (static_cast<double>(items) * static_cast<double>(stride1) * PU * 8.0 <= 16.0 * 1024 * 1024 * 1024) {
          path         = hash_path::PARTITIONED_LDS;
          uint64_t* partial1 = sc.alloc<uint64_t>(static_cast<size_t>(items) * static_cast<size_t>(stride1) * PU);
          int32_t* d_count1  = sc.alloc<int32_t>(static_cast<size_t>(items));
          if (collapse) {
            collapse_args c1{};
            c1.plan        = p;
            c1.nrows       = n;
            c1.chunk       = chunk_rows;
            c1.nitems      = static_cast<int32_t>(items);
            c1.out_stride  = stride1;
            c1.out_records = partial1;
            c1.out_count   = d_count1;
            c1.overflow    = aa.overflow;
            launch_collapse_runs(c1, sc.alloc<collapse_args>(1), s);
          } else {
            agg_args a1    = aa;
            a1.input       = IN_COLUMNS;
            a1.seg         = SEG_ROW_CHUNKS;
            a1.nrows       = n;
            a1.chunk       = chunk_rows;
            a1.out_records = partial1;
            a1.out_count   = d_count1;
            a1.nitems      = static_cast<int32_t>(items);
            launch_aggregate(a1, sc.alloc<agg_args>(1), s);
          }
          nitems  = static_cast<int32_t>(items);
          d_count = d_count1;
          int32_t const ov1 = overflow_and_counts();
          int64_t n2 = 0;
          for (int32_t c : h_count) n2 += c;
          if (env.debug)
            fprintf(stderr, "[cudf_amd] pre-aggregation: %.1f %% of the rows repeat their predecessor's key, %ld chunks of %ld rows -> %ld partial records, overflow=%d\n",
                    100.0 * adjacent_equal, (long)items, (long)chunk_rows, (long)n2, ov1);
          if (ov1 != 0 || n2 * 2 > n) {  // a chunk held too many keys, or nothing was gained: the ordinary paths
            pre_failed = true;
            fresh_scratch();
            return outcome::retry_free;
          }
          // chunks' partial records -> one contiguous run (the partition kernels read segments of one buffer)
          int64_t* d_prefix1 = sc.alloc<int64_t>(static_cast<size_t>(items) + 1);
          launch_count_prefix(d_count1, static_cast<int32_t>(items), d_prefix1, s);
          uint64_t* compact = sc.alloc<uint64_t>(static_cast<size_t>(std::max<int64_t>(n2, 1)) * PU);
          launch_compact_records(partial1, stride1, d_prefix1, static_cast<int32_t>(items), PU, compact, s);
          int64_t* d_seg = sc.alloc<int64_t>(2);
          launch_store_i64x2(0, n2, d_seg, s);
          // tables for the merged groups
          double const need2 = std::min(est_groups * safety, static_cast<double>(n2)) / std::max(1.0, ag.cap * 0.25);
          auto pow2_up = [](double x) { int64_t v = 1; while (static_cast<double>(v) < x) v <<= 1; return v; };
          int64_t Q1 = std::clamp<int64_t>(pow2_up(need2), 16, 1024), Q2 = 1;
          if (need2 > 1024.0) {
            int64_t const tot = pow2_up(need2);
            Q1 = std::min<int64_t>(pow2_up(std::sqrt(static_cast<double>(tot))), 1024);
            Q2 = std::clamp<int64_t>(tot / Q1, 2, 1024);
          }
          int lq1 = 0, lq2 = 0;
          while ((int64_t{1} << lq1) < Q1) ++lq1;
          while ((int64_t{1} << lq2) < Q2) ++lq2;
          plan_dev p2 = p;  // what the partition kernels see: records of KU key units + NACC accumulator units
          p2.NPAY     = p.NACC;
          p2.simple   = 0;
          part_args pq{};
          pq.plan         = p2;
          pq.geom.nseg    = 1;
          pq.geom.slices  = static_cast<int32_t>(std::clamp<int64_t>(n2 / 16384, 16, 512));
          pq.geom.P       = static_cast<int32_t>(Q1);
          pq.geom.shift   = 64 - lq1;
          pq.geom.block   = 1024;
          pq.geom.tile_rows = 8 * 1024;
          pq.from_columns = 0;
          pq.in_records   = compact;
          pq.seg_offsets  = d_seg;
          exact_pipeline(pq, sc.alloc<part_args>(1), p2, n2, PU, IN_PARTIAL_RECORDS, Q1, Q2, lq1, lq2, aa);
          int32_t const ov2 = overflow_and_counts();
          if (ov2 == 0) return outcome::done;
          escalate();  // a merged table overflowed: more tables (the chunks are aggregated again)
          return outcome::retry_counted;
        }
      }
  return outcome::skip;
}

// granule (records) of the write-combining scatter for a fan-out, 0 = run-per-tile kernel: 64-byte granules for
// 16-byte records at P = 1024 (a 128-byte carry area would not leave room for a tile), else 128-256 bytes
int32_t aggregate_call::wc_granule_for(int64_t P) const
{
  if (!env.wc) return 0;
  int const G = RU == 2 ? static_cast<int>(env.wc_g >= 0 ? env.wc_g : (P > 512 ? 4 : 8)) : (RU == 4 ? 4 : 8);
  if (partition_wc_fits(RU, static_cast<int>(P), G)) return G;
  return (RU == 3 && partition_wc_fits(RU, static_cast<int>(P), 4)) ? 4 : 0;  // 24-byte records: 96-byte granules
}

// ---------------- path P: radix-partition raw records on hash bits, then one LDS table per partition
outcome aggregate_call::run_partitioned(attempt_plan& ap)
{
  agg_args& aa = ap.aa;
      path = hash_path::PARTITIONED_LDS;
      auto pow2_at_least = [](double x) {
        int64_t v = 1;
        while (static_cast<double>(v) < x) v <<= 1;
        return v;
      };
      int64_t const maxP1 = 1024;  // LDS: 8192-row stage + P * 12 B + pid must fit 160 KiB
      int64_t P1 = env.forced_p ? env.forced_p : std::clamp<int64_t>(pow2_at_least(ap.need), 256, maxP1);
      int64_t P2 = 1;
      if (!env.forced_p && ap.need > static_cast<double>(maxP1)) {
        int64_t const tot = pow2_at_least(ap.need);
        P1 = pow2_at_least(std::sqrt(static_cast<double>(tot)));
        P2 = tot / P1;
        P1 = std::min<int64_t>(P1, maxP1);
        P2 = std::clamp<int64_t>(P2, 2, maxP1);
      }
      int log2P1 = 0, log2P2 = 0;
      while ((int64_t{1} << log2P1) < P1) ++log2P1;
      while ((int64_t{1} << log2P2) < P2) ++log2P2;

      part_args pa{};
      pa.plan         = p;
      pa.geom.nseg    = 1;
      // optimistic: one persistent workgroup per CU (longer regions for the aggregate); exact: 2 per CU (the
      // histogram pass wants the parallelism: 1.6 ms at 512 slices vs 2.7 ms at 256)
      bool const will_try_optimistic = allow_optimistic && P2 == 1 && !env.exact && n >= (int64_t{1} << 22);
      // (small inputs: one slice per 16K rows - the single-workgroup scan walks slices x P counters)
      pa.geom.slices  = static_cast<int32_t>((env.slices >= 0 ? env.slices : (will_try_optimistic ? 256 : std::clamp<int64_t>(n / 16384, 16, 512))));
      pa.geom.P       = static_cast<int32_t>(P1);
      pa.geom.shift   = 64 - log2P1;
      pa.geom.block   = 1024;
      pa.geom.tile_rows = static_cast<int32_t>(env.rpt) * 1024;
      pa.from_columns = 1;
      pa.nrows        = n;
      size_t const items1 = static_cast<size_t>(pa.geom.slices);
      part_args* d_pa     = sc.alloc<part_args>(1);
      // single-level partitions of big inputs: try the optimistic single-pass partition first
      // Region sizing: rows of a (slice, partition) cell = sum over the ~G/P keys of the partition of their rows in
      // the slice; its relative spread has a key-count part 1/sqrt(G/P) (which keys hash there) and a row-sampling
      // part 1/sqrt(mean). Six sigmas of slack; if that needs more than 2x the memory, use the exact pipeline.
      double const cell_mean   = static_cast<double>(n) / static_cast<double>(items1) / static_cast<double>(P1);
      // (half the estimated key count: an over-estimate would under-size the regions)
      double const keys_per_p  = std::max(1.0, 0.5 * est_groups / static_cast<double>(P1));
      double const rel_sigma   = std::sqrt(1.0 / keys_per_p + 1.0 / std::max(1.0, cell_mean));
      bool const optimistic = allow_optimistic && P2 == 1 && !env.exact && n >= (int64_t{1} << 22) && 6.0 * rel_sigma <= 1.0;
      partition_plan const pp{P1, P2, log2P1, log2P2};
      if (optimistic) return try_optimistic_one_level(ap, pp, pa, d_pa);
      outcome const o2 = try_optimistic_two_level(ap, pp, pa, d_pa);
      if (o2 != outcome::skip) return o2;
      exact_pipeline(pa, d_pa, p, n, RU, IN_RAW_RECORDS, P1, P2, log2P1, log2P2, aa);
  if (overflow_and_counts() == 0) return outcome::done;
  escalate();
  return outcome::retry_counted;
}

// single-level partitions of big inputs: the optimistic single-pass partition (no histogram pass)
outcome aggregate_call::try_optimistic_one_level(attempt_plan& ap, partition_plan const& pp, part_args& pa, part_args* d_pa)
{
  agg_args& aa = ap.aa;
  int64_t const P1 = pp.P1;
  size_t const items1      = static_cast<size_t>(pa.geom.slices);
  double const cell_mean   = static_cast<double>(n) / static_cast<double>(items1) / static_cast<double>(P1);
  double const keys_per_p  = std::max(1.0, 0.5 * est_groups / static_cast<double>(P1));
  double const rel_sigma   = std::sqrt(1.0 / keys_per_p + 1.0 / std::max(1.0, cell_mean));
  uint64_t* recA           = nullptr;
        int64_t const capR  = (static_cast<int64_t>(cell_mean * (1.0 + 6.0 * rel_sigma) + 16.0) + 7) / 8 * 8;
        pa.optimistic       = 1;
        pa.region_cap       = capR;
        pa.region_count     = sc.alloc<int32_t>(items1 * P1);
        pa.overflow         = d_overflow;
        recA                = sc.alloc<uint64_t>(items1 * static_cast<size_t>(P1) * static_cast<size_t>(capR) * RU);
        pa.out_records      = recA;
        // 16-byte records: write-combining scatter (whole aligned granules only); 64-byte granules at P = 1024
        // (the carry area of 128-byte granules would not leave room for a tile), 128-byte granules at P <= 512
        pa.wc_granule = wc_granule_for(P1);
        pa.cyclic_tiles = pa.wc_granule != 0 && env.cyclic;
        bool const hot = setup_hot(pa, P1);
        if (env.stamps) pa.stamps = sc.alloc<unsigned long long>(items1 * 8);
        store_args(pa, d_pa, s);
        launch_partition_scatter(pa, d_pa, s);
        if (pa.stamps != nullptr) {
          std::vector<unsigned long long> h(items1 * 8);
          CUDF_HIP_TRY(hipMemcpyAsync(h.data(), pa.stamps, h.size() * 8, hipMemcpyDeviceToHost, s));
          CUDF_HIP_TRY(hipStreamSynchronize(s));
          double tot[8] = {0};
          for (size_t w = 0; w < items1; ++w) for (int i = 0; i < 8; ++i) tot[i] += static_cast<double>(h[w * 8 + i]);
          double all = 0; for (double t : tot) all += t;
          fprintf(stderr, "[cudf_amd] scatter phase shares (wave 0 of each WG, s_memtime): rank %.1f%% | barrier %.1f%% | scan %.1f%% | stage %.1f%% | prefetch-issue %.1f%% | barrier %.1f%% | write-out %.1f%% | barrier %.1f%%  (avg cycles/WG %.0f)\n",
                  100 * tot[0] / all, 100 * tot[1] / all, 100 * tot[2] / all, 100 * tot[3] / all, 100 * tot[4] / all, 100 * tot[5] / all, 100 * tot[6] / all, 100 * tot[7] / all, all / items1);
        }
        if (env.debug) { CUDF_HIP_TRY(hipStreamSynchronize(s)); fprintf(stderr, "[cudf_amd] optimistic scatter done capR=%ld P=%ld slices=%zu\n", (long)capR, (long)P1, items1); }
        nitems         = static_cast<int32_t>(P1);
        partial        = sc.alloc<uint64_t>(static_cast<size_t>(nitems + 1) * ag.cap * PU);
        d_count        = sc.alloc<int32_t>(nitems + 1);
        aa.input       = IN_RAW_RECORDS;
        aa.seg         = SEG_STRIDED;
        aa.records     = recA;
        aa.src_count   = pa.region_count;
        aa.src_stride  = capR;
        aa.fan         = pa.geom.slices;
        aa.nsrc        = static_cast<int32_t>(items1 * P1);
        aa.out_records = partial;
        aa.out_count   = d_count;
        aa.nitems      = nitems;
        launch_aggregate(aa, sc.alloc<agg_args>(1), s);
        if (hot) merge_hot(pa, aa);
        if (env.debug) { CUDF_HIP_TRY(hipStreamSynchronize(s)); fprintf(stderr, "[cudf_amd] optimistic aggregate done\n"); }

        int32_t const h_ov = overflow_and_counts();
        if (env.debug) fprintf(stderr, "[cudf_amd] optimistic overflow flag = %d\n", h_ov);
        if (h_ov == 0) return outcome::done;
        if ((h_ov & 1) == 0) {  // the regions held, a table overflowed: more tables, still without a histogram pass
          escalate();
          return outcome::retry_counted;
        }
        // a region overflowed (skewed keys): redo with exact offsets
        allow_optimistic = false;
        fresh_scratch();
        return outcome::retry_free;
}

outcome aggregate_call::try_optimistic_two_level(attempt_plan& ap, partition_plan const& pp, part_args& pa, part_args* d_pa)
{
  agg_args& aa = ap.aa;
  int64_t const P1 = pp.P1, P2 = pp.P2;
  int const log2P1 = pp.log2P1, log2P2 = pp.log2P2;
  uint64_t* recA   = nullptr;
      bool const warm_tail = hot_keys.size() >= static_cast<std::size_t>(HOT_MAX_KEYS);
      if (allow_optimistic && P2 > 1 && !env.exact && !warm_tail && n >= (int64_t{1} << 22) && env.optimistic2) {
        int64_t const S1      = 256;
        int64_t const slices2 = std::max<int64_t>(1, 512 / P1);
        double const mean1    = static_cast<double>(n) / static_cast<double>(S1 * P1);
        double const sigma1   = std::sqrt(1.0 / std::max(1.0, 0.5 * est_groups / static_cast<double>(P1)) + 1.0 / std::max(1.0, mean1));
        double const mean2    = static_cast<double>(n) / static_cast<double>(P1 * slices2 * P2);
        double const sigma2   = std::sqrt(1.0 / std::max(1.0, 0.5 * est_groups / static_cast<double>(P1 * P2)) + 1.0 / std::max(1.0, mean2));
        // (very many tiny tables: the per-table partial buffers dominate the memory; keep to one attempt there)
        bool const partials_fit = static_cast<double>(P1 * P2) * ag.cap * PU * 8.0 <= 32.0 * 1024 * 1024 * 1024;
        if (6.0 * sigma1 <= 1.0 && 6.0 * sigma2 <= 1.0 && S1 / slices2 <= 256 && partials_fit) {
          int64_t const cap1 = (static_cast<int64_t>(mean1 * (1.0 + 6.0 * sigma1) + 16.0) + 7) / 8 * 8;
          int64_t const cap2 = (static_cast<int64_t>(mean2 * (1.0 + 6.0 * sigma2) + 16.0) + 7) / 8 * 8;
          pa.geom.slices     = static_cast<int32_t>(S1);
          pa.optimistic      = 1;
          pa.region_cap      = cap1;
          pa.region_count    = sc.alloc<int32_t>(static_cast<size_t>(S1 * P1));
          pa.overflow        = d_overflow;
          recA               = sc.alloc<uint64_t>(static_cast<size_t>(S1 * P1) * static_cast<size_t>(cap1) * RU);
          pa.out_records     = recA;
          pa.wc_granule      = wc_granule_for(P1);
          pa.cyclic_tiles    = pa.wc_granule != 0 && env.cyclic;
          bool const hot     = setup_hot(pa, P1);  // (a key with percents of the rows would leave one table's workgroup alone with them)
          store_args(pa, d_pa, s);
          launch_partition_scatter(pa, d_pa, s);
          part_args pb{};
          pb.plan            = p;
          pb.geom.nseg       = static_cast<int32_t>(P1);
          pb.geom.slices     = static_cast<int32_t>(slices2);
          pb.geom.P          = static_cast<int32_t>(P2);
          pb.geom.shift      = 64 - log2P1 - log2P2;
          pb.geom.block      = 1024;
          pb.geom.tile_rows  = pa.geom.tile_rows;
          pb.from_columns    = 0;
          pb.in_records      = recA;
          pb.from_regions    = 1;
          pb.in_region_count = pa.region_count;
          pb.in_region_cap   = cap1;
          pb.in_slices       = static_cast<int32_t>(S1);
          pb.optimistic      = 1;
          pb.region_cap      = cap2;
          size_t const nreg2 = static_cast<size_t>(P1 * P2 * slices2);
          pb.region_count    = sc.alloc<int32_t>(nreg2);
          pb.overflow        = d_overflow;
          uint64_t* recB     = sc.alloc<uint64_t>(nreg2 * static_cast<size_t>(cap2) * RU);
          pb.out_records     = recB;
          pb.wc_granule      = wc_granule_for(P2);
          part_args* d_pb = sc.alloc<part_args>(1);
          store_args(pb, d_pb, s);
          launch_partition_scatter(pb, d_pb, s);
          nitems         = static_cast<int32_t>(P1 * P2);
          partial        = sc.alloc<uint64_t>(static_cast<size_t>(nitems + 1) * ag.cap * PU);
          d_count        = sc.alloc<int32_t>(nitems + 1);
          aa.input       = IN_RAW_RECORDS;
          aa.seg         = SEG_STRIDED;
          aa.records     = recB;
          aa.src_count   = pb.region_count;
          aa.src_stride  = cap2;
          aa.fan         = static_cast<int32_t>(slices2);
          aa.nsrc        = static_cast<int32_t>(nreg2);
          aa.out_records = partial;
          aa.out_count   = d_count;
          aa.nitems      = nitems;
          launch_aggregate(aa, sc.alloc<agg_args>(1), s);
          if (hot) merge_hot(pa, aa);
          int32_t const h_ov = overflow_and_counts();
          if (env.debug)
            fprintf(stderr, "[cudf_amd] two-level optimistic P1=%ld P2=%ld slices2=%ld cap1=%ld cap2=%ld RU=%d wc=%d/%d overflow=%d\n",
                    (long)P1, (long)P2, (long)slices2, (long)cap1, (long)cap2, RU, pa.wc_granule, pb.wc_granule, h_ov);
          if (h_ov == 0) return outcome::done;
          if ((h_ov & 1) == 0) {  // the regions held, a table overflowed
            escalate();
            return outcome::retry_counted;
          }
          allow_optimistic = false;  // a region overflowed: redo with exact offsets
          fresh_scratch();
          return outcome::retry_free;
        }
      }
  return outcome::skip;
}

}  // namespace cudf::groupby::detail
