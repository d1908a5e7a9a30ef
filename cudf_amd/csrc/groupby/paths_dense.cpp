// SPDX-License-Identifier: Apache-2.0
// aggregate_call, dense integer keys: T - the key range fits ONE direct-address table, every workgroup
// aggregates row tiles straight from the columns; D - ring scatter (or the write-combining scatter) into partitions whose key
// ranges fit one direct-address LDS table each. Replaces, for such keys, the reference's cuco::static_set insert + global atomics
// (cpp/src/groupby/hash/compute_global_memory_aggs.cuh:74-187) and its shared-memory path (compute_shared_memory_aggs.cu:260-353).
#include "call.hpp"
#include "../common/wc_scatter.hpp"

#include <algorithm>
#include <cmath>

namespace cudf::groupby::detail {

// ---------------- path T: a key range small enough for ONE direct-address table (up to 8192 groups for SUM + COUNT, whatever
// the hash tables would hold): every workgroup aggregates its row tiles straight from the columns into a table of its own
// (no hash, no probe, no key compare, no partition pass) and k_dense_merge_dump_wide folds the images.
outcome aggregate_call::try_dense_one_table()
{
    if (env.forced_p == 0 && allow_dense && env.dense_one_table) {
      dense_map dm{};
      bool ok = dense_map_from_sample(dm, true);
      int bits = 6;
      while (bits < 30 && (uint64_t{1} << bits) < dm.range) ++bits;
      int const slots         = 1 << bits;
      std::size_t const image = ok ? dense_table_bytes(p, slots) : 0;
      ok = ok && dm.range <= static_cast<uint64_t>(slots) && image <= 96 * 1024;
      if (ok) {
        path   = hash_path::DENSE_DIRECT;
        dm.mult      = 1;
        dm.mult_inv  = 1;
        dm.bits      = bits;
        dm.log2P     = 0;
        int const DPU    = (dm.nkeys > 0 ? p.KU : 1) + p.NACC;
        int const nwg    = static_cast<int>(std::clamp<int64_t>(n / 16384, 1, image <= 48 * 1024 ? 512 : 256));
        int const dsplit = slots / 64;  // (items of the image fold: 64 slots each)
        dense_agg_args da{};
        da.plan        = p;
        da.map         = dm;
        da.nsplit      = nwg;
        da.slots       = slots;
        da.image_bytes = static_cast<int32_t>(image);
        da.occ_acc     = dense_occ_acc(p);
        da.KU          = dm.nkeys > 0 ? p.KU : 1;
        da.tables      = sc.alloc<uint64_t>(static_cast<size_t>(nwg) * image / 8);
        partial        = sc.alloc<uint64_t>(static_cast<size_t>(slots) * DPU);
        d_count        = sc.alloc<int32_t>(dsplit);
        da.out_records = partial;
        da.out_count   = d_count;
        da.overflow    = d_overflow;
        da.nitems      = 1;
        da.block       = 1024;
        da.nrows       = n;
        if (dm.nkeys > 0) {
          uint32_t* ones = sc.alloc<uint32_t>(16);
          CUDF_HIP_TRY(hipMemsetAsync(ones, 0xff, 64, s));
          da.ones = ones;
        }
        dense_agg_args* d_da = sc.alloc<dense_agg_args>(1);
        store_args(da, d_da, s);
        launch_aggregate_dense_columns(da, d_da, s);
        launch_dense_merge_dump_wide(da, d_da, s);
        nitems    = dsplit;
        final_cap = 64;
        int32_t const h_ov = overflow_and_counts();
        if (env.debug)
          fprintf(stderr, "[cudf_amd] dense keys (one table): nkeys=%d lo=%lld range=%llu slots=%d image=%zu B workgroups=%d overflow=%d\n", dm.nkeys,
                  (long long)dm.lo, (unsigned long long)dm.range, slots, image, nwg, h_ov);
        if (h_ov == 0) return outcome::done;
        // a key outside the sampled range: the hash tables
        allow_dense = false;
        final_cap   = 0;
        fresh_scratch();
        return outcome::retry_free;
      }
    }
  return outcome::skip;
}

// ---------------- path D: dense integer keys -> direct-address LDS tables (no hash, no probe, no key words)
// ---- ring scatter (dense_ring_kernels.hip): 10-byte records in two streams, one or two levels of fan-out 16 ... 256, the
// largest tables that fit (one 1024-thread aggregate workgroup per CU; a partition's regions are shared out to several
// workgroups when there are fewer partitions than CUs)
outcome aggregate_call::try_dense_ring()
{
  // (heavy hitters in the sample: only the single-level ring scatter of one plain key takes them out of the partition)
  bool const ring_env = env.dense_ring;
  // (below big_min_rows only the one-table path T is tried: the partition passes are for big inputs)
  if (!(allow_dense && n >= env.big_min_rows && env.forced_p == 0 && (hot_keys.empty() || (dense_candidate && ring_env)) && !env.exact)) return outcome::skip;
  if (dense_candidate && p.NPAY > 1) return ring_env ? try_dense_ring_multi() : outcome::skip;  // one value stream per column
      dense_map dm{};
      bool dense_ok = dense_map_from_sample(dm);
      int bits = 14;
      while (bits < 31 && (uint64_t{1} << bits) < dm.range) ++bits;
        if (dense_ok && ring_env) {
          int rlog2P = 7;
          while (rlog2P < 17 && dense_table_bytes(p, 1 << std::max(bits - rlog2P, 0)) > 150 * 1024) ++rlog2P;
          if (env.dense_log2p > 0) rlog2P = static_cast<int>(env.dense_log2p);
          bool const two_level = rlog2P > 8;
          int const l1 = two_level ? (rlog2P + 1) / 2 : rlog2P, l2 = rlog2P - l1;
          int const slots         = 1 << std::max(bits - rlog2P, 0);
          std::size_t const image = dense_table_bytes(p, slots);
          int32_t const ntables = static_cast<int32_t>(int64_t{1} << rlog2P);
          int const nsplit      = static_cast<int>(std::clamp<int64_t>((env.dense_nsplit >= 0 ? env.dense_nsplit : 256 / ntables), 1, two_level ? 1 : 16));
          // (heavy hitters: one level, and their merged item - at most HOT_MAX_KEYS groups - must fit the stride of the items)
          bool const ring_ok = (hot_keys.empty() || slots / nsplit >= HOT_MAX_KEYS) &&
                               dm.range <= (uint64_t{1} << bits) && bits <= 30 && bits - rlog2P >= 6 && rlog2P <= 16 && l1 >= 4 && l1 <= 8 &&
                               (l2 == 0 || (l2 >= 4 && l2 <= 8)) && image <= 150 * 1024 &&
                               // (a column with heavy hitters: the sample's distinct count says little about the tail - log-uniform
                               // keys over 10M values looked like 380K groups - but tables over a range of at most an eighth of the
                               // rows cost little whatever the number of groups turns out to be)
                               static_cast<double>(dm.range) <= std::max(8.0 * std::max(est_groups, 4096.0), hot_mass > 0.02 ? static_cast<double>(n) / 8.0 : 0.0);
          if (ring_ok) {
            path  = hash_path::DENSE_DIRECT;
            dm.mult     = 0x9E3779B1u;  // odd: index -> (index * mult) mod 2^bits is a bijection
            uint32_t inv = dm.mult;     // Newton: inv = mult^-1 mod 2^32
            for (int it = 0; it < 5; ++it) inv *= 2u - dm.mult * inv;
            dm.mult_inv = inv;
            dm.bits     = bits;
            dm.log2P    = rlog2P;
            int const DPU      = (dm.nkeys > 0 ? p.KU : 1) + p.NACC;  // units of a dumped partial record
            int64_t const PD = int64_t{1} << l1, P2D = int64_t{1} << l2, S = 256;
            auto region_cap_for = [&](double mean, double parts) {
              double const keys_per_p = std::max(1.0, 0.5 * est_groups / parts);
              // (skewed keys - Zipf over 10M keys leaves 3 % per partition after the 256 heaviest - widen the spread of a
              // partition's share: counted only from 10 % on, below that the equal-weights term covers it)
              double const skew       = skew_m2 * parts / std::max(0.05, (1.0 - hot_mass) * (1.0 - hot_mass));
              double const rel_sigma  = std::sqrt(1.0 / keys_per_p + 1.0 / std::max(1.0, mean) + (skew > 0.01 ? skew : 0.0));
              return (static_cast<int64_t>(mean * (1.0 + 6.0 * std::min(rel_sigma, 1.0)) + 64.0) + 63) / 64 * 64;
            };
            // (workgroup w takes the 4096-row tiles w, w + S, ...: the busiest workgroup has ceil(tiles / S) of them)
            int64_t const ring_tile = 4 * 1024, wg_rows = std::min<int64_t>(n, ((n + ring_tile - 1) / ring_tile + S - 1) / S * ring_tile);
            int64_t const capR = region_cap_for(static_cast<double>(wg_rows) / static_cast<double>(PD), static_cast<double>(PD));
            dense_ring_args ra{};
            ra.plan         = p;
            ra.map          = dm;
            ra.static_shapes = static_cast<int32_t>(env.static_shapes);
            ra.from_columns = 1;
            ra.nrows        = n;
            ra.P            = static_cast<int32_t>(PD);
            ra.capl         = 13 - l1;
            ra.shift        = bits - l1;  // level 1: the top l1 bits of the scrambled index
            ra.slices       = static_cast<int32_t>(S);
            ra.nseg         = 1;
            ra.region_cap   = capR;
            ra.region_count = sc.alloc<int32_t>(static_cast<size_t>(S * PD));
            ra.overflow     = d_overflow;
            ra.out_val      = sc.alloc<uint64_t>(static_cast<size_t>(S * PD) * static_cast<size_t>(capR));
            ra.tag16        = two_level ? 0 : 1;
            ra.out_tag      = sc.alloc<uint16_t>(static_cast<size_t>(S * PD) * static_cast<size_t>(capR) * (two_level ? 2 : 1));
            bool const ring_hot = !hot_keys.empty();
            if (ring_hot) {  // (hot_eligible: one plain key, SUM / COUNT accumulators)
              uint64_t* d_hot = sc.alloc<uint64_t>(HOT_MAX_KEYS);
              CUDF_HIP_TRY(hipMemcpyAsync(d_hot, hot_keys.data(), hot_keys.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s));
              ra.hot_n     = static_cast<int32_t>(hot_keys.size());
              ra.hot_keys  = d_hot;
              ra.hot_out   = sc.alloc<uint64_t>(static_cast<size_t>(S) * HOT_SLOTS * PU);
              ra.hot_count = sc.alloc<int32_t>(static_cast<size_t>(S));
            }
            if (dm.nkeys > 0) {
              uint32_t* ones = sc.alloc<uint32_t>(16);
              CUDF_HIP_TRY(hipMemsetAsync(ones, 0xff, 64, s));
              ra.ones = ones;
            }
            dense_ring_args* d_ra = sc.alloc<dense_ring_args>(1);
            store_args(ra, d_ra, s);
            dense_agg_args da{};
            da.plan         = p;
            da.map          = dm;
            da.rec_val      = ra.out_val;
            da.rec_tag      = static_cast<uint16_t const*>(ra.out_tag);
            da.region_count = ra.region_count;
            da.region_cap   = capR;
            da.slices       = static_cast<int32_t>(S);
            int64_t cap2    = 0;
            dense_ring_args rb{};
            dense_ring_args* d_rb = nullptr;
            if (two_level) {
              // level 2: work item (g, s) reads level-1 partition g as the strided list of its regions s, s + slices2, ... and appends
              // to the regions of the global partitions g * P2 + d (the next l2 bits); the aggregate walks those
              int64_t const slices2 = std::max<int64_t>(1, 512 / PD);
              cap2 = region_cap_for(static_cast<double>(n) / static_cast<double>(PD * slices2 * P2D), static_cast<double>(PD * P2D));
              size_t const nreg2 = static_cast<size_t>(PD * P2D * slices2);
              rb                 = ra;
              rb.from_columns    = 0;
              rb.hot_n           = 0;  // (the heavy hitters left the stream on the first level)
              rb.P               = static_cast<int32_t>(P2D);
              rb.capl            = 13 - l2;
              rb.shift           = bits - rlog2P;  // the l2 bits below the level-1 digit
              rb.slices          = static_cast<int32_t>(slices2);
              rb.nseg            = static_cast<int32_t>(PD);
              rb.in_val          = ra.out_val;
              rb.in_tag          = static_cast<uint32_t const*>(ra.out_tag);
              rb.in_region_count = ra.region_count;
              rb.in_region_cap   = capR;
              rb.in_slices       = static_cast<int32_t>(S);
              rb.region_cap      = cap2;
              rb.region_count    = sc.alloc<int32_t>(nreg2);
              rb.out_val         = sc.alloc<uint64_t>(nreg2 * static_cast<size_t>(cap2));
              rb.tag16           = 1;
              rb.out_tag         = sc.alloc<uint16_t>(nreg2 * static_cast<size_t>(cap2));
              d_rb               = sc.alloc<dense_ring_args>(1);
              store_args(rb, d_rb, s);
              da.rec_val      = rb.out_val;
              da.rec_tag      = static_cast<uint16_t const*>(rb.out_tag);
              da.region_count = rb.region_count;
              da.region_cap   = cap2;
              da.slices       = static_cast<int32_t>(slices2);
            }
            da.nsplit       = nsplit;
            da.slots        = slots;
            da.image_bytes  = static_cast<int32_t>(image);
            da.occ_acc      = dense_occ_acc(p);
            da.KU           = dm.nkeys > 0 ? p.KU : 1;
            nitems          = ntables * nsplit;  // (partial records: partition d's slots in nsplit shares)
            da.tables       = sc.alloc<uint64_t>(nsplit > 1 ? static_cast<size_t>(nitems) * image / 8 : 2);
            // (+ one item of the same stride for the merged heavy hitters)
            partial         = sc.alloc<uint64_t>((static_cast<size_t>(ntables) * slots + static_cast<size_t>(slots / nsplit)) * DPU);
            d_count         = sc.alloc<int32_t>(nitems + 1);
            da.out_records  = partial;
            da.out_count    = d_count;
            da.overflow     = d_overflow;
            da.nitems       = ntables;
            da.block        = 1024;
            dense_agg_args* d_da = sc.alloc<dense_agg_args>(1);
            store_args(da, d_da, s);
            launch_dense_ring_scatter(ra, d_ra, s);
            if (two_level) launch_dense_ring_scatter(rb, d_rb, s);
            launch_aggregate_dense(da, d_da, true, true, s);
            if (nsplit > 1) launch_dense_merge_dump(da, d_da, nsplit, s);
            final_cap          = slots / nsplit;
            if (ring_hot) {  // the workgroups' heavy-hitter partials -> one more item behind the tables' (hash-table merge kernel)
              CUDF_EXPECTS(final_cap >= HOT_MAX_KEYS, "dense keys: heavy-hitter item");
              agg_args hm{};
              hm.plan        = p;
              hm.geom        = ag;
              hm.overflow    = d_overflow;
              hm.input       = IN_PARTIAL_RECORDS;
              hm.seg         = SEG_STRIDED;
              hm.records     = ra.hot_out;
              hm.src_count   = ra.hot_count;
              hm.src_stride  = HOT_SLOTS;
              hm.fan         = static_cast<int32_t>(S);
              hm.nsrc        = static_cast<int32_t>(S);
              hm.out_records = partial + static_cast<size_t>(nitems) * final_cap * DPU;
              hm.out_count   = d_count + nitems;
              hm.nitems      = 1;
              launch_aggregate(hm, sc.alloc<agg_args>(1), s);
              nitems += 1;
            }
            int32_t const h_ov = overflow_and_counts();
            if (env.debug)
              fprintf(stderr, "[cudf_amd] dense keys (ring): nkeys=%d lo=%lld range=%llu bits=%d P=%ld x %ld slots=%d image=%zu B nsplit=%d capR=%ld cap2=%ld overflow=%d\n",
                      dm.nkeys, (long long)dm.lo, (unsigned long long)dm.range, bits, (long)PD, (long)P2D, slots, image, nsplit, (long)capR, (long)cap2, h_ov);
            if (h_ov == 0) return outcome::done;
            // a region overflowed (skewed or clustered keys) or a key lay outside the sampled range: redo by hash
            allow_dense = false;
            final_cap   = 0;
            fresh_scratch();
            return outcome::retry_free;
          }
        }
  return outcome::skip;
}

// ---- several value columns (dense_multi_kernels.hip): one ring-scatter pass over the input writes one value stream per column
// next to the shared 16-bit tags; the dense aggregate then runs once per COLUMN over (tags, that column's values) with the
// column's accumulators only - its table image is as small as for a single column, so the fan-out stays at 128 - and
// k_dense_merge_dump_multi folds the columns' images into the partial records of the whole plan.
// Reference: all (column, aggregation) pairs of a call in one pass (compute_global_memory_aggs.cuh:139-147).
outcome aggregate_call::try_dense_ring_multi()
{
  int const nval = p.NPAY;
  dense_map dm{};
  if (!dense_map_from_sample(dm)) return outcome::skip;
  int bits = 14;
  while (bits < 31 && (uint64_t{1} << bits) < dm.range) ++bits;
  // ---- per-column sub-plans: the accumulators that read column c; the row counts (no column of their own) go with the column
  // whose accumulators take the fewest bytes, so that no table grows past the one-column size without need
  plan_dev sub[RING_MAX_VALUES];
  dense_multi_merge_args ma{};
  int acc_bytes[RING_MAX_VALUES] = {0, 0, 0};
  for (int q = 0; q < p.NACC; ++q)
    if (p.acc[q].pay >= 0 && p.acc[q].pay < nval) acc_bytes[p.acc[q].pay] += acc_is_narrow(p.acc[q].op, p.acc[q].src) ? 4 : 8;
  int count_col = 0;
  for (int c = 1; c < nval; ++c)
    if (acc_bytes[c] < acc_bytes[count_col]) count_col = c;
  for (int c = 0; c < nval; ++c) {
    plan_dev& sp = sub[c];
    sp           = p;
    sp.NPAY      = 1;
    sp.NACC      = 0;
    sp.unit[1]        = p.unit[1 + c];
    sp.simple_base[1] = p.simple_base[1 + c];
    for (int q = 0; q < p.NACC; ++q) {
      acc_desc const& d = p.acc[q];
      if (!(d.pay == c || (d.pay < 0 && c == count_col))) continue;
      acc_desc e = d;
      if (e.pay >= 0) e.pay = 0;
      ma.acc_col[q]    = static_cast<int8_t>(c);
      ma.acc_narrow[q] = acc_is_narrow(d.op, d.src) ? 1 : 0;
      ma.acc_off[q]    = static_cast<uint32_t>(sp.NACC);  // (its index in the sub-plan for now: the byte offset once the tables are sized)
      sp.acc[sp.NACC++] = e;
    }
    if (sp.NACC == 0) return outcome::skip;  // (a value column without an accumulator of its own: not a plan this path expects)
  }
  int const rlog2P = 7;
  int const slots  = 1 << std::max(bits - rlog2P, 0);
  std::size_t image[RING_MAX_VALUES];
  bool fits = true;
  for (int c = 0; c < nval; ++c) {
    image[c] = dense_table_bytes(sub[c], slots);
    fits     = fits && image[c] <= 150 * 1024;
  }
  int const cap = dense_ring_multi_cap(nval);
  int64_t const PD = int64_t{1} << rlog2P, S = 256;
  bool const ok = fits && dm.range <= (uint64_t{1} << bits) && bits <= 30 && bits - rlog2P >= 6 && bits - rlog2P <= 15 &&
                  static_cast<double>(dm.range) <= 8.0 * std::max(est_groups, 4096.0) &&
                  dense_ring_multi_lds_bytes(nval, static_cast<int>(PD), cap) + 64 <= 160 * 1024;
  if (!ok) return outcome::skip;
  path        = hash_path::DENSE_DIRECT;
  dm.mult     = 0x9E3779B1u;  // odd: index -> (index * mult) mod 2^bits is a bijection
  uint32_t inv = dm.mult;     // Newton: inv = mult^-1 mod 2^32
  for (int it = 0; it < 5; ++it) inv *= 2u - dm.mult * inv;
  dm.mult_inv = inv;
  dm.bits     = bits;
  dm.log2P    = rlog2P;
  int const DPU = 1 + p.NACC;
  // (workgroup w takes the row tiles w, w + S, ...: the busiest workgroup has ceil(tiles / S) of them)
  int64_t const ring_tile = nval == 2 ? 2 * 1024 : 1024, wg_rows = std::min<int64_t>(n, ((n + ring_tile - 1) / ring_tile + S - 1) / S * ring_tile);
  double const mean       = static_cast<double>(wg_rows) / static_cast<double>(PD);
  double const keys_per_p = std::max(1.0, 0.5 * est_groups / static_cast<double>(PD));
  double const rel_sigma  = std::sqrt(1.0 / keys_per_p + 1.0 / std::max(1.0, mean));
  int64_t const capR      = (static_cast<int64_t>(mean * (1.0 + 6.0 * std::min(rel_sigma, 1.0)) + 64.0) + 63) / 64 * 64;
  ring_multi_args ra{};
  ra.plan          = p;
  ra.map           = dm;
  ra.nrows         = n;
  ra.nval          = nval;
  ra.P             = static_cast<int32_t>(PD);
  ra.cap           = cap;
  ra.shift         = bits - rlog2P;
  ra.slices        = static_cast<int32_t>(S);
  ra.region_cap    = capR;
  ra.stream_stride = S * PD * capR;
  ra.region_count  = sc.alloc<int32_t>(static_cast<size_t>(S * PD));
  ra.overflow      = d_overflow;
  ra.out_val       = sc.alloc<uint64_t>(static_cast<size_t>(ra.stream_stride) * nval);
  ra.out_tag       = sc.alloc<uint16_t>(static_cast<size_t>(ra.stream_stride));
  ring_multi_args* d_ra = sc.alloc<ring_multi_args>(1);
  store_args(ra, d_ra, s);
  launch_dense_ring_scatter_multi(ra, d_ra, s);
  int32_t const ntables = static_cast<int32_t>(PD);
  int const dsplit      = 2;  // (merge items: two per partition, 256 workgroups)
  // (one aggregate launch per column: a partition's regions are shared out to two workgroups so that every launch has 256 of
  // them - with one per partition half of the CUs idled: 3.45 instead of 1.76 ms per column at 1B rows)
  int const nsplit      = static_cast<int>(std::clamp<int64_t>(env.dense_nsplit >= 0 ? env.dense_nsplit : 256 / ntables, 1, 16));
  nitems  = ntables * dsplit;
  partial = sc.alloc<uint64_t>(static_cast<size_t>(ntables) * slots * DPU);
  d_count = sc.alloc<int32_t>(nitems);
  for (int c = 0; c < nval; ++c) {
    dense_agg_args da{};
    da.plan         = sub[c];
    da.map          = dm;
    da.rec_val      = ra.out_val + static_cast<int64_t>(c) * ra.stream_stride;
    da.rec_tag      = ra.out_tag;
    da.region_count = ra.region_count;
    da.region_cap   = capR;
    da.slices       = static_cast<int32_t>(S);
    da.nsplit       = nsplit;
    da.keep_images  = 1;
    da.slots        = slots;
    da.image_bytes  = static_cast<int32_t>(image[c]);
    da.occ_acc      = dense_occ_acc(sub[c]);
    da.KU           = 1;
    da.tables       = sc.alloc<uint64_t>(static_cast<size_t>(ntables) * nsplit * image[c] / 8);
    da.out_records  = partial;  // (not written: the images are folded by the merge below)
    da.out_count    = d_count;
    da.overflow     = d_overflow;
    da.nitems       = ntables;
    da.block        = 1024;
    dense_agg_args* d_da = sc.alloc<dense_agg_args>(1);
    store_args(da, d_da, s);
    launch_aggregate_dense(da, d_da, true, true, s);
    ma.tables[c]      = da.tables;
    ma.image_bytes[c] = da.image_bytes;
  }
  for (int q = 0; q < p.NACC; ++q) ma.acc_off[q] = dense_acc_offset(sub[ma.acc_col[q]], slots, static_cast<int>(ma.acc_off[q]));
  ma.plan        = p;
  ma.map         = dm;
  ma.ncols       = nval;
  ma.nsplit      = nsplit;
  ma.slots       = slots;
  ma.occ_acc     = dense_occ_acc(p);
  ma.occ_off0    = dense_occ_offset(sub[0], slots);
  ma.out_records = partial;
  ma.out_count   = d_count;
  ma.nitems      = ntables;
  dense_multi_merge_args* d_ma = sc.alloc<dense_multi_merge_args>(1);
  store_args(ma, d_ma, s);
  launch_dense_merge_dump_multi(ma, d_ma, dsplit, s);
  final_cap          = slots / dsplit;
  int32_t const h_ov = overflow_and_counts();
  if (env.debug)
    fprintf(stderr, "[cudf_amd] dense keys (ring, %d value columns): lo=%lld range=%llu bits=%d P=%ld slots=%d ring cap=%d capR=%ld overflow=%d\n", nval,
            (long long)dm.lo, (unsigned long long)dm.range, bits, (long)PD, slots, cap, (long)capR, h_ov);
  if (h_ov == 0) return outcome::done;
  // a region overflowed (skewed or clustered keys) or a key lay outside the sampled range: redo by hash
  allow_dense = false;
  final_cap   = 0;
  fresh_scratch();
  return outcome::retry_free;
}

// ---- write-combining scatter with 16-byte records (CUDF_AMD_GB_DENSE_RING=0, or a geometry the rings do not take). (Rounds 2-3 could
// also run it chunk by chunk through the Infinity Cache; measured slower - profiles/r2_mall_pipeline.txt - and removed in round 4.)
outcome aggregate_call::try_dense_wc()
{
  bool const ring_env = env.dense_ring;
  if (!(allow_dense && n >= env.big_min_rows && env.forced_p == 0 && (hot_keys.empty() || (dense_candidate && ring_env)) && !env.exact)) return outcome::skip;
  if (dense_candidate && p.NPAY > 1) return outcome::skip;  // (16-byte records carry one value)
      dense_map dm{};
      bool dense_ok = dense_map_from_sample(dm);
      int bits = 14;
      while (bits < 31 && (uint64_t{1} << bits) < dm.range) ++bits;
      {
        double const unit_bytes = static_cast<double>(dense_table_bytes(p, 4096)) / 4096.0;  // LDS bytes per key of the range
        // One level: tables of ~32 KiB (two 1024-thread workgroups per CU), 256 to 1024 of them. A range that needs more than
        // 1024 tables of 150 KiB takes two levels (P1 x P2) with the largest tables that fit.
        int log2P = 8;
        while (log2P < 10 && std::ldexp(unit_bytes, bits - log2P) > 32.0 * 1024) ++log2P;
        while (log2P < 20 && std::ldexp(unit_bytes, bits - log2P) > 150.0 * 1024) ++log2P;
        if (env.dense_log2p > 0) log2P = static_cast<int>(env.dense_log2p);
        bool const two_level      = log2P > 10;
        int const log2P1          = two_level ? (log2P + 1) / 2 : log2P;
        int const log2P2          = log2P - log2P1;
        int const slots           = 1 << std::max(bits - log2P, 0);
        std::size_t const image   = dense_table_bytes(p, slots);
        dense_ok = dense_ok && dm.range <= (uint64_t{1} << bits) && bits <= 30 && bits - log2P >= 6 && log2P <= 20 && image <= 150 * 1024 &&
                   static_cast<double>(dm.range) <= 8.0 * std::max(est_groups, 4096.0);
        int64_t const PD = int64_t{1} << log2P1, P2D = int64_t{1} << log2P2;
        // scatter workgroups: one of 1024 threads per CU (128-byte granules up to 512 partitions), or - CUDF_AMD_GB_SCATTER_BLOCK=512 -
        // two of 512 threads per CU with 64-byte granules (measured slower: profiles/r2_mall_pipeline.txt)
        int const SB     = env.scatter_block == 512 ? 512 : 1024;
        int const GD     = static_cast<int>((env.wc_g >= 0 ? env.wc_g : ((PD > 512 || SB == 512) ? 4 : 8)));
        if (dense_ok && partition_wc_fits(2, static_cast<int>(PD), GD, SB) && (!two_level || partition_wc_fits(2, static_cast<int>(P2D), P2D > 512 ? 4 : 8))) {
          path  = hash_path::DENSE_DIRECT;
          dm.mult     = 0x9E3779B1u;  // odd: index -> (index * mult) mod 2^bits is a bijection
          uint32_t inv = dm.mult;     // Newton: inv = mult^-1 mod 2^32
          for (int it = 0; it < 5; ++it) inv *= 2u - dm.mult * inv;
          dm.mult_inv = inv;
          dm.bits     = bits;
          dm.log2P    = log2P;
          int const DPU = (dm.nkeys > 0 ? p.KU : 1) + p.NACC;  // units of a dumped partial record
          int64_t const S        = 256 * (1024 / SB), tile_rows = 5 * SB;
          double const cell_mean = static_cast<double>(n) / static_cast<double>(S * PD);
          double const keys_per_p = std::max(1.0, 0.5 * est_groups / static_cast<double>(PD));
          double const rel_sigma  = std::sqrt(1.0 / keys_per_p + 1.0 / std::max(1.0, cell_mean));
          int64_t const capR      = (static_cast<int64_t>(cell_mean * (1.0 + 6.0 * std::min(rel_sigma, 1.0)) + 16.0) + 7) / 8 * 8;
          part_args pa{};
          pa.plan          = p;
          pa.geom.nseg     = 1;
          pa.geom.slices   = static_cast<int32_t>(S);
          pa.geom.P        = static_cast<int32_t>(PD);
          pa.geom.shift    = bits - log2P1;  // level 1: the top log2P1 bits of the scrambled index
          pa.geom.block    = SB;
          pa.geom.tile_rows = static_cast<int32_t>(tile_rows);
          pa.from_columns  = 1;
          pa.nrows         = n;
          pa.optimistic    = 1;
          pa.region_cap    = capR;
          pa.region_count  = sc.alloc<int32_t>(static_cast<size_t>(S * PD));
          pa.overflow      = d_overflow;
          pa.out_records   = sc.alloc<uint64_t>(static_cast<size_t>(S * PD) * static_cast<size_t>(capR) * 2);
          pa.wc_granule    = GD;
          pa.cyclic_tiles  = 1;
          pa.use_dense     = 1;
          pa.dense         = dm;
          part_args* d_pa  = sc.alloc<part_args>(1);
          store_args(pa, d_pa, s);
          dense_agg_args da{};
          da.plan         = p;
          da.map          = dm;
          da.records      = pa.out_records;
          da.region_count = pa.region_count;
          da.region_cap   = capR;
          da.slices       = static_cast<int32_t>(S);
          int64_t cap2    = 0;
          part_args pb{};
          part_args* d_pb = nullptr;
          if (two_level) {
            // level 2: work item (g, s) reads level-1 partition g as the strided list of its regions s, s + slices2, ... and appends
            // to the regions of the global partitions g * P2 + d (the next log2P2 bits); the aggregate walks those
            int64_t const slices2 = std::max<int64_t>(1, 512 / PD);
            double const mean2    = static_cast<double>(n) / static_cast<double>(PD * slices2 * P2D);
            double const sigma2   = std::sqrt(1.0 / std::max(1.0, 0.5 * est_groups / static_cast<double>(PD * P2D)) + 1.0 / std::max(1.0, mean2));
            cap2                  = (static_cast<int64_t>(mean2 * (1.0 + 6.0 * std::min(sigma2, 1.0)) + 16.0) + 7) / 8 * 8;
            pb.plan            = p;
            pb.geom.nseg       = static_cast<int32_t>(PD);
            pb.geom.slices     = static_cast<int32_t>(slices2);
            pb.geom.P          = static_cast<int32_t>(P2D);
            pb.geom.shift      = bits - log2P;  // the log2P2 bits below the level-1 digit
            pb.geom.block      = 1024;
            pb.geom.tile_rows  = 5 * 1024;
            pb.from_columns    = 0;
            pb.in_records      = pa.out_records;
            pb.from_regions    = 1;
            pb.in_region_count = pa.region_count;
            pb.in_region_cap   = capR;
            pb.in_slices       = static_cast<int32_t>(S);
            pb.optimistic      = 1;
            pb.region_cap      = cap2;
            size_t const nreg2 = static_cast<size_t>(PD * P2D * slices2);
            pb.region_count    = sc.alloc<int32_t>(nreg2);
            pb.overflow        = d_overflow;
            pb.out_records     = sc.alloc<uint64_t>(nreg2 * static_cast<size_t>(cap2) * 2);
            pb.wc_granule      = P2D > 512 ? 4 : 8;
            pb.use_dense       = 1;
            pb.dense           = dm;
            d_pb               = sc.alloc<part_args>(1);
            store_args(pb, d_pb, s);
            da.records      = pb.out_records;
            da.region_count = pb.region_count;
            da.region_cap   = cap2;
            da.slices       = static_cast<int32_t>(slices2);
          }
          da.slots        = slots;
          da.image_bytes  = static_cast<int32_t>(image);
          da.occ_acc      = dense_occ_acc(p);
          da.KU           = dm.nkeys > 0 ? p.KU : 1;
          nitems          = static_cast<int32_t>(int64_t{1} << log2P);
          da.tables       = sc.alloc<uint64_t>(2);
          partial         = sc.alloc<uint64_t>(static_cast<size_t>(nitems) * slots * DPU);
          d_count         = sc.alloc<int32_t>(nitems);
          da.out_records  = partial;
          da.out_count    = d_count;
          da.overflow     = d_overflow;
          da.nitems       = nitems;
          da.block        = 1024;
          dense_agg_args* d_da = sc.alloc<dense_agg_args>(1);
          store_args(da, d_da, s);
          launch_partition_scatter(pa, d_pa, s);
          if (two_level) launch_partition_scatter(pb, d_pb, s);
          launch_aggregate_dense(da, d_da, true, true, s);
          final_cap          = slots;
          int32_t const h_ov = overflow_and_counts();
          if (env.debug)
            fprintf(stderr, "[cudf_amd] dense keys: nkeys=%d lo=%lld range=%llu bits=%d P=%ld x %ld slots=%d image=%zu B capR=%ld cap2=%ld overflow=%d\n",
                    dm.nkeys, (long long)dm.lo, (unsigned long long)dm.range, bits, (long)PD, (long)P2D, slots, image, (long)capR, (long)cap2, h_ov);
          if (h_ov == 0) return outcome::done;
          // a region overflowed (skewed or clustered keys) or a key lay outside the sampled range: redo by hash
          allow_dense = false;
          final_cap   = 0;
          fresh_scratch();
          return outcome::retry_free;
        }
      }
  return outcome::skip;
}

}  // namespace cudf::groupby::detail
