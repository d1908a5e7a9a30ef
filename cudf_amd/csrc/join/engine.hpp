// SPDX-License-Identifier: Apache-2.0
// Hash-join engine for MI355X: open-addressing multiset of {32-bit hash tag, build row} packed in one 64-bit
// slot, linear probing, built with global 64-bit CAS (27 G CAS/s measured, profiles/microbench_r1.txt) and
// probed with 64-lane wave-aggregated output allocation (one global atomic per wave and round instead of the
// reference's 32-lane ballot + per-warp LDS staging buffer, partitioned_retrieve_kernels.cuh:57-211).
// Replaces cuco::static_multiset<pair<u32,i32>> with double hashing / CG 2 / bucket 2
// (cpp/src/join/hash_join/hash_join_impl.cuh:17-60, hash_join.cu:62-149, retrieve_impl.cuh:29-133,
// size_impl.cuh:26-61).
#pragma once
#include "../common/device_table.hpp"

#include <hip/hip_runtime.h>
#include <cstdint>

namespace cudf::detail::join {

constexpr uint64_t EMPTY_SLOT = ~uint64_t{0};

constexpr int BUILD_SKIP_ENTRIES = 1 << 16, BUILD_SKIP_AFTER = 32;
struct join_args {
  device_table build;
  device_table probe;
  uint64_t* table;         // capacity 8-byte slots {32-bit hash tag | build row}, walked in aligned pairs
  // Build only: BUILD_SKIP_ENTRIES advisory hints {44 hash bits | n}: "the first n steps of this hash's probe sequence are
  // full" (kernels.hip k_build, seq_next). An insert that has walked BUILD_SKIP_AFTER steps looks its hint up and jumps
  // ahead; an insert that ended further than that along its sequence leaves one.
  uint64_t* build_skip;
  uint64_t capacity;
  int32_t nulls_equal;     // null_equality::EQUAL
  int32_t check_nulls;     // some key column (either side) has nulls
  int32_t kind;            // 0 inner, 1 left, 2 full
  int32_t single64;        // fast path: one 8-byte integer key column, no nulls on either side
  // outputs
  unsigned long long* total;   // match-pair counter (count pass) / output cursor (retrieve pass)
  size_type* out_probe;        // probe-side row of each pair
  size_type* out_build;        // build-side row of each pair (JoinNoMatch for unmatched probe rows)
  uint64_t out_capacity;       // pairs that fit in out_probe/out_build
  uint8_t* build_matched;      // full join: build rows seen by some probe row
  // probe rows are cut into `nblocks` contiguous chunks, one per workgroup, identical in the count and the
  // retrieve pass: block_offsets[b] (exclusive scan of the per-block pair counts) makes the output position of
  // every pair a block-local matter — no global atomic on the probe path
  int32_t nblocks;
  int64_t chunk;
  unsigned long long* block_counts;   // [nblocks + 1]; after launch_scan: exclusive offsets, [nblocks] = total
  // Per-probe-row result of the count pass: MATCH_NONE, or the build row of the first match with bit 31 set when
  // there are further matches. The retrieve pass streams this array (coalesced) and only re-walks the table for
  // rows with several matches, instead of repeating 500M random probes (C3: 29 ms -> ~2 ms).
  uint32_t* match_cache;
  // optional per-probe-row match counts (join_match_context); left/full kinds count a lonely row as 1
  size_type* row_counts;
  // probe rows are rows [probe_row_base, ...) of a larger left table (partitioned_*_join): added to emitted indices
  int64_t probe_row_base;
  // Dense build keys (one 8-byte integer key column whose valid values span a small range; NULLs never match): a
  // direct-address table instead of the hash table. Unique build keys (dense_has_dups == 0): dense_head[key - dense_lo] = the build
  // row with that key (-1: none). Some key repeats (dense_has_dups == 1): dense_head[i] .. dense_head[i + 1] bound key i's list of
  // build rows in dense_next (launch_dense_csr). During the build, dense_next[row] = the previous build row with the same key.
  // A probe is one range test (keys outside [lo, lo + range) touch no memory at all) and one 4-byte load; no hash, no tag, no
  // key comparison. The reference probes cuco::static_multiset for every row (retrieve_impl.cuh:29-133).
  int32_t* dense_head;
  int32_t* dense_next;
  uint64_t dense_lo;
  uint64_t dense_range;
  int32_t dense_has_dups;  // some key occurs more than once on the build side (chains longer than one)
  int32_t* dense_dups;     // build only: device flag behind dense_has_dups
  // Hot build keys (row lists of more than BIG_LIST rows): the retrieve pass only RESERVES the output range of such a probe row in
  // its workgroup's share and appends {probe row, first list entry, entries, output position} to this work list; k_dense_big_emit
  // then writes those pairs with the WHOLE grid (1000 probes of a key with 100,000 build rows that sat in one workgroup's chunk
  // of probe rows kept that one workgroup busy for 120 ms). big_count[0] = entries appended (may exceed BIG_LIST_CAP: the rows
  // beyond it were emitted by their own wave).
  uint32_t* big_list;      // BIG_LIST_CAP x 4 words
  uint32_t* big_count;
};
constexpr uint32_t BIG_LIST = 1024, BIG_LIST_CAP = 1u << 16;
constexpr uint32_t MATCH_NONE  = 0xffffffffu;
constexpr uint32_t MATCH_MULTI = 0x80000000u;

// ---- LDS radix join (radix_kernels.hip; round 3): one 8-byte integer key whose NULLs never match, inner join, big inputs.
// BOTH sides are partitioned on the top bits of the key hash - two ring-scatter levels (the groupby's ring scatter: per-partition
// rings in LDS, one returning ds_add per row, whole 128-byte granules only) of {key (8 B), row id (4 B)} rows in two streams -
// into partitions whose BUILD rows fit an open-addressing table in LDS (8192 slots of key + row, load ~0.35); one workgroup per
// partition then builds that table and streams the partition's probe rows past it: every probe is an LDS access instead of a
// random 16-byte read into a GB-sized table (475M of them at the fabric's 45 G requests/s were 15.6 of C3-sparse's 19.5 ms).
// Pairs come out partition-major (the reference promises no order either). Replaces, for such joins, cuco::static_multiset
// build + probe (cpp/src/join/hash_join/hash_join.cu:62-149, retrieve_impl.cuh:29-133, size_impl.cuh:26-61).
// Region (q, w) of a level with S slices lies at [(q * S + w) * region_cap, + region_count[q * S + w]) of both streams.
constexpr int RADIX_RING_SLOTS = 8192;  // key-ring slots of a scatter workgroup (all partitions); the row rings are twice as long
struct radix_scatter_args {
  int32_t level;      // 1: rows of a key column; 2: the regions of level-1 partition `seg`
  // level 1
  uint64_t const* keys;        // element i = key of row i (key_width bytes each)
  int32_t key_width;           // 8, or 4: a 4-byte integer column, widened to 64 bits by the scatter (key_signed: sign-extended)
  int32_t key_signed;
  int32_t key_class;           // CLS_F32 / CLS_F64: a float column - its normalised bits (-0 -> +0, one NaN) are the key; 0: integers
  // two 4-byte integer key columns (keys, keys2), packed into the 8-byte key; mask2: the second column's validity (nullptr: none)
  uint32_t const* keys2;
  bitmask_type const* mask2;
  int64_t mask2_offset;
  // kw == 2 (round 4): TWO key columns of 4 or 8 bytes each, any class - the row's key is the pair of their (normalised, zero-extended)
  // bits, carried as two 8-byte streams (out_key / out_key1); key2 / key2_width / key2_class describe the second column, mask2 its validity
  int32_t kw;                  // 8-byte words of a key: 1 (default: 0 is read as 1) or 2
  void const* key2;
  int32_t key2_width, key2_class;
  // pack != 0 (kw == 1, level 1): two INTEGER key columns (keys / key2) whose build-side ranges fit 63 bits together travel as ONE word,
  // (c0 - pack_lo0) << pack_bits1 | (c1 - pack_lo1) - a bijection on the build side's value box, so the single-word join is exact. A probe
  // row outside the box matches nothing: dropped, or (pack_keep_outside: left joins) kept under a word with bit 63 set, which no build
  // word has. Columns are widened like `keys` (key_signed / key2_signed for 4-byte columns); lo / range are in that 64-bit arithmetic.
  int32_t pack, pack_bits1, pack_keep_outside, key2_signed;
  uint64_t pack_lo0, pack_lo1, pack_range0, pack_range1;  // range = max - min
  bitmask_type const* mask;    // validity of the key column (bit mask_offset + i), nullptr: no NULLs; NULL rows are dropped
  int64_t mask_offset;
  int64_t nrows;
  // level 2: work item (seg, s) reads the level-1 regions (seg, w), w = s, s + slices, ... < in_slices as one virtual row range
  uint64_t const* in_key;
  uint64_t const* in_key1;     // kw == 2: the second key word
  uint32_t const* in_row;
  int32_t const* in_region_count;
  int64_t in_region_cap;
  int32_t in_slices;
  int32_t nseg;
  // both
  int32_t P;          // fan-out: a power of two, 16 ... 256
  int32_t capl;       // log2 of the ring capacity per partition: P << capl = RADIX_RING_SLOTS / kw
  int32_t shift;      // digit = (key hash >> shift) & (P - 1)
  int32_t slices;
  uint64_t* out_key;
  uint64_t* out_key1;          // kw == 2
  uint32_t* out_row;
  int64_t region_cap; // a multiple of 32 records
  int32_t* region_count;
  int32_t* overflow;  // bit 0: a region overflowed
  // Dense mode (level 1; dense_part_args below): rows become ONE stream of 8-byte records {key - dense_lo | row id << 32} in
  // out_key, partitioned by the top bits of the offset (digit = offset >> shift); keys outside the range are dropped.
  int32_t dense;
  uint64_t dense_lo, dense_range;
  int32_t pending_budget;  // dense: a workgroup whose rings made it wait more often than this (net of tiles) reports overflow
  int32_t rpt;             // dense, block 1024: rows per thread and tile, 4 (tiles of 4096 rows) or 8
  int32_t block;           // dense: threads per workgroup, 1024 (tiles of 4096 rows) or 512 (2048 rows; two workgroups per CU when the rings take <= 64 KB)
};
struct radix_join_args {
  uint64_t const* b_key;   // build partitions: regions (q, s), s < b_slices
  uint64_t const* b_key1;  // kw == 2: the second key word of the build / probe records (p_key1)
  uint64_t const* p_key1;
  int32_t kw;              // 1 (0 is read as 1): the table's slot state is the key; 2: the state is a 64-bit hash of the two words, a slot
                           // points at the build record and every candidate is verified against the partition's records (L2-resident)
  uint32_t const* b_row;
  int32_t const* b_count;
  int64_t b_cap;
  int32_t b_slices;
  uint64_t const* p_key;   // probe partitions
  uint32_t const* p_row;
  int32_t const* p_count;
  int64_t p_cap;
  int32_t p_slices;
  int32_t nparts;
  int32_t cap;             // LDS table slots (a power of two)
  int32_t fill_limit;      // build rows a table takes before the partition reports overflow
  unsigned long long* pair_counts;  // [nparts + 1]: pairs per partition (count pass), then their exclusive prefix (launch_scan)
  // The count pass also STAGES the pairs it finds - {probe row, build row} as one 8-byte word, partition q's pairs at
  // stage[q * stage_cap ...) - so that the retrieve pass is a copy (launch_radix_emit_staged) instead of a second build + probe;
  // a partition with more pairs than stage_cap (duplicated keys) sets bit 2 of *overflow and is joined again by the retrieve pass.
  uint64_t* stage;
  int64_t stage_cap;
  size_type* out_probe;
  size_type* out_build;
  uint64_t out_capacity;
  int64_t probe_row_base;
  int32_t* overflow;       // bit 1: a build partition does not fit its table; bit 2: a partition's pairs did not fit its stage
  int32_t left;            // left join: a probe record without a partner yields {probe row, JoinNoMatch}
};
// left join: the pairs {row, JoinNoMatch} of the probe rows with a NULL key (bit clear in mask), appended at *cursor (device word
// holding the number of pairs written so far)
void launch_radix_null_rows(bitmask_type const* mask, int64_t mask_offset, int64_t nrows, int64_t row_base, size_type* out_probe, size_type* out_build,
                            unsigned long long out_capacity, unsigned long long* cursor, hipStream_t stream);
void launch_radix_scatter(radix_scatter_args const& a, radix_scatter_args* d_args, hipStream_t stream);
// the largest number of rows of one partition (regions summed over its slices) -> *out_max (preset to 0)
void launch_radix_partition_max(int32_t const* region_count, int32_t nparts, int32_t slices, int32_t* out_max, hipStream_t stream);
void launch_radix_join(radix_join_args const& a, radix_join_args* d_args, bool retrieve, hipStream_t stream);
// copies the staged pairs of every partition that fit its stage to out_probe / out_build (pair_counts holds the exclusive prefix)
void launch_radix_emit_staged(radix_join_args const& a, radix_join_args* d_args, hipStream_t stream);

// ---- Partitioned dense join (dense_part_kernels.hip; round 3): big inner joins on a dense UNIQUE build key. Both sides go through
// launch_radix_scatter's dense mode - 8-byte records {key - dense_lo | row << 32} in regions (p, s), p = offset >> shift: one
// contiguous slice of the direct-address table per partition - so that the random accesses of a partition stay inside one XCD's L2.
struct dense_part_args {
  uint64_t const* recs;         // regions (p, s) at (p * S + s) * region_cap
  int32_t const* region_count;  // [P * S]
  int64_t region_cap;
  int32_t P, S;                 // partitions of the region layout (the scatter's ring count: a power of two), slices
  int32_t P_used;               // partitions that can hold rows: (range - 1 >> shift) + 1 <= P; eight per launch
  int32_t* head;                // the direct-address table (join_args::dense_head, unique keys: head[offset] = build row or -1)
  int32_t const* overflow;      // the scatter's flag: nothing is done when it is set
  // lookup only: workgroup b stages the pairs of all its regions at stage[b * stage_cap ...), pair_counts[b] = their number
  unsigned long long* pair_counts;  // [dense_part_grid() + 1]
  uint64_t* stage;
  int64_t stage_cap;                // >= dense_part_regions_per_workgroup(P, S) * region_cap
  int64_t probe_row_base;
};
int32_t dense_part_grid();
int64_t dense_part_regions_per_workgroup(int32_t P, int32_t S);
// ---- Ordered direct probe of a dense UNIQUE table (dense_part_kernels.hip k_dense_probe_staged): every WAVE walks its own
// contiguous range of probe rows in order, looks the keys up directly (no partition pass) and appends its pairs to its own stage
// (no atomics): pair_counts[w] pairs at stage[w * wave_rows ...). The copy (launch_radix_emit_staged, nparts = waves) then yields
// the pairs in PROBE-ROW ORDER - which is what makes the caller's payload gather by them fast.
struct dense_stage_args {
  uint64_t const* keys;         // key_width bytes per row
  int32_t key_width, key_signed;
  bitmask_type const* mask;     // nullptr: no NULLs
  int64_t mask_offset;
  int64_t nrows;
  int64_t wave_rows;            // rows per wave (a multiple of 64)
  int32_t nwaves;               // ceil(nrows / wave_rows): 4 per workgroup
  uint64_t dense_lo, dense_range;
  int32_t const* head;
  uint64_t* stage;              // [nwaves * wave_rows]
  unsigned long long* pair_counts;  // [nwaves + 1]
  int64_t probe_row_base;
};
void launch_dense_probe_staged(dense_stage_args const& a, dense_stage_args* d_args, hipStream_t stream);
// LEFT join against a dense unique table: every probe row yields exactly one pair - its build row or JoinNoMatch - so the output
// is the probe rows in order and needs no count pass, no stage and no compaction: out_probe[i] = i + probe_row_base,
// out_build[i] = head[key_i - lo] (or JoinNoMatch for a NULL key, a key outside the range, an empty entry). Uses dense_stage_args'
// keys / mask / nrows / dense_* / head / probe_row_base.
void launch_dense_left_direct(dense_stage_args const& a, dense_stage_args* d_args, size_type* out_probe, size_type* out_build, hipStream_t stream);
// A strided sample of about `samples` probe rows: out[0] = rows sampled, out[1] = those whose key is valid and inside [dense_lo,
// dense_lo + dense_range) - the share of the probe side that costs the direct probe a random table access (the rest is rejected by the
// range test for free). Uses dense_stage_args' keys / mask / nrows / dense_*.
void launch_dense_inrange_sample(dense_stage_args const& a, dense_stage_args* d_args, int64_t samples, unsigned long long* out, hipStream_t stream);

void launch_dense_part_store(dense_part_args const& a, dense_part_args* d_args, hipStream_t stream);
void launch_dense_count_filled(int32_t const* head, uint64_t n, unsigned long long* out, hipStream_t stream);
void launch_sum_region_counts(int32_t const* counts, int64_t n, unsigned long long* out, hipStream_t stream);  // *out preset to 0
void launch_dense_part_lookup(dense_part_args const& a, dense_part_args* d_args, hipStream_t stream);

void launch_build(join_args const& a, join_args* d_args, hipStream_t stream);
// minimum and maximum of the valid keys of the (single 8-byte integer) build column: out[0] = min, out[1] = max (bit patterns);
// out must hold {max value, min value} of the ordering before (launch_key_minmax initialises it)
void launch_key_minmax(join_args const& a, join_args* d_args, int is_signed, uint64_t* out, hipStream_t stream);
// fills dense_head / dense_next (dense_head preset to -1); sets *dense_dups when a key repeats
void launch_dense_build(join_args const& a, join_args* d_args, hipStream_t stream);
// Some build key repeats: ROW LISTS instead of chains. dense_head (dense_csr_entries(range) zeroed words) becomes the exclusive
// prefix of the keys' row counts - key i's build rows are dense_next[dense_head[i], dense_head[i + 1]) - so that a probe row
// learns its match count from two adjacent words and a hot key's pairs are emitted 64 at a time by the whole wave (a chain had
// to be walked link by link by one lane: 149 ms for 1000 probes of a key with 100,000 build rows). cursor: `range` zeroed words
// (scratch), tile_sums: dense_csr_entries(range) / 8192 words (scratch).
// Reference: cuco::static_multiset keeps duplicates as separate entries along the probe sequence (hash_join.cu:62-99).
std::size_t dense_csr_entries(uint64_t range);
void launch_dense_csr(join_args const& a, join_args* d_args, int32_t* cursor, uint32_t* tile_sums, hipStream_t stream);
void launch_count(join_args const& a, join_args* d_args, hipStream_t stream);
// exclusive scan of block_counts in place (block_counts[nblocks] = total pairs)
void launch_scan(join_args const& a, hipStream_t stream);
void launch_retrieve(join_args const& a, join_args* d_args, hipStream_t stream);
// after launch_retrieve with a.big_list != nullptr: the pairs of the work list, by the whole grid
void launch_dense_big_emit(join_args const& a, join_args* d_args, hipStream_t stream);
// full join: appends (JoinNoMatch, r) for every build row with build_matched[r] == 0
void launch_complement(join_args const& a, join_args* d_args, hipStream_t stream);
// finalize_partitioned_full_join: build_matched[r] = 1 for every r != JoinNoMatch in right_indices[0..n)
void launch_mark_matched(size_type const* right_indices, std::size_t n, uint8_t* build_matched, hipStream_t stream);

}  // namespace cudf::detail::join
