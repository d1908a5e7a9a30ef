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
  uint64_t* table;         // capacity slots of slot_words 64-bit words
  // Build only: BUILD_SKIP_ENTRIES advisory hints {44 hash bits | n}: "the first n steps of this hash's probe sequence are
  // full" (kernels.hip k_build, seq_next). An insert that has walked BUILD_SKIP_AFTER steps looks its hint up and jumps
  // ahead; an insert that ended further than that along its sequence leaves one.
  uint64_t* build_skip;
  uint64_t capacity;
  // slot_words == 2: the build side is one 8-byte integer key column whose NULLs (if any) are never inserted; the
  // key sits next to its {tag | row} entry, so a probe reads ONE 16-byte slot instead of the slot and then a
  // random build-key element
  int32_t slot_words;
  // The table is cut into 2^part_bits slices by the TOP bits of the row hash: home slot = part * slice +
  // (low32(hash) * slice >> 32). A probe side that has been radix-partitioned on the same bits (below) walks one
  // ~3 MB slice at a time, which stays in the XCD's L2 instead of taking an HBM round trip per probe.
  int32_t part_bits;
  uint64_t slice;          // slots per slice (== capacity when part_bits == 0)
  // Partitioned probe (inner join, single 8-byte key): probe records (key, probe row) in per-(partition,
  // workgroup) regions written by k_probe_partition; the count / retrieve passes then take one region per workgroup
  uint64_t* precs;             // 16-byte records
  int32_t* region_count;       // [nparts * pslices]
  int64_t region_cap;          // records per region
  int32_t pslices;             // partition-pass workgroups
  int32_t partitioned;         // count / retrieve passes iterate regions instead of row chunks
  int32_t* overflow;           // a region was too small: fall back to the direct probe
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
// probe-side radix partition (write-combining scatter, common/wc_scatter.hpp) into a.precs / a.region_count
void launch_probe_partition(join_args const& a, join_args* d_args, hipStream_t stream);
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
