// SPDX-License-Identifier: Apache-2.0
// Hash-groupby engine for MI355X: shared host/device plan structures and kernel launchers.
//
// Design: rows are packed into fixed-size RECORDS of 8-byte units
//   raw record     = [KU key units | NPAY payload units]      (payload = values converted to their 8-byte
//                                                               accumulator class, + a validity-flags half)
//   partial record = [KU key units | NACC accumulator units]   (one per (group, work item))
// and aggregated ONLY in LDS hash tables (ds_add_f64 / ds_add_u64 / ds_min/max) — scattered global atomics
// run at ~24 G ops/s on gfx950 (profiles/microbench_r1.txt) and cannot carry 1B rows. When the groups do
// not fit one LDS table the records are first radix-partitioned on the top bits of a 64-bit key hash (one or two
// levels; optimistic write-combining scatter into per-workgroup regions, or histogram / scan / scatter with exact
// offsets as the fallback), so each partition's groups fit one table.
// Replaces the reference's cuco::static_set + global-atomic design
// (cpp/src/groupby/hash/compute_groupby.cu:51-155, compute_global_memory_aggs.cuh:74-187,
// single_pass_functors.cuh:86-157) and its LDS path (compute_mapping_indices.cuh:92-151,
// compute_shared_memory_aggs.cu:260-353).
#pragma once
#include "../common/device_table.hpp"

#include <hip/hip_runtime.h>
#include <cstdint>

namespace cudf::groupby::detail {

using cudf::detail::device_column;
using cudf::detail::MAX_COLS;

constexpr int MAX_KU    = 4;   // key units (8 B each) per record
constexpr int MAX_PAY   = 8;   // raw payload units
constexpr int MAX_ACC   = 12;  // accumulators per group
constexpr int MAX_UNITS = 16;  // units per record (raw or partial)
constexpr int MAX_LOCAL_COLS = 6;  // columns the register-resident record builder resolves (device_common.hpp units_local)

// Source of one 32-bit half of a key/payload unit.
constexpr int8_t H_NONE     = -1;
constexpr int8_t H_KEYNULLS = -2;  // bit c set = key column c is NULL at this row (null_policy::INCLUDE)
constexpr int8_t H_VALVALID = -3;  // bit v set = value column v is valid at this row
constexpr int8_t H_ROWID    = -4;  // the row's index in the input table (ARGMIN / ARGMAX)

struct unit_desc {
  int8_t full;  // 1: the unit is the 8-byte column `lo`
  int8_t lo;    // column index (>= 0) or H_*
  int8_t hi;
  int8_t is_key;
};

// Merge operator of an accumulator (identities: reference device_operators.cuh:60-76,132-139,190-197).
enum acc_op : int8_t { ADD_I64 = 0, ADD_F64, MIN_I64, MIN_U64, MIN_F64, MAX_I64, MAX_U64, MAX_F64, MUL_I64, MUL_F64, ANY_U64 };
// ANY_U64: keeps the value of one (any) contributing row - the un-normalised bits of a float key column, so that the
// output key is a representative INPUT row as in the reference (compute_groupby.cu:104-111) and not the normalised
// +0.0 / canonical NaN the key units carry.
// How a RAW row contributes (reference device_aggregators.cuh:24-112,428-446: null source elements are
// skipped for everything except COUNT_ALL).
enum acc_src : int8_t { SRC_VALUE = 0, SRC_ONE_IF_VALID, SRC_ONE, SRC_SQUARE, SRC_ARG_IDX, SRC_ARG_IDX_OF_MAX, SRC_LO32, SRC_HI32 };
// SRC_LO32 / SRC_HI32: the low 32 bits (zero-extended) / the high 32 bits (sign-extended) of a 64-bit value: their two int64
// sums give the EXACT sum of up to 2^31 int64 values (hi * 2^32 + lo), from which SUM_OVERFLOW takes its overflow flag.
// SRC_ARG_IDX (of a minimum) / SRC_ARG_IDX_OF_MAX: the row-index half of an ARGMIN / ARGMAX pair (arg_desc): untouched by the first sweep, filled by a second
// sweep over the same rows once the extreme value of every group is final.

// COUNT accumulators (adding 1 per row, or partial counts of at most 2^31 - 1 rows) take 4 bytes in the LDS table; the
// partial records still carry them as 8-byte units.
constexpr bool acc_is_narrow(int op, int src) { return op == ADD_I64 && (src == SRC_ONE || src == SRC_ONE_IF_VALID); }

struct acc_desc {
  int8_t op;
  int8_t src;
  int8_t pay;        // raw payload unit holding the value (or -1)
  int8_t valid_bit;  // bit in VALVALID, or -1 if the value column has no nulls
};

// ARGMIN / ARGMAX = (MIN / MAX accumulator of the value, MIN_I64 accumulator of the row index among the rows that
// attain it). Ties resolve to the smallest row index: one of the outcomes of the reference's arrival-order CAS loop
// (device_aggregators.cuh:131-160), made deterministic.
struct arg_desc {
  int8_t valacc;
  int8_t idxacc;
  int8_t is_float;  // compare values numerically (-0.0 == +0.0) instead of bitwise
  int8_t pad;
};
constexpr int MAX_ARG = 4;

struct plan_dev {
  device_column cols[MAX_COLS];  // key columns first, then the distinct value columns
  int32_t ncols;
  int32_t nkeycols;
  int32_t KU;              // key units
  int32_t NPAY;            // raw payload units
  int32_t NACC;            // accumulators
  int32_t drop_null_keys;  // null_policy::EXCLUDE and some key column is nullable
  int32_t flags_unit;      // raw unit holding VALVALID (-1: none)
  int32_t flags_hi;        // 1: in the high half
  int32_t rowid_unit;      // raw unit holding the row index in its low half (-1: none)
  int32_t narg;
  arg_desc arg[MAX_ARG];
  unit_desc unit[MAX_UNITS];
  uint64_t key_mask[MAX_KU];
  acc_desc acc[MAX_ACC];
  // Fast path: every record unit is an 8-byte column without nulls that needs no conversion or normalisation
  // (int64/uint64/float64 values, 8-byte integer keys): unit u of row r is simple_base[u][r].
  int32_t simple;
  int32_t simple_vec16;  // every simple_base pointer is 16-byte aligned: rows can be loaded two at a time (dwordx4)
  uint64_t const* simple_base[MAX_UNITS];
};

// ---- finalize: partial records -> typed output columns
enum out_kind : int8_t {
  OUT_KEY = 0,    // key column `a0` (index into plan cols)
  OUT_ACC,        // accumulator a0 cast to the target type
  OUT_MEAN,       // double(acc a0) / acc a1   (a0 class in `cls`)
  OUT_COUNT,      // accumulator a0 as INT32
  OUT_M2,         // double(a0 = sum of squares) - double(a1 = sum)^2 / (a2 = count); 0 for an empty group
  OUT_VAR,        // M2 / (count - ddof); null when count - ddof <= 0
  OUT_STD,        // sqrt(VAR)
  OUT_MEAN_INT,   // MEAN of a duration / decimal column: (acc a0 truncated to the source width) / (count a1), integer division
  OUT_SUMOV_SUM,  // SUM_OVERFLOW, sum child: the exact sum (a0, or a0 = hi and a2 = lo halves) wrapped to the source width
  OUT_SUMOV_FLAG  // SUM_OVERFLOW, overflow child: the exact sum does not fit the source type
};
struct out_desc {
  void* data;
  bitmask_type* mask;   // nullptr: not nullable
  int32_t* null_count;  // device counter (nullable columns only)
  int8_t kind;
  int8_t a0, a1, a2;
  int8_t ddof;
  int8_t valid_acc;     // accumulator whose value > 0 means valid (-1: always valid)
  int8_t cls;           // accumulator class of a0: elem_class (SINT/UINT/F64)
  int8_t width;         // output element width
  int8_t out_cls;       // elem_class of the output type
  int8_t key_unit, key_hi, key_full;  // OUT_KEY: where the key lives
  int8_t key_acc;       // OUT_KEY of a float column: accumulator (ANY_U64, float64 bits) holding a representative row's value (-1 none)
  int8_t key_null_bit;  // OUT_KEY with INCLUDE: bit in KEYNULLS (-1 none)
  int8_t keynulls_unit, keynulls_hi;
};
constexpr int MAX_OUT = 40;
struct finalize_dev {
  out_desc out[MAX_OUT];
  int32_t nout;
};

// ---- launch geometry chosen by the host
struct part_geom {
  int32_t nseg;        // input segments (1 for level 1)
  int32_t slices;      // work items per segment
  int32_t P;           // partitions per segment (power of two)
  int32_t shift;       // partition digit = (hash >> shift) & (P - 1)
  int32_t tile_rows;   // rows staged in LDS per tile
  int32_t block;       // threads per workgroup
};

struct agg_geom {
  int32_t cap;         // LDS table slots per work item
  int32_t fill_limit;  // groups per table before the item reports overflow
  int32_t block;
};

// Input modes of the aggregate kernel.
enum agg_input : int32_t { IN_COLUMNS = 0, IN_RAW_RECORDS = 1, IN_PARTIAL_RECORDS = 2 };
// How work item i finds its input records.
enum seg_mode : int32_t {
  SEG_ROW_CHUNKS = 0,  // item i = rows [i*chunk, min(n,(i+1)*chunk))          (IN_COLUMNS)
  SEG_OFFSETS    = 1,  // item i = records [off[i], off[i+1])                   (partitioned records)
  SEG_STRIDED    = 2   // item i = `fan` sources s = i*fan..: records [s*stride, s*stride+count[s])  (merge)
};

struct agg_args {
  plan_dev plan;
  agg_geom geom;
  int32_t input;          // agg_input
  int32_t seg;            // seg_mode
  int64_t nrows;          // SEG_ROW_CHUNKS
  int64_t chunk;          // SEG_ROW_CHUNKS
  int64_t const* offsets; // SEG_OFFSETS (nitems + 1)
  int32_t const* src_count;  // SEG_STRIDED
  int64_t src_stride;        // SEG_STRIDED (records)
  int32_t fan;               // SEG_STRIDED
  int32_t nsrc;              // SEG_STRIDED: number of source items
  uint64_t const* records;   // input records (raw or partial)
  uint64_t* out_records;     // partial records out: item i writes [i*cap, i*cap + out_count[i])
  int32_t* out_count;        // groups per item
  int32_t* overflow;         // set to 1 if any item exceeded fill_limit
  int32_t nitems;
};

// Dense integer keys (one plain 8-byte integer key column whose values span a small range): direct addressing instead of
// hashing and probing. index = key - lo in [0, range); scrambled = (index * mult) mod 2^bits is a bijection on `bits`
// bits (mult odd) that spreads runs and strides of keys over the partitions; partition = its top log2P bits, table
// slot = the remaining low bits. A table holds no key words and no state words: slot s of partition d IS the key
// lo + ((d << (bits - log2P) | s) * mult_inv mod 2^bits).
constexpr int DENSE_MAX_KEYS = 4;
// One integer key column of a COMPOSITE dense key: digit = value - lo in [0, range); the row's index is the mixed-radix number
// sum(digit_c * stride_c). unit / half say where the column's bits sit in the key units of a partial record (engine.hpp plan).
struct dense_key {
  uint64_t lo;
  uint32_t range;
  uint32_t stride;
  int8_t col;        // plan column
  int8_t unit;
  int8_t half;       // 0: low half, 1: high half, 2: the whole unit
  int8_t is_signed;  // narrow signed types are sign-extended before lo is subtracted
  int32_t width;     // bytes
};
struct dense_map {
  uint64_t lo;        // single plain 8-byte key (nkeys == 0): index = key - lo
  uint64_t range;     // indices are in [0, range)
  uint32_t mult;
  uint32_t mult_inv;  // mult * mult_inv = 1 mod 2^32
  int32_t bits;       // 2^bits >= range, bits <= 30
  int32_t log2P;      // partition = top log2P bits of scrambled (over all levels), table slot = the low bits - log2P bits
  // Composite keys (nkeys >= 1): 1-4 integer key columns of any width, nullable under null_policy::EXCLUDE (a row with a NULL
  // key is dropped), one value column of any fixed-width type, nullable. The 16-byte record is
  // { index (low 32 bits) | validity of the value (bit 32), value as its 8-byte accumulator class }.
  int32_t nkeys;
  int32_t value_col;  // plan column of the value
  int32_t value_nullable;
  dense_key key[DENSE_MAX_KEYS];
};

struct part_args {
  plan_dev plan;
  part_geom geom;
  int32_t from_columns;        // 1: source = plan columns, 0: source = raw records
  int64_t nrows;               // from_columns
  uint64_t const* in_records;  // !from_columns
  int64_t const* seg_offsets;  // !from_columns: nseg + 1 record offsets
  uint32_t* counts;            // [nseg*slices][P] histogram
  int64_t* item_base;          // [nseg*slices][P] exclusive output offsets (records)
  int64_t* out_offsets;        // [nseg*P + 1] partition boundaries (records)
  uint64_t* out_records;
  // Optimistic single-pass partition (no histogram pass): workgroup w appends partition d's records to its own
  // fixed-capacity region [(d*slices + w) * region_cap, +region_cap); fill counts go to region_count[d*slices + w].
  // If any region would overflow (skewed keys) the workgroup raises *overflow before writing and the host redoes
  // the call with the exact histogram/scan/scatter pipeline.
  int32_t optimistic;
  int64_t region_cap;
  int32_t* region_count;
  int32_t* overflow;
  // Second optimistic level: the input of work item (g, s) is not a contiguous segment but the level-1 regions
  // (g, w), w = s, s + slices, s + 2*slices, ... < in_slices, each [(g*in_slices + w) * in_region_cap, + count).
  // The item sees them as one virtual row range (device_common.hpp region_input); its output regions are those of
  // the GLOBAL partition g*P + d: [((g*P + d) * slices + s) * region_cap, +region_cap).
  int32_t from_regions;
  int32_t const* in_region_count;
  int64_t in_region_cap;
  int32_t in_slices;
  // Write-combining scatter (optimistic, 16-byte records): records per output granule (4 = 64 B, 8 = 128 B); every
  // global store of the tile loop is a whole, aligned granule; 0 = classic run-per-tile scatter.
  int32_t wc_granule;
  // Write-combining scatter of the input columns: workgroup w takes row tiles w, w + slices, ... instead of one contiguous
  // chunk, so that sorted / clustered keys spread over every workgroup's regions as uniform ones do.
  int32_t cyclic_tiles;
  // Heavy hitters (write-combining scatter of plain 16-byte records only): rows whose key is one of hot_keys[0..hot_n)
  // are aggregated (SUM of the value, row COUNT) in an LDS table of HOT_SLOTS entries placed hot_lds_offset bytes into
  // the workgroup's LDS instead of being scattered; workgroup w writes its non-empty entries as partial records to
  // hot_out[w * HOT_SLOTS ...] and their number to hot_count[w].
  int32_t hot_n;
  int32_t hot_lds_offset;
  uint64_t const* hot_keys;
  uint64_t* hot_out;
  int32_t* hot_count;
  // diagnostics (CUDF_AMD_GB_STAMPS=1): per-workgroup cycle totals of the tile phases, 8 x u64 per workgroup
  unsigned long long* stamps;
  // Dense integer keys (write-combining scatter of plain 16-byte records): partition digit from the dense map; a key
  // outside [lo, lo + range) raises bit 2 of *overflow (the sampled range was wrong: the caller redoes the call by hash).
  int32_t use_dense;
  dense_map dense;
};

// Chunked pipeline: the input rows are partitioned one CHUNK at a time into a reused ring of regions
// that stays resident in the 256 MiB Infinity Cache, and each chunk is aggregated before the next one overwrites it.
// A chunk's row range is a by-value kernel parameter (the device-resident part_args stay as they are).
struct chunk_range {
  int64_t begin, end;  // rows [begin, end) of the input columns; end == 0: all rows (no chunking)
};

// Aggregation of dense-key records into direct-address LDS tables (dense_kernels.hip). Work item d = partition d reads the
// regions (d, w), w < slices, of an optimistic scatter. The table IMAGE (the accumulator arrays as they lie in LDS) of
// every partition is carried across the chunks of a call in `tables`; the last chunk dumps partial records
// [key | accumulators] for k_finalize: item d at out_records[d * slots * PU ...], count in out_count[d].
struct dense_agg_args {
  plan_dev plan;
  dense_map map;
  uint64_t const* records;
  // 10-byte records of the ring scatter (dense_ring_args), two streams per region: rec_val[i] = the value, rec_tag[i] = table slot
  // (low bits) | validity of the value (bit 15). rec_tag == nullptr: 16-byte {key or index | validity, value} records in `records`.
  uint64_t const* rec_val;
  uint16_t const* rec_tag;
  // nsplit > 1: a partition's regions are shared out to nsplit workgroups (work item = partition * nsplit + h takes the regions
  // [h * slices / nsplit, (h + 1) * slices / nsplit)); every workgroup leaves its table image in `tables` and
  // launch_dense_merge_dump folds the nsplit images of a partition into the partial records.
  int32_t nsplit;
  int32_t keep_images;  // 1: every workgroup leaves its table image in `tables` even with nsplit == 1 (several value columns:
                        // launch_dense_merge_dump_multi folds the columns' images)
  int32_t const* region_count;
  int64_t region_cap;
  int32_t slices;
  int32_t slots;        // per table: 1 << (bits - log2P)
  int32_t image_bytes;  // multiple of 16
  int32_t occ_acc;      // accumulator that counts every row (SRC_ONE): a slot is occupied iff it is > 0; -1: occupancy bitmap
  int32_t KU;           // key units of a partial record (composite keys: the plan's; single plain key: 1)
  uint64_t* tables;     // [P][image_bytes / 8]
  uint64_t* out_records;
  int32_t* out_count;
  int32_t* overflow;
  int32_t nitems;
  int32_t block;
  // launch_aggregate_dense_columns (one table is all the key range needs): rows of the plan's columns, a dummy validity word
  int64_t nrows;
  uint32_t const* ones;
};
std::size_t dense_table_bytes(plan_dev const& plan, int slots);  // LDS image of one table (multiple of 16)
int dense_occ_acc(plan_dev const& plan);                          // dense_agg_args::occ_acc of a plan
void launch_aggregate_dense(dense_agg_args const& a, dense_agg_args const* d_args, bool first_chunk, bool last_chunk, hipStream_t stream);
void store_args(dense_agg_args const& a, dense_agg_args* d_args, hipStream_t stream);
// nsplit > 1: the nsplit table images of every partition -> partial records. Work item (d, j), j < dsplit, dumps the slots
// [j * slots / dsplit, (j + 1) * slots / dsplit) of partition d to out_records[(d * dsplit + j) * (slots / dsplit) * PU ...], count in
// out_count[d * dsplit + j].
void launch_dense_merge_dump(dense_agg_args const& a, dense_agg_args const* d_args, int dsplit, hipStream_t stream);
// The whole key range fits ONE direct-address table (low cardinality): a.nsplit workgroups aggregate the row tiles w, w + nsplit, ...
// straight from the plan's columns (a.nrows rows), each into a table of its own, and leave the images in a.tables for
// launch_dense_merge_dump (nitems = 1). A key outside the range raises bit 2 of *a.overflow. Replaces, for such keys, the
// reference's shared-memory aggregation path (compute_shared_memory_aggs.cu:260-353) and the hash tables of aggregate_kernels.hip.
void launch_aggregate_dense_columns(dense_agg_args const& a, dense_agg_args const* d_args, hipStream_t stream);
// ... and the fold of those images: work item b dumps the slots [64 b, 64 b + 64) to out_records[b * 64 * PU ...], count in out_count[b]
void launch_dense_merge_dump_wide(dense_agg_args const& a, dense_agg_args const* d_args, hipStream_t stream);

// Ring scatter of dense-key rows (dense_ring_kernels.hip): every partition owns a ring of record slots in LDS, a row reserves its
// position with one returning LDS atomic, and only whole aligned 128-byte granules (16 values, 64 or 32 tags) leave the
// workgroup - two barriers per tile. Records are two streams per region: value (8 B) and tag (the bits of the scrambled index
// below this level's digit | validity of the value in the top bit: 2 B on the last level, 4 B on the first of two). Level 1 reads the input columns (workgroup w
// takes the row tiles w, w + slices, ...), level 2 reads level-1 partition g as the strided list of its regions
// s, s + slices, ... (work item g * slices + s) and writes the regions of the global partitions g * P + d.
// Region (q, w) of an output with S slices lies at [(q * S + w) * region_cap, + region_count[q * S + w]) of both streams.
struct dense_ring_args {
  plan_dev plan;
  dense_map map;
  int32_t from_columns;
  int64_t nrows;
  int32_t P;       // fan-out of this level: a power of two, 16 ... 256
  int32_t capl;    // log2 of the ring capacity per partition: P << capl = 8192 records
  int32_t shift;   // digit = (x >> shift) & (P - 1), tag = x & ((1 << shift) - 1): x = scrambled index (level 1), input tag (level 2)
  int32_t slices;
  int32_t nseg;    // level 2: level-1 partitions
  uint64_t const* in_val;
  uint32_t const* in_tag;
  int32_t const* in_region_count;
  int64_t in_region_cap;
  int32_t in_slices;
  uint64_t* out_val;
  void* out_tag;       // uint16_t (tag16) or uint32_t per record
  int32_t tag16;       // the last level: 16-bit tags = table slot | validity << 15; else 32-bit: low index bits | validity << 31
  int64_t region_cap;  // a multiple of 64 records
  int32_t* region_count;
  int32_t* overflow;   // bit 0: a region overflowed; bit 2: a key outside the dense range
  uint32_t const* ones;  // composite keys: one all-ones word, read in place of the validity word of a column without a mask
  int32_t static_shapes; // composite keys: take the loader compiled for the columns' shape where there is one (CUDF_AMD_GB_STATIC_SHAPES)
  // Heavy hitters (one plain key, SUM / COUNT plans: hot_plan_ok): rows whose key is one of hot_keys[0 .. hot_n) are aggregated
  // (SUM of the value, row COUNT) in an LDS table of HOT_SLOTS entries behind the rings and never scattered; workgroup w writes
  // its non-empty entries as partial records [key | accumulators in plan order] to hot_out[w * HOT_SLOTS ...], their number to
  // hot_count[w] (the same hand-over as part_args::hot_*).
  int32_t hot_n;
  uint64_t const* hot_keys;
  uint64_t* hot_out;
  int32_t* hot_count;
};
constexpr int DENSE_RING_SLOTS = 8192;  // value-ring slots of a workgroup (all partitions); the tag rings are twice as long: 128 KiB of LDS
void store_args(dense_ring_args const& a, dense_ring_args* d_args, hipStream_t stream);
void launch_dense_ring_scatter(dense_ring_args const& a, dense_ring_args const* d_args, hipStream_t stream);
// ---- several value columns on the dense path (dense_multi_kernels.hip): one plain 8-byte integer key column and 2-3 plain 8-byte
// value columns. The ring scatter writes one VALUE STREAM PER COLUMN next to the shared 16-bit tag stream (a row costs
// 2 + 8 * nval bytes in the record buffer); region (d, w) lies at [(d * slices + w) * region_cap, + region_count) of every stream,
// value stream j at out_val + j * stream_stride. The aggregate then runs once per COLUMN over (tags, value stream j) with that
// column's accumulators only (a sub-plan: its table image is slots x 8-12 bytes, as for one column) and
// launch_dense_merge_dump_multi folds the columns' images into partial records [key | accumulators of the whole plan].
// The reference aggregates every (column, aggregation) pair of a call in one pass over the rows
// (cpp/src/groupby/hash/compute_global_memory_aggs.cuh:139-147); this is the same single pass over the INPUT.
constexpr int RING_MAX_VALUES = 3;
struct ring_multi_args {
  plan_dev plan;         // simple plan: simple_base[0] = the key column, simple_base[1 + j] = value column j
  dense_map map;
  int64_t nrows;
  int32_t nval;          // value streams: 2 or 3
  int32_t P;             // partitions: a power of two, 16 ... 256 (16 waves x P / 16 owner lanes)
  int32_t cap;           // records per partition ring: 48 (nval = 2) or 32 (nval = 3); the tag rings hold 128
  int32_t shift;         // digit = scrambled index >> shift, tag = its low `shift` bits | 1 << 15
  int32_t slices;
  uint64_t* out_val;
  int64_t stream_stride; // records between the value streams
  uint16_t* out_tag;
  int64_t region_cap;    // a multiple of 64 records
  int32_t* region_count;
  int32_t* overflow;     // bit 0: a region overflowed; bit 2: a key outside the dense range
};
void store_args(ring_multi_args const& a, ring_multi_args* d_args, hipStream_t stream);
void launch_dense_ring_scatter_multi(ring_multi_args const& a, ring_multi_args const* d_args, hipStream_t stream);
std::size_t dense_ring_multi_lds_bytes(int nval, int P, int cap);
int dense_ring_multi_cap(int nval);  // ring capacity per partition for nval value streams at 128 partitions

struct dense_multi_merge_args {
  plan_dev plan;         // the WHOLE plan (its NACC accumulators in order)
  dense_map map;
  int32_t ncols;         // value columns
  int32_t nsplit;        // images per (partition, column)
  int32_t slots;
  int32_t occ_acc;       // accumulator of the whole plan that counts every row (SRC_ONE), or -1: occupancy bitmap of column 0's images
  uint32_t occ_off0;     // byte offset of that bitmap in an image of column 0
  uint64_t const* tables[RING_MAX_VALUES];  // column c: [partition][nsplit] images of image_bytes[c]
  int32_t image_bytes[RING_MAX_VALUES];
  int8_t acc_col[MAX_ACC];     // accumulator q lives in the images of column acc_col[q] ...
  int8_t acc_narrow[MAX_ACC];  // ... as a 4-byte count (1) or an 8-byte value (0) ...
  uint32_t acc_off[MAX_ACC];   // ... at this byte offset
  uint64_t* out_records;       // work item (d, j): [(d * dsplit + j) * (slots / dsplit) * (1 + NACC) ...]
  int32_t* out_count;
  int32_t nitems;              // partitions
};
void store_args(dense_multi_merge_args const& a, dense_multi_merge_args* d_args, hipStream_t stream);
void launch_dense_merge_dump_multi(dense_multi_merge_args const& a, dense_multi_merge_args const* d_args, int dsplit, hipStream_t stream);
// byte offset of accumulator q / of the occupancy bitmap in the table image of `plan` (dense_kernels.hip make_dense_layout)
uint32_t dense_acc_offset(plan_dev const& plan, int slots, int q);
uint32_t dense_occ_offset(plan_dev const& plan, int slots);

// Minimum and maximum of a plain 8-byte integer key column over the strided sample of launch_estimate (signed compare
// for signed keys): out[0] = min, out[1] = max as bit patterns. `out` must hold {max value, min value} of the ordering before.
void launch_key_range(plan_dev const* d_plan, int64_t nrows, int64_t sample, int is_signed, uint64_t* out, hipStream_t stream);
// The same for every key column of a plan (any integer width, nullable): out[2 c] = min, out[2 c + 1] = max of the VALID sampled
// values of key column c, as int64 (signed columns sign-extended, unsigned zero-extended).
void launch_key_ranges(plan_dev const* d_plan, int nkeycols, int64_t nrows, int64_t sample, int64_t* out, hipStream_t stream);

// Launchers (partition_kernels.hip, aggregate_kernels.hip). All asynchronous on `stream`. Kernel arguments live in DEVICE memory (`d_args`,
// one slot per launch family, written by a one-thread kernel on the same stream): passed by value, the
// dynamically indexed column descriptors were copied to scratch by the compiler (792 B/lane) and every
// descriptor access became a vector memory load.
void store_args(part_args const& a, part_args* d_args, hipStream_t stream);
void launch_partition_hist(part_args const& a, part_args const* d_args, hipStream_t stream);
void launch_partition_scan(part_args const& a, part_args const* d_args, hipStream_t stream);
void launch_partition_scatter(part_args const& a, part_args const* d_args, hipStream_t stream, chunk_range chunk = {0, 0});
void launch_aggregate(agg_args const& a, agg_args* d_args, hipStream_t stream);
// Gathers `total` groups (items' partial records, prefix[] = exclusive scan of the item counts) into the
// typed output columns.
struct finalize_args {
  plan_dev plan;
  finalize_dev fin;
};
void launch_finalize(finalize_args const& a, finalize_args* d_args, uint64_t const* records, int64_t cap,
                     int64_t const* prefix, int32_t nitems, int64_t total, hipStream_t stream);
// prefix[i] = counts[0] + ... + counts[i - 1] for i <= nitems (the items' group counts -> where each item's groups go in the output)
// ---- path A front end for plain shapes (collapse_runs.hip): runs of equal keys -> one partial record each, no hash table.
// Chunk `item` = rows [item * chunk, + chunk); its records [key | accumulators] go to out_records[item * out_stride * PU ...),
// their number to out_count[item]; a chunk with more than out_stride records raises bit 1 of *overflow.
struct collapse_args {
  plan_dev plan;  // simple, one key unit, <= 2 payload units, no NULLs (collapse_runs_applies)
  int64_t nrows;
  int64_t chunk;
  int32_t nitems;
  int64_t out_stride;
  uint64_t* out_records;
  int32_t* out_count;
  int32_t* overflow;
};
bool collapse_runs_applies(plan_dev const& p);
void launch_collapse_runs(collapse_args const& a, collapse_args* d_args, hipStream_t stream);

void launch_count_prefix(int32_t const* counts, int32_t nitems, int64_t* prefix, hipStream_t stream);
// item i's records [i * cap, i * cap + prefix[i + 1] - prefix[i]) of `units` 8-byte units each -> out[prefix[i] ...] (contiguous)
void launch_compact_records(uint64_t const* records, int64_t cap, int64_t const* prefix, int32_t nitems, int units, uint64_t* out, hipStream_t stream);
void launch_store_i64x2(int64_t a, int64_t b, int64_t* dst, hipStream_t stream);  // dst[0] = a, dst[1] = b, stream-ordered
// Distinct-count estimate on a strided sample (linear counting into a bitmap); result written to *d_bits.
// range_mode 1 / 2 (one plain 8-byte integer key column, signed / unsigned): the pass also leaves the minimum and maximum of the
// sampled keys in range_out[0], [1]; blk_range holds 2 x ceil(sample / 256) scratch words.
void launch_estimate(plan_dev const& plan, plan_dev* d_plan, int64_t nrows, int64_t sample, uint32_t* bitmap,
                     int32_t bitmap_bits_log2, uint32_t* d_bits_set, uint32_t* hot_buckets, hipStream_t stream, int range_mode = 0,
                     uint64_t* blk_range = nullptr, uint64_t* range_out = nullptr, uint32_t* blk_adj = nullptr, uint32_t* adj_out = nullptr);
// (blk_adj: 2 x ceil(sample / 256) scratch words; adj_out[0] = sampled rows whose successor row was looked at, adj_out[1] = those
// whose successor carries the same key - sorted and clustered inputs show up here)

// Distinct-count of the key rows over ALL rows (HyperLogLog, HLL_REGISTERS 32-bit registers holding ranks): the planner
// runs it after a table overflowed, i.e. when the sample misjudged the group count (skewed key frequencies).
constexpr int HLL_REGISTERS = 1 << 14;
void launch_distinct_count(plan_dev const& plan, plan_dev* d_plan, int64_t nrows, uint32_t* regs, hipStream_t stream);

// Heavy hitters of a plain 8-byte key column, from the same strided sample as the estimate: sample counts per hash bucket
// (HOT_BUCKETS counters, filled by launch_estimate's pass over the sample), then exact sample counts of the keys of the buckets with at least `min_count` rows, in an
// open-addressing table of HOT_TABLE (key, count) entries (empty key = all ones) that the host reads back and sorts.
constexpr int HOT_BUCKETS = 65536, HOT_TABLE = 4096, HOT_MAX_KEYS = 256, HOT_SLOTS = 512;
// (d_bits_set: the distinct keys of the sample as launch_estimate left them; a key counts as frequent from
// hot_keys_threshold(min_count, sample, bits_set) = max(min_count, 8x the mean count of a key of the sample) rows)
void launch_hot_keys(plan_dev const* d_plan, int64_t nrows, int64_t sample, uint32_t min_count, uint32_t* buckets, uint32_t const* d_bits_set,
                     uint64_t* table_keys, uint32_t* table_counts, hipStream_t stream);
uint32_t hot_keys_threshold(uint32_t min_count, int64_t sample, uint32_t bits_set);

std::size_t aggregate_lds_bytes(plan_dev const& plan, agg_geom const& g);
// LDS bytes of one table slot: key units, accumulators (4 bytes for counts), state word
int aggregate_slot_bytes(plan_dev const& plan);
std::size_t partition_lds_bytes(plan_dev const& plan, part_geom const& g);
// write-combining scatter: can records of U units be partitioned P ways with granules of G records?
// (block = 512: two workgroups per CU, each within 80 KiB)
bool partition_wc_fits(int U, int P, int G, int block = 1024);
// LDS bytes of the heavy-hitter table; byte offset of it behind the write-combining scatter's own LDS
std::size_t partition_hot_lds_bytes();

}  // namespace cudf::groupby::detail
