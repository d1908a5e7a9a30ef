/* TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the reference's hash-groupby / hash-join /
 * hash-partition semantics. Nothing under cudf_amd/ may include, link or call this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it (as the checker, never as the product).
 *
 * Parity status: PINNED at the cudf:: boundary by the reference's own known-answer tests, transcribed as data
 * into tests/golden/kat_groupby.json and tests/golden/kat_join.json (sources listed there), and by the public
 * MurmurHash3_x86_32 verification vectors. The reference itself (CUDA + cuco + rmm) cannot be built or
 * imported in this image; no reference binary exists under oracle/_ref.
 */
#ifndef CUDF_AMD_ORACLE_H
#define CUDF_AMD_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Arrow-layout host column: same fields as cudf::column_view (reference column_view.hpp:236-244). */
typedef struct {
  int32_t type_id;      /* cudf::type_id value */
  int32_t size;         /* rows */
  const void* data;     /* head pointer (element i at data[offset+i]) */
  const uint32_t* mask; /* validity bits, LSB-first, bit (offset+i); NULL = all valid */
  int32_t null_count;
  int32_t offset;
} orc_column;

typedef struct {
  orc_column values;
  const int32_t* kinds; /* cudf::aggregation::Kind values */
  int32_t nkinds;
} orc_request;

/* Owned result column (malloc'ed). */
typedef struct {
  int32_t type_id;
  int32_t size;
  void* data;
  uint32_t* mask; /* NULL if the column is not nullable */
  int32_t null_count;
  uint8_t* aux;   /* SUM_OVERFLOW: the overflow child (one bool per group); `data` is the sum child, `mask` the struct's; else NULL */
} orc_out_column;

typedef struct {
  int32_t nkeys;
  orc_out_column* keys; /* one per key column, G rows, first-appearance order */
  int32_t nresults;     /* total result columns = sum over requests of nkinds */
  orc_out_column* results;
} orc_groupby_result;

enum { ORC_OK = 0, ORC_LOGIC_ERROR = 1, ORC_INVALID_ARGUMENT = 2, ORC_DATA_TYPE_ERROR = 3, ORC_NOT_IMPLEMENTED = 4 };

const char* orc_last_error(void);

/* cudf::groupby::groupby(keys, null_handling).aggregate(requests) — hash path semantics,
 * SURVEY.md Appendix A rules 1-12. include_null_keys = (null_policy::INCLUDE). */
int orc_groupby(const orc_column* keys, int32_t nkeys, int32_t include_null_keys, const orc_request* requests,
                int32_t nrequests, orc_groupby_result** out);
void orc_groupby_free(orc_groupby_result* r);

/* cudf::inner_join / left_join / full_join(left_keys, right_keys, compare_nulls) — Appendix A rules 13-18.
 * kind: 0 inner, 1 left, 2 full. nulls_equal = (null_equality::EQUAL). Pairs are emitted in (left row, then
 * right row) ascending order; out_left and out_right are malloc'ed int32 arrays of out_n entries
 * (JoinNoMatch = INT32_MIN for the unmatched side). */
int orc_join(const orc_column* left, int32_t nleft, const orc_column* right, int32_t nright, int32_t nulls_equal,
             int32_t kind, int32_t** out_left, int32_t** out_right, int64_t* out_n);
/* Size only (may exceed INT32_MAX; reference join_tests.cpp:2379-2394). */
int orc_join_size(const orc_column* left, int32_t nleft, const orc_column* right, int32_t nright,
                  int32_t nulls_equal, int32_t kind, uint64_t* out_n);
void orc_free(void* p);

/* Row hash: MurmurHash3_x86_32 per element, first column is the init, others folded with hash_combine,
 * null element -> UINT32_MAX, floats normalised (-0 -> +0, NaN -> canonical) (reference
 * detail/row_operator/hashing.cuh:41-134, hashing/detail/murmurhash3_x86_32.cuh:21-67,
 * hashing/detail/hashing.hpp:83-86, hash_functions.cuh:19-37). */
int orc_row_hash(const orc_column* cols, int32_t ncols, uint32_t seed, uint32_t* out);
uint32_t orc_murmur3_32(const void* bytes, uint64_t len, uint32_t seed);

/* cudf::hash_partition partition map: out_part[i] = row_hash(i) % num_partitions; out_offsets[p] = start of
 * partition p in the (stable) reordered table; out_order[j] = source row of output row j
 * (reference partitioning.cu:54-92,569-760). */
int orc_hash_partition(const orc_column* cols, int32_t ncols, int32_t num_partitions, uint32_t seed,
                       int32_t* out_part, int32_t* out_offsets, int32_t* out_order);

#ifdef __cplusplus
}
#endif
#endif
