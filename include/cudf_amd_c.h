/* SPDX-License-Identifier: Apache-2.0
 * Flat C ABI of libcudf_amd.so — plain pointers and sizes, no C++ / torch types.
 *
 * The reference has no C ABI: its boundary is the C++ API of libcudf.so that pylibcudf (Cython) and the JNI
 * layer bind directly. This header is the equivalent a ctypes / cgo / JNI / N-API binding
 * would use; every entry point names the reference interface it stands for. The same library also exports
 * the source-compatible C++ API under include/cudf/ (cudf::groupby::groupby, cudf::inner_join, ...), which is
 * what a Cython .pxd would declare.
 *
 * Conventions: all device pointers are HIP device pointers valid on the current device; `stream` is a
 * hipStream_t passed as void* (NULL = default stream); every function returns a cudf_amd_status and, on
 * failure, leaves the message in cudf_amd_last_error() (thread-local). Inputs are non-owning views, outputs
 * are owning handles released with cudf_amd_table_free (ownership rule of reference groupby.hpp:106-108).
 */
#ifndef CUDF_AMD_C_H
#define CUDF_AMD_C_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Maps the C++ exception types of the reference's error convention (cpp/include/cudf/utilities/error.hpp:35,
 * 63,86,97; pylibcudf exception_handler.pxd:34-64). */
typedef enum {
  CUDF_AMD_OK               = 0,
  CUDF_AMD_LOGIC_ERROR      = 1, /* cudf::logic_error        -> RuntimeError    */
  CUDF_AMD_INVALID_ARGUMENT = 2, /* std::invalid_argument    -> ValueError      */
  CUDF_AMD_DATA_TYPE_ERROR  = 3, /* cudf::data_type_error    -> TypeError       */
  CUDF_AMD_DEVICE_ERROR     = 4, /* cudf::cuda_error (HIP)   -> RuntimeError    */
  CUDF_AMD_BAD_ALLOC        = 5, /* std::bad_alloc           -> MemoryError     */
  CUDF_AMD_OTHER_ERROR      = 6,
  CUDF_AMD_OUT_OF_RANGE     = 7  /* std::out_of_range        -> IndexError      */
} cudf_amd_status;

/* cudf::column_view (reference cpp/include/cudf/column/column_view.hpp:236-244): element i at
 * data[(offset+i)], validity bit (offset+i) of null_mask (LSB-first, 1 = valid); null_mask NULL = all valid. */
typedef struct {
  int32_t type_id; /* cudf::type_id (reference types.hpp:185-217) */
  int32_t size;
  const void* data;
  const uint32_t* null_mask;
  int32_t null_count;
  int32_t offset;
  int32_t scale; /* fixed-point scale, 0 otherwise */
} cudf_amd_column_view;

/* cudf::groupby::aggregation_request (reference groupby.hpp:54-57); kinds = cudf::aggregation::Kind values
 * (aggregation.hpp:78-121). */
typedef struct {
  cudf_amd_column_view values;
  const int32_t* kinds;
  int32_t num_kinds;
  /* one parameter per aggregation, or NULL for the defaults: ddof of VARIANCE / STD (make_variance_aggregation(ddof),
   * aggregation.hpp:259-266; default 1), n of NTH_ELEMENT; ignored by the other kinds */
  const int32_t* params;
  /* a second parameter per aggregation, or NULL for the defaults: the null_policy of NTH_ELEMENT (default 1 = INCLUDE,
   * aggregation.hpp:366-367) and of NUNIQUE (default 0 = EXCLUDE, :349), the cudf::interpolation of QUANTILE (default 0 = LINEAR,
   * :320-321; types.hpp:173-180) */
  const int32_t* params2;
  /* QUANTILE: aggregation k asks for quantiles[quantile_offsets[k] .. quantile_offsets[k + 1]) (num_kinds + 1 offsets); both NULL
   * when the request holds no QUANTILE. Its result column holds groups x quantiles values, group-major (group_quantiles.cu:84-87) */
  const double* quantiles;
  const int32_t* quantile_offsets;
} cudf_amd_aggregation_request;

/* Owning cudf::table / vector of cudf::column. */
typedef struct cudf_amd_table_s* cudf_amd_table_t;
/* Owning cudf::hash_join. */
typedef struct cudf_amd_hash_join_s* cudf_amd_hash_join_t;

const char* cudf_amd_last_error(void);
const char* cudf_amd_version(void);
/* ABI number of this header: bumped whenever a struct layout or the meaning of an argument changes, so that a consumer built
 * against an older header can refuse to run. 2: cudf_amd_aggregation_request gained `params`; cudf_amd_hash_partition writes
 * num_partitions + 1 offsets (the reference's current contract). 3: loopback communicators, message limit, shuffle_join.
 * 4: cudf_amd_aggregation_request gained `params2`, `quantiles`, `quantile_offsets` (the sort-groupby kinds). */
#define CUDF_AMD_ABI_VERSION 4
int32_t cudf_amd_abi_version(void);

/* ---- device memory / stream plumbing for bindings without a device allocator of their own */
cudf_amd_status cudf_amd_malloc(void** ptr, size_t bytes, void* stream);
cudf_amd_status cudf_amd_free(void* ptr, void* stream);
/* kind: 0 host->device, 1 device->host, 2 device->device; asynchronous on `stream`. */
cudf_amd_status cudf_amd_memcpy(void* dst, const void* src, size_t bytes, int32_t kind, void* stream);
cudf_amd_status cudf_amd_memset(void* dst, int32_t value, size_t bytes, void* stream);
cudf_amd_status cudf_amd_stream_synchronize(void* stream);
/* Live / peak bytes handed out by the library's current device resource since load. */
cudf_amd_status cudf_amd_memory_stats(uint64_t* current_bytes, uint64_t* peak_bytes);

/* ---- per-kernel timing (HIP events on the launch stream); off by default.
 * report: one line per kernel "name launches total_ms\n" written into buf (NUL-terminated, truncated to n). */
cudf_amd_status cudf_amd_profile_enable(int32_t on);
cudf_amd_status cudf_amd_profile_reset(void);
cudf_amd_status cudf_amd_profile_report(char* buf, size_t n);

/* ---- owning tables */
int32_t cudf_amd_table_num_columns(cudf_amd_table_t t);
int32_t cudf_amd_table_num_rows(cudf_amd_table_t t);
cudf_amd_status cudf_amd_table_column(cudf_amd_table_t t, int32_t i, cudf_amd_column_view* out);
/* children of a STRUCT column (SUM_OVERFLOW: {sum, overflow}; reference column_view.hpp:469 child()): their number, and
 * child j as a view (valid while the table lives) */
int32_t cudf_amd_table_column_num_children(cudf_amd_table_t t, int32_t i);
cudf_amd_status cudf_amd_table_column_child(cudf_amd_table_t t, int32_t i, int32_t j, cudf_amd_column_view* out);
void cudf_amd_table_free(cudf_amd_table_t t);

/* ---- cudf::groupby::groupby(keys, null_handling, keys_are_sorted).aggregate(requests, stream, mr)
 * (reference cpp/include/cudf/groupby.hpp:121-125,181-184; src/groupby/groupby.cu:219-236).
 * include_null_keys: 0 = null_policy::EXCLUDE, 1 = INCLUDE. out_results holds the result columns of all requests
 * flattened in request order (results[i].results[j]). out_path (optional) reports which kernel family ran
 * (cudf::groupby::hash_path). */
cudf_amd_status cudf_amd_groupby_aggregate(const cudf_amd_column_view* keys, int32_t num_keys,
                                           int32_t include_null_keys, int32_t keys_are_sorted,
                                           const cudf_amd_aggregation_request* requests, int32_t num_requests,
                                           void* stream, cudf_amd_table_t* out_keys, cudf_amd_table_t* out_results,
                                           int32_t* out_path);

/* ---- cudf::inner_join / left_join / full_join(left_keys, right_keys, compare_nulls, stream, mr)
 * (reference cpp/include/cudf/join/join.hpp:160-166 and the left/full overloads; src/join/join.cu:30-118).
 * nulls_equal: 1 = null_equality::EQUAL, 0 = UNEQUAL. kind: 0 inner, 1 left, 2 full.
 * out_indices: table of two INT32 columns {left_indices, right_indices}; unmatched side = JoinNoMatch (INT32_MIN).
 * pylibcudf wraps the two device_uvectors into columns the same way (join.pyx:51-64). */
cudf_amd_status cudf_amd_join(const cudf_amd_column_view* left_keys, int32_t num_left,
                              const cudf_amd_column_view* right_keys, int32_t num_right, int32_t nulls_equal,
                              int32_t kind, void* stream, cudf_amd_table_t* out_indices);

/* ---- cudf::hash_join (reference cpp/include/cudf/join/hash_join.hpp:71): build once on `right`, probe many times.
 * has_nulls: 1 = nullable_join::YES, 0 = NO, -1 = use the (right, compare_nulls) constructor. load_factor in (0, 1].
 * The caller keeps `right`'s memory alive while the object lives (as the reference requires). */
cudf_amd_status cudf_amd_hash_join_create(const cudf_amd_column_view* right_keys, int32_t num_right, int32_t has_nulls,
                                          int32_t nulls_equal, double load_factor, void* stream,
                                          cudf_amd_hash_join_t* out);
void cudf_amd_hash_join_destroy(cudf_amd_hash_join_t h);
/* output_size < 0: unknown (a count pass runs first). */
cudf_amd_status cudf_amd_hash_join_probe(cudf_amd_hash_join_t h, const cudf_amd_column_view* left_keys, int32_t num_left,
                                         int32_t kind, int64_t output_size, void* stream, cudf_amd_table_t* out_indices);
/* inner_join_size / left_join_size / full_join_size: std::size_t, may exceed INT32_MAX. */
cudf_amd_status cudf_amd_hash_join_size(cudf_amd_hash_join_t h, const cudf_amd_column_view* left_keys, int32_t num_left,
                                        int32_t kind, void* stream, uint64_t* out_size);

/* hash_join::{inner,left,full}_join_match_context (reference hash_join.hpp:276-329): out_counts = table of ONE INT32
 * column, the number of pairs each left row contributes (left/full kinds count a row without a match as 1). */
cudf_amd_status cudf_amd_hash_join_match_counts(cudf_amd_hash_join_t h, const cudf_amd_column_view* left_keys,
                                                int32_t num_left, int32_t kind, void* stream, cudf_amd_table_t* out_counts);
/* hash_join::partitioned_{inner,left,full}_join (reference hash_join.hpp:353-412): the join of rows
 * [left_start, left_end) of the left table whose match counts are `match_counts` (device pointer to num_rows INT32,
 * from cudf_amd_hash_join_match_counts); left indices refer to the complete left table. kind 2 emits the probe side
 * only: finish with cudf_amd_hash_join_finalize_full. */
cudf_amd_status cudf_amd_hash_join_probe_range(cudf_amd_hash_join_t h, const cudf_amd_column_view* left_keys,
                                               int32_t num_left, const int32_t* match_counts, int32_t kind,
                                               int32_t left_start, int32_t left_end, void* stream,
                                               cudf_amd_table_t* out_indices);
/* hash_join::finalize_partitioned_full_join (reference hash_join.hpp:414-441): concatenates the partial results
 * (device pointers, sizes in elements) and appends (JoinNoMatch, r) for every right row no partial matched. */
cudf_amd_status cudf_amd_hash_join_finalize_full(const int32_t* const* left_partials, const int32_t* const* right_partials,
                                                 const uint64_t* partial_sizes, int32_t num_partials,
                                                 int32_t left_num_rows, int32_t right_num_rows, void* stream,
                                                 cudf_amd_table_t* out_indices);

/* ---- Arrow C Data Interface (reference cpp/include/cudf/interop.hpp: from_arrow :685-689, to_arrow_host :618-621,
 * to_arrow_schema :473-475). The structs are the ones the Arrow specification publishes.
 * cudf_amd_from_arrow: host Arrow struct array (one child per column) -> owning device table; the input is not
 * released. cudf_amd_to_arrow_host: device columns -> host Arrow data MOVED into the caller's *out_schema / *out_array
 * (the caller, or whoever imports them, calls their release callbacks). */
struct ArrowSchema;
struct ArrowArray;
cudf_amd_status cudf_amd_from_arrow(const struct ArrowSchema* schema, const struct ArrowArray* array, void* stream,
                                    cudf_amd_table_t* out_table);
cudf_amd_status cudf_amd_to_arrow_host(const cudf_amd_column_view* columns, int32_t num_columns, const char* const* names,
                                       void* stream, struct ArrowSchema* out_schema, struct ArrowArray* out_array);

/* ---- cudf::hash_partition(input, columns_to_hash, num_partitions, HASH_MURMUR3, seed, stream, mr)
 * (reference cpp/include/cudf/partitioning.hpp:84-101; src/partitioning/partitioning.cu:925-947). out_offsets receives
 * num_partitions + 1 row offsets (partition i = rows [offsets[i], offsets[i+1]); the last one is the row count), as the
 * reference's std::vector<size_type>. A column index outside the table returns CUDF_AMD_OUT_OF_RANGE. */
cudf_amd_status cudf_amd_hash_partition(const cudf_amd_column_view* input, int32_t num_columns,
                                        const int32_t* columns_to_hash, int32_t num_hash_columns, int32_t num_partitions,
                                        uint32_t seed, void* stream, cudf_amd_table_t* out_table, int32_t* out_offsets);

/* ---- multi-GPU exchange (include/cudf/distributed.hpp; the reference's data flow: cpp/libcudf_streaming/src/
 * partition_utils.cpp:72-185). One process per GPU; RCCL over xGMI, resolved at run time.
 * cudf_amd_comm_unique_id: rank 0 fills 128 bytes (ncclUniqueId) and hands them to the other ranks over its control plane.
 * cudf_amd_comm_create: collective over all ranks of the new communicator.
 * cudf_amd_range_partition: rows reordered by destination = (murmur3 row hash of the key columns * num_destinations) >> 32;
 *   out_offsets receives num_destinations + 1 row offsets.
 * cudf_amd_shuffle: collective; every rank receives the rows it owns (from rank 0, then rank 1, ...).
 * cudf_amd_shuffle_groupby: BASELINE config 5 = cudf_amd_shuffle of (keys, values) + the local hash groupby; the ranks'
 *   results are disjoint, their union is the global result.
 * cudf_amd_combine_groupby: the decomposable form (cudf::distributed::combine_groupby; reference
 *   cpp/src/groupby/streaming_groupby/merge.cu:91-144): local groupby with partial aggregations -> cudf_amd_shuffle of the partial
 *   GROUPS -> merge groupby on the owner -> finalisation (counts to INT32, MEAN = merged sum / merged count). Same arguments and
 *   results as cudf_amd_shuffle_groupby; SUM / PRODUCT / SUM_OF_SQUARES / MIN / MAX / COUNT_* / MEAN of numeric columns, anything
 *   else: CUDF_AMD_INVALID_ARGUMENT before the exchange. */
typedef struct cudf_amd_comm_s* cudf_amd_comm_t;
cudf_amd_status cudf_amd_comm_unique_id(uint8_t* out_id_128_bytes);
cudf_amd_status cudf_amd_comm_create(const uint8_t* id_128_bytes, int32_t world_size, int32_t rank, cudf_amd_comm_t* out);
void cudf_amd_comm_destroy(cudf_amd_comm_t comm);
/* Loopback world (include/cudf/distributed.hpp `transport`): world_size virtual ranks inside this process on the current device;
 * out_comms receives world_size communicators. Every rank must be driven by its own host thread (collectives rendezvous). */
cudf_amd_status cudf_amd_comm_create_loopback(int32_t world_size, cudf_amd_comm_t* out_comms);
/* Largest single message of the payload exchange in bytes (default 1 GiB); larger slices travel in several rounds. Every rank
 * of a communicator must set the same value. */
cudf_amd_status cudf_amd_comm_set_max_message_bytes(cudf_amd_comm_t comm, int64_t bytes);
/* Host-side bookkeeping of the exchange (no device work; cudf::distributed::plan_exchange): counts[p * world + q] = rows rank p
 * sends to rank q -> what `rank` receives from each peer (world entries), where the slices start in its receive buffers
 * (world + 1 offsets), the largest message between two different ranks in rows. */
cudf_amd_status cudf_amd_plan_exchange(const int64_t* counts, int32_t world_size, int32_t rank, int64_t* out_recv_count,
                                       int64_t* out_recv_offset, int64_t* out_biggest);
/* cudf::distributed::shuffle_join: inner join of two row-sharded tables; both sides shuffled by key, the owner rank joins.
 * out_global_row_ids: a table of two INT64 columns (global left row id, global right row id), global id = rows of the lower
 * ranks + local index. Collective. */
cudf_amd_status cudf_amd_shuffle_join(cudf_amd_comm_t comm, const cudf_amd_column_view* left_keys, int32_t num_left,
                                      const cudf_amd_column_view* right_keys, int32_t num_right, int32_t nulls_equal, void* stream,
                                      cudf_amd_table_t* out_global_row_ids);
cudf_amd_status cudf_amd_range_partition(const cudf_amd_column_view* input, int32_t num_columns, const int32_t* key_columns,
                                         int32_t num_key_columns, int32_t num_destinations, void* stream,
                                         cudf_amd_table_t* out_table, int32_t* out_offsets);
cudf_amd_status cudf_amd_shuffle(cudf_amd_comm_t comm, const cudf_amd_column_view* input, int32_t num_columns,
                                 const int32_t* key_columns, int32_t num_key_columns, void* stream, cudf_amd_table_t* out_table);
cudf_amd_status cudf_amd_shuffle_groupby(cudf_amd_comm_t comm, const cudf_amd_column_view* keys, int32_t num_keys,
                                         int32_t include_null_keys, const cudf_amd_aggregation_request* requests,
                                         int32_t num_requests, void* stream, cudf_amd_table_t* out_keys,
                                         cudf_amd_table_t* out_results);
cudf_amd_status cudf_amd_combine_groupby(cudf_amd_comm_t comm, const cudf_amd_column_view* keys, int32_t num_keys,
                                         int32_t include_null_keys, const cudf_amd_aggregation_request* requests,
                                         int32_t num_requests, void* stream, cudf_amd_table_t* out_keys,
                                         cudf_amd_table_t* out_results);

/* ---- cudf::hashing::murmurhash3_x86_32(input, seed) -> UINT32 column (reference cpp/include/cudf/hashing.hpp). */
cudf_amd_status cudf_amd_murmurhash3_x86_32(const cudf_amd_column_view* input, int32_t num_columns, uint32_t seed,
                                            void* stream, cudf_amd_table_t* out_column);

/* ---- cudf::gather(source_table, gather_map, bounds_policy) (reference cpp/include/cudf/copying.hpp).
 * nullify: 1 = out_of_bounds_policy::NULLIFY (JoinNoMatch and other out-of-range indices give NULL rows). */
cudf_amd_status cudf_amd_gather(const cudf_amd_column_view* source, int32_t num_columns,
                                const cudf_amd_column_view* gather_map, int32_t nullify, void* stream,
                                cudf_amd_table_t* out_table);

#ifdef __cplusplus
}
#endif
#endif /* CUDF_AMD_C_H */
