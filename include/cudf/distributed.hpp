// SPDX-License-Identifier: Apache-2.0
// Multi-GPU exchange for the hash-groupby / hash-join path: one process per GPU, RCCL over xGMI.
// The reference's data flow for this step is cpp/libcudf_streaming/src/partition_utils.cpp:72-185 (hash_partition ->
// pack -> shuffle -> unpack -> local groupby), run by rapidsmpf; here the partition is a HASH-RANGE split of the rows by owner
// rank (destination = (row_hash * world) >> 32, the top of the 32-bit MurmurHash3 row hash - SURVEY.md section 8e), and the
// exchange is RCCL point-to-point inside one group per round: counts by ncclAllGather, payload by
// ncclGroupStart { ncclSend / ncclRecv per peer } ncclGroupEnd, one buffer per column, messages of at most 1 GiB.
// RCCL is resolved at run time (dlopen): the library loads and every other entry point works without it.
// The five calls the exchange makes (all_gather, group_start, send, recv, group_end) sit behind `transport`; besides RCCL there
// is a LOOPBACK transport - V virtual ranks inside one process on one device, each driven by its own host thread, sends and
// receives matched into device copies at group_end - so that the exchange logic (counts, offsets, rounds of bounded messages,
// the per-peer Send / Recv loop) runs with real peers on a one-GPU box.
#pragma once
#include <cudf/groupby.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf/utilities/default_stream.hpp>

#include <hip/hip_runtime_api.h>

#include <array>
#include <cstdint>
#include <memory>
#include <span>
#include <vector>

namespace cudf {
namespace distributed {

constexpr std::size_t UNIQUE_ID_BYTES = 128;  // ncclUniqueId
using unique_id = std::array<char, UNIQUE_ID_BYTES>;

// What the exchange needs from a communication layer. Every call is stream-ordered on `stream` like its RCCL namesake; sends and
// receives are only issued between group_start() and group_end() and complete (in stream order) after group_end().
class transport {
 public:
  virtual ~transport()                                                                                  = default;
  virtual void all_gather(void const* send, void* recv, std::size_t count_int64, hipStream_t stream)   = 0;
  virtual void group_start()                                                                            = 0;
  virtual void send(void const* buf, std::size_t bytes, int peer, hipStream_t stream)                   = 0;
  virtual void recv(void* buf, std::size_t bytes, int peer, hipStream_t stream)                         = 0;
  virtual void group_end()                                                                              = 0;
  // A rank that leaves a collective operation by an exception OUTSIDE one of the calls above (input validation, an allocation
  // between two exchanges) tells its peers, so that they fail instead of waiting for it (loopback: releases their barriers;
  // RCCL: nothing to do here - the control plane that launched the ranks tears the job down).
  virtual void abort() noexcept {}
  [[nodiscard]] virtual void* native_handle() const noexcept { return nullptr; }  // ncclComm_t of the RCCL transport
};

// One rank's end of a communicator. RCCL: rank 0 calls make_unique_id() and hands the bytes to the other ranks through
// whatever control plane launched them (torch.distributed, MPI, a file); then every rank constructs its communicator.
// Loopback: make_loopback(V) returns the V ends of an in-process world; every end must be driven by its own host thread.
class communicator {
 public:
  static unique_id make_unique_id();
  communicator(unique_id const& id, int world_size, int rank);
  static std::vector<std::unique_ptr<communicator>> make_loopback(int world_size);
  ~communicator();
  communicator(communicator const&)            = delete;
  communicator& operator=(communicator const&) = delete;
  [[nodiscard]] int rank() const noexcept { return _rank; }
  [[nodiscard]] int size() const noexcept { return _world; }
  [[nodiscard]] void* handle() const noexcept { return _transport->native_handle(); }  // ncclComm_t (nullptr: loopback)
  [[nodiscard]] transport& link() noexcept { return *_transport; }
  // Largest single message of the payload exchange (default 1 GiB: RCCL 2.26 truncated larger ones silently); bigger slices
  // travel in several rounds. Every rank of a communicator must use the same value.
  [[nodiscard]] std::int64_t max_message_bytes() const noexcept { return _max_message_bytes; }
  void set_max_message_bytes(std::int64_t bytes);

 private:
  communicator(std::unique_ptr<transport> t, int world_size, int rank);
  std::unique_ptr<transport> _transport;
  int _world{1}, _rank{0};
  std::int64_t _max_message_bytes{std::int64_t{1} << 30};
};

// Destination rank of every row: (murmurhash3_x86_32 row hash of the key columns, seed 0) * world >> 32.
// Rows are reordered so that the rows of one destination are contiguous (in no particular order inside it); returns the table and
// world + 1 row offsets (the shape of cudf::hash_partition, with hash-range instead of modulo ownership).
std::pair<std::unique_ptr<table>, std::vector<size_type>> range_partition(
  table_view const& input, std::vector<size_type> const& key_columns, int num_destinations,
  stream_ref stream = get_default_stream(), rmm::device_async_resource_ref mr = get_current_device_resource_ref());

// Every rank passes its local rows; every rank receives the rows whose keys it owns (from rank 0, then rank 1, ...).
// Collective: all ranks of the communicator must call it, with tables of the same column types.
std::unique_ptr<table> shuffle(table_view const& input, std::vector<size_type> const& key_columns, communicator& comm,
                               stream_ref stream = get_default_stream(),
                               rmm::device_async_resource_ref mr = get_current_device_resource_ref());

// BASELINE config 5: shuffle the rows of (keys, request value columns) by key, then the local hash groupby. The groups
// of different ranks are disjoint: the union of the ranks' results is the global result, no merge step.
std::pair<std::unique_ptr<table>, std::vector<groupby::aggregation_result>> shuffle_groupby(
  table_view const& keys, std::span<groupby::aggregation_request const> requests, communicator& comm,
  null_policy null_handling = null_policy::EXCLUDE, stream_ref stream = get_default_stream(),
  rmm::device_async_resource_ref mr = get_current_device_resource_ref());

// The DECOMPOSABLE ("combiner") form of config 5 - the only one whose exchange does not grow with the rows (SURVEY.md section 8e,
// reference cpp/src/groupby/streaming_groupby/merge.cu:91-144): the local hash groupby of this rank's rows with the PARTIAL
// aggregations of every request (SUM -> sum, COUNT_* -> count, MIN / MAX -> themselves, MEAN -> sum + count of valid values,
// PRODUCT -> product, SUM_OF_SQUARES -> sum of squares), the shuffle of the partial GROUPS by key (hash-range ownership: at most
// groups x world rows travel), the merge groupby on the owner (SUM of sums and of counts, MIN of minima, MAX of maxima, PRODUCT of
// products), and the finalisation after the merge (counts back to INT32, MEAN = merged sum / merged count, NULL where no valid value
// arrived). Same result types and nullability as groupby::aggregate on the union of the ranks' rows; the ranks' results are disjoint.
// Aggregations that do not decompose this way (M2 / VARIANCE / STD, ARGMIN / ARGMAX, SUM_WITH_OVERFLOW, MEAN of a decimal or duration
// column) raise std::invalid_argument on every rank before anything is exchanged. Collective.
std::pair<std::unique_ptr<table>, std::vector<groupby::aggregation_result>> combine_groupby(
  table_view const& keys, std::span<groupby::aggregation_request const> requests, communicator& comm,
  null_policy null_handling = null_policy::EXCLUDE, stream_ref stream = get_default_stream(),
  rmm::device_async_resource_ref mr = get_current_device_resource_ref());

// Inner join of two tables sharded by rows over the ranks (data flow of the reference's
// cpp/libcudf_streaming/src/partition_utils.cpp:72-185 with a hash join as the local step): both sides are shuffled by key with the
// same hash-range ownership, the GLOBAL row id (this rank's first row id + local index; first row id = rows of the lower ranks)
// travels as a payload column, and the owner rank runs the local cudf::inner_join. Every matching (left row, right row) pair
// of the whole tables is returned exactly once, by the rank that owns its key: two INT64 columns of global row ids.
std::pair<std::unique_ptr<column>, std::unique_ptr<column>> shuffle_join(
  table_view const& left_keys, table_view const& right_keys, communicator& comm, null_equality compare_nulls = null_equality::EQUAL,
  stream_ref stream = get_default_stream(), rmm::device_async_resource_ref mr = get_current_device_resource_ref());

// Host-side bookkeeping of the payload exchange, exposed for tests (no device work): from the all-gathered count matrix
// counts[p * world + q] = rows rank p sends to rank q, what rank `me` receives from each peer, where those slices start in its
// receive buffers (world + 1 offsets), and the largest message between two DIFFERENT ranks in rows (every rank runs
// ceil(biggest / rows per message) rounds).
struct exchange_plan {
  std::vector<std::int64_t> recv_count, recv_offset;
  std::int64_t biggest{0};
};
exchange_plan plan_exchange(std::vector<std::int64_t> const& counts, int world, int me);

}  // namespace distributed
}  // namespace cudf
