// SPDX-License-Identifier: Apache-2.0
// Multi-GPU exchange for the hash-groupby / hash-join path: one process per GPU, RCCL over xGMI.
// The reference's data flow for this step is cpp/libcudf_streaming/src/partition_utils.cpp:72-185 (hash_partition ->
// pack -> shuffle -> unpack -> local groupby), run by rapidsmpf; here the partition is a HASH-RANGE split of the rows by owner
// rank (destination = (row_hash * world) >> 32, the top of the 32-bit MurmurHash3 row hash - SURVEY.md section 8e), and the
// exchange is RCCL point-to-point inside one group per round: counts by ncclAllGather, payload by
// ncclGroupStart { ncclSend / ncclRecv per peer } ncclGroupEnd, one buffer per column, messages of at most 1 GiB.
// RCCL is resolved at run time (dlopen): the library loads and every other entry point works without it.
#pragma once
#include <cudf/groupby.hpp>
#include <cudf/table/table.hpp>
#include <cudf/table/table_view.hpp>
#include <cudf/utilities/default_stream.hpp>

#include <array>
#include <cstdint>
#include <memory>
#include <span>
#include <vector>

namespace cudf {
namespace distributed {

constexpr std::size_t UNIQUE_ID_BYTES = 128;  // ncclUniqueId
using unique_id = std::array<char, UNIQUE_ID_BYTES>;

// One rank's end of an RCCL communicator. Rank 0 calls make_unique_id() and hands the bytes to the other ranks through
// whatever control plane launched them (torch.distributed, MPI, a file); then every rank constructs its communicator.
class communicator {
 public:
  static unique_id make_unique_id();
  communicator(unique_id const& id, int world_size, int rank);
  ~communicator();
  communicator(communicator const&)            = delete;
  communicator& operator=(communicator const&) = delete;
  [[nodiscard]] int rank() const noexcept { return _rank; }
  [[nodiscard]] int size() const noexcept { return _world; }
  [[nodiscard]] void* handle() const noexcept { return _comm; }  // ncclComm_t

 private:
  void* _comm{nullptr};
  int _world{1}, _rank{0};
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

}  // namespace distributed
}  // namespace cudf
