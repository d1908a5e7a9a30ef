"""Multi-GPU groupby over RCCL (torch.distributed, one process per GPU): the MI355X counterpart of the
reference's shuffle data-flow (cpp/libcudf_streaming/src/partition_utils.cpp:72-185: hash_partition -> pack ->
rapidsmpf shuffler -> unpack -> local groupby). Two variants:

  shuffle       hash-partition the local rows by destination rank -> all-to-all-v of every column -> local groupby.
                This is BASELINE config 5. Groups are disjoint across ranks, no final merge. xGMI carries
                (N-1)/N of the input bytes.
  preaggregate  local groupby first (<= G partial rows per rank) -> hash-partition the partials -> all-to-all-v
                -> merge (SUM of sums, SUM of counts). The decomposable form; reference precedent
                cudf::groupby::streaming_groupby::merge (cpp/src/groupby/streaming_groupby/merge.cu:91-144).
                xGMI carries only the partials.

The exchange itself (counts all-to-all, then one all_to_all_single per column with split sizes) is independent of
the device: `backend` supplies the local hash partition and the local groupby, so the collective logic is covered
by world_size-2 gloo tests on CPU with a host backend (tests/test_distributed_gloo.py).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib
from .column import Column, Table, _stream_ptr


# ------------------------------------------------------------------ native exchange (C++: RCCL Send/Recv inside the library)
class Communicator:
    """cudf::distributed::communicator (include/cudf/distributed.hpp): this rank's end of an RCCL communicator created by
    the library itself. The 128-byte ncclUniqueId travels from rank 0 to the other ranks over `group` (any torch.distributed
    backend; only the control plane - the payload goes through the library's own RCCL communicator)."""

    def __init__(self, group=None, world_size=None, rank=None):
        lib = _lib.load()
        if world_size is None:
            world_size = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._handle = C.c_void_p()
        ident = (C.c_uint8 * 128)()
        # EVERY rank takes part in the id broadcast, whatever happened on rank 0: rank 0 broadcasts the id or an error marker,
        # and all ranks raise together after the broadcast (a rank 0 that raised before it left the others waiting in a
        # collective the rest of the job had already moved past).
        error = None
        if rank == 0:
            try:
                _lib.check(lib.cudf_amd_comm_unique_id(ident))
            except Exception as e:  # noqa: BLE001 - travels to every rank below
                error = repr(e)
        if world_size > 1:
            box = [("error", error) if error is not None else ("id", bytes(ident))]
            dist.broadcast_object_list(box, src=0, group=group)
            kind, payload = box[0]
            if kind == "error":
                error = payload
            else:
                ident = (C.c_uint8 * 128)(*payload)
        if error is not None:
            raise _lib.CudfAmdError(f"communicator: rank 0 could not create the RCCL unique id: {error}")
        _lib.check(lib.cudf_amd_comm_create(ident, world_size, rank, C.byref(self._handle)))
        self.world_size, self.rank = world_size, rank

    @classmethod
    def loopback(cls, world_size: int):
        """The world_size ends of an in-process LOOPBACK world (include/cudf/distributed.hpp `transport`): virtual ranks on the
        current device, sends and receives matched into device copies. Every end must be driven by its own host thread - the
        collectives rendezvous (ctypes releases the GIL for the duration of a call)."""
        lib = _lib.load()
        handles = (C.c_void_p * world_size)()
        _lib.check(lib.cudf_amd_comm_create_loopback(world_size, handles))
        ends = []
        for r in range(world_size):
            c = cls.__new__(cls)
            c._handle = C.c_void_p(handles[r])
            c.world_size, c.rank = world_size, r
            ends.append(c)
        return ends

    def set_max_message_bytes(self, nbytes: int):
        """Largest single message of the payload exchange (default 1 GiB); every rank must set the same value."""
        _lib.check(_lib.load().cudf_amd_comm_set_max_message_bytes(self._handle, int(nbytes)))

    def close(self):
        """Destroys the RCCL communicator (call before torch.distributed.destroy_process_group / interpreter teardown:
        ncclCommDestroy after the HIP runtime has shut down can hang)."""
        if self._handle:
            _lib.load().cudf_amd_comm_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def range_partition(table: Table, key_columns, num_destinations: int, stream=None):
    """-> (Table with the rows of one destination contiguous, num_destinations + 1 row offsets); destination of a row =
    (murmur3 row hash of its key columns * num_destinations) >> 32 (hash-range ownership)."""
    cols = (C.c_int32 * max(1, len(key_columns)))(*key_columns)
    offs = (C.c_int32 * (num_destinations + 1))()
    out = C.c_void_p()
    _lib.check(_lib.load().cudf_amd_range_partition(table._views(), table.num_columns(), cols, len(key_columns), num_destinations,
                                                    _stream_ptr(stream), C.byref(out), offs))
    return Table._from_handle(out, stream), list(offs)


def shuffle(comm: Communicator, table: Table, key_columns, stream=None) -> Table:
    """Collective: every rank passes its rows and receives the rows whose keys it owns."""
    cols = (C.c_int32 * max(1, len(key_columns)))(*key_columns)
    out = C.c_void_p()
    _lib.check(_lib.load().cudf_amd_shuffle(comm._handle, table._views(), table.num_columns(), cols, len(key_columns),
                                            _stream_ptr(stream), C.byref(out)))
    return Table._from_handle(out, stream)


def shuffle_groupby(comm: Communicator, keys: Table, requests, null_handling=0, stream=None, _entry="cudf_amd_shuffle_groupby"):
    """BASELINE config 5 inside the library: hash-range partition of the rows -> RCCL exchange -> local hash groupby.
    requests: cudf_amd.groupby.GroupByRequest list. -> (keys Table, [results Table per request]) of the groups this rank owns."""
    reqs, keep = [], []
    for r in requests:
        reqs.append(_lib.AggregationRequest.of(r._values._view(), r._aggregations, keep))
    rarr = (_lib.AggregationRequest * max(1, len(reqs)))(*reqs)
    out_keys, out_res = C.c_void_p(), C.c_void_p()
    _lib.check(getattr(_lib.load(), _entry)(comm._handle, keys._views(), keys.num_columns(), int(null_handling), rarr,
                                             len(reqs), _stream_ptr(stream), C.byref(out_keys), C.byref(out_res)))
    flat = Table._from_handle(out_res, stream).columns()
    results, p = [], 0
    for r in requests:
        results.append(Table(flat[p:p + len(r._aggregations)]))
        p += len(r._aggregations)
    return Table._from_handle(out_keys, stream), results


def combine_groupby(comm: Communicator, keys: Table, requests, null_handling=0, stream=None):
    """The decomposable form of config 5 inside the library (cudf::distributed::combine_groupby): local groupby with partial
    aggregations -> exchange of the partial GROUPS -> merge on the owner -> finalisation (MEAN after the merge). Same results as
    shuffle_groupby; SUM / PRODUCT / SUM_OF_SQUARES / MIN / MAX / COUNT / MEAN only."""
    return shuffle_groupby(comm, keys, requests, null_handling, stream, _entry="cudf_amd_combine_groupby")


def shuffle_join(comm: Communicator, left_keys: Table, right_keys: Table, nulls_equal=True, stream=None):
    """cudf::distributed::shuffle_join: inner join of two row-sharded key tables. Collective. -> (global left row ids, global right
    row ids) as INT64 Columns: the pairs whose key this rank owns."""
    out = C.c_void_p()
    _lib.check(_lib.load().cudf_amd_shuffle_join(comm._handle, left_keys._views(), left_keys.num_columns(), right_keys._views(),
                                                 right_keys.num_columns(), 1 if nulls_equal else 0, _stream_ptr(stream), C.byref(out)))
    cols = Table._from_handle(out, stream).columns()
    return cols[0], cols[1]


def plan_exchange(counts, world_size: int, rank: int):
    """cudf::distributed::plan_exchange (host arithmetic only): counts[p][q] = rows rank p sends to rank q ->
    (rows received from each peer, world_size + 1 receive offsets, largest message between two different ranks in rows)."""
    flat = (C.c_int64 * (world_size * world_size))(*[int(counts[p][q]) for p in range(world_size) for q in range(world_size)])
    rc, ro, big = (C.c_int64 * world_size)(), (C.c_int64 * (world_size + 1))(), C.c_int64()
    _lib.check(_lib.load().cudf_amd_plan_exchange(flat, world_size, rank, rc, ro, C.byref(big)))
    return list(rc), list(ro), big.value


class GpuBackend:
    """Local compute through the product (HIP kernels behind the C ABI)."""

    def __init__(self, stream=None):
        self.stream = stream

    def inner_join(self, left_keys, right_keys):
        """Local inner join of two 1-D key tensors -> (left row indices, right row indices) as int64 cuda tensors."""
        import cudf_amd
        from cudf_amd import join as J
        li, ri = J.inner_join(cudf_amd.Table([cudf_amd.Column.from_torch(left_keys)]),
                              cudf_amd.Table([cudf_amd.Column.from_torch(right_keys)]), stream=self.stream)
        return li.to_torch().to(torch.int64), ri.to_torch().to(torch.int64)

    def partition(self, columns, num_partitions):
        """columns: list of 1-D cuda tensors, column 0 is the key. -> (list of partitioned tensors, offsets list)."""
        import cudf_amd
        from cudf_amd import partitioning
        tbl = cudf_amd.Table([cudf_amd.Column.from_torch(c) for c in columns])
        out, offs = partitioning.hash_partition(tbl, [0], num_partitions, stream=self.stream)
        return [c.to_torch() for c in out.columns()], offs[:num_partitions]  # (start offsets; the last entry is the row count)

    def groupby_sum(self, keys, value_columns, count=False):
        """SUM of every value column per key (and COUNT_VALID of the first when count=True).
        -> (keys, [sums...], count or None) as cuda tensors."""
        import cudf_amd
        from cudf_amd import aggregation as agg, groupby as gb
        from cudf_amd.types import NullPolicy
        reqs = []
        for i, v in enumerate(value_columns):
            aggs = [agg.sum()] + ([agg.count(NullPolicy.EXCLUDE)] if (count and i == 0) else [])
            reqs.append(gb.GroupByRequest(cudf_amd.Column.from_torch(v), aggs))
        g = gb.GroupBy(cudf_amd.Table([cudf_amd.Column.from_torch(keys)]))
        uk, res = g.aggregate(reqs, stream=self.stream)
        sums = [r.columns()[0].to_torch() for r in res]
        cnt = res[0].columns()[1].to_torch() if count else None
        return uk.columns()[0].to_torch(), sums, cnt


# RCCL (2.26, ROCm 7.0) silently truncates all_to_all_single messages beyond 2^31 bytes per peer (measured: a
# 2.4 GB self-exchange returned half of the rows as zeros), so big columns go in rounds of bounded messages.
MAX_MESSAGE_BYTES = 1 << 30


def exchange(columns, offsets, group=None, max_message_bytes=None):
    """All-to-all-v of partitioned columns. `offsets[p]` = first row of the slice destined for rank p.
    -> list of received tensors (rows from rank 0, then rank 1, ...). No message exceeds max_message_bytes."""
    limit = max_message_bytes or MAX_MESSAGE_BYTES
    world = dist.get_world_size(group)
    n = columns[0].numel()
    bounds = list(offsets) + [n]
    send_counts = [bounds[p + 1] - bounds[p] for p in range(world)]
    dev = columns[0].device
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    rc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rc, sc, group=group)
    biggest = torch.max(torch.stack([sc.max(), rc.max()])).reshape(1)
    dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)  # every rank runs the same number of rounds
    recv_counts = [int(x) for x in rc.tolist()]
    biggest = int(biggest.item())
    total = sum(recv_counts)
    soff = bounds[:-1]
    roff = [0] * world
    for p in range(1, world):
        roff[p] = roff[p - 1] + recv_counts[p - 1]
    list_ok = dist.get_backend(group) == "nccl"  # gloo has no list all_to_all
    out = []
    for c in columns:
        r = torch.empty(total, dtype=c.dtype, device=dev)
        chunk = max(1, limit // c.element_size())
        rounds = max(1, -(-biggest // chunk))
        if rounds == 1:
            dist.all_to_all_single(r, c, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
        else:
            for k in range(rounds):
                ins = [c[soff[p] + min(k * chunk, send_counts[p]): soff[p] + min((k + 1) * chunk, send_counts[p])]
                       for p in range(world)]
                outs = [r[roff[p] + min(k * chunk, recv_counts[p]): roff[p] + min((k + 1) * chunk, recv_counts[p])]
                        for p in range(world)]
                if list_ok:
                    dist.all_to_all(outs, ins, group=group)  # views: no staging copies
                else:
                    packed = torch.empty(sum(o.numel() for o in outs), dtype=c.dtype, device=dev)
                    dist.all_to_all_single(packed, torch.cat(ins), output_split_sizes=[o.numel() for o in outs],
                                           input_split_sizes=[i.numel() for i in ins], group=group)
                    pos = 0
                    for o in outs:
                        o.copy_(packed[pos:pos + o.numel()])
                        pos += o.numel()
        out.append(r)
    return out


import weakref

_COMMS = weakref.WeakKeyDictionary()  # process group OBJECT -> Communicator (an id() can be reused once a group is collected)
_DEFAULT_COMM = []                    # the default group (group=None)


def _native_comm(group=None):
    """One library communicator per torch process group (created on first use: a collective)."""
    if group is None:
        if not _DEFAULT_COMM:
            _DEFAULT_COMM.append(Communicator(None))
        return _DEFAULT_COMM[0]
    if group not in _COMMS:
        _COMMS[group] = Communicator(group)
    return _COMMS[group]


def close_communicators():
    """Destroys every cached library communicator (before destroy_process_group())."""
    for c in list(_COMMS.values()) + list(_DEFAULT_COMM):
        c.close()
    _COMMS.clear()
    _DEFAULT_COMM.clear()


def distributed_groupby_sum_count(keys, vals, stream=None, mode="shuffle", backend=None, group=None,
                                  max_message_bytes=None):
    """Global SUM(vals) and COUNT per key over all ranks; every rank returns the groups it owns
    (keys, sums, counts). The union over ranks is the global result; ownership is by key hash."""
    if mode == "shuffle_native":  # the literal config-5 form, entirely inside the library (C++ + RCCL)
        import cudf_amd
        from cudf_amd import aggregation as agg, groupby as gb
        from cudf_amd.types import NullPolicy
        comm = _native_comm(group)
        req = gb.GroupByRequest(cudf_amd.Column.from_torch(vals), [agg.sum(), agg.count(NullPolicy.EXCLUDE)])
        uk, res = shuffle_groupby(comm, cudf_amd.Table([cudf_amd.Column.from_torch(keys)]), [req], stream=stream)
        return uk.columns()[0].to_torch(), res[0].columns()[0].to_torch(), res[0].columns()[1].to_torch()
    if mode == "combine_native":  # the decomposable form, entirely inside the library
        import cudf_amd
        from cudf_amd import aggregation as agg, groupby as gb
        from cudf_amd.types import NullPolicy
        comm = _native_comm(group)
        req = gb.GroupByRequest(cudf_amd.Column.from_torch(vals), [agg.sum(), agg.count(NullPolicy.EXCLUDE)])
        uk, res = combine_groupby(comm, cudf_amd.Table([cudf_amd.Column.from_torch(keys)]), [req], stream=stream)
        return uk.columns()[0].to_torch(), res[0].columns()[0].to_torch(), res[0].columns()[1].to_torch()
    backend = backend or GpuBackend(stream)
    world = dist.get_world_size(group)
    if mode == "shuffle":
        cols, offs = backend.partition([keys, vals], world)
        rk, rv = exchange(cols, offs, group, max_message_bytes)
        k, sums, cnt = backend.groupby_sum(rk, [rv], count=True)
        return k, sums[0], cnt
    if mode == "preaggregate":
        k, sums, cnt = backend.groupby_sum(keys, [vals], count=True)
        cols, offs = backend.partition([k, sums[0], cnt.to(torch.int64)], world)
        rk, rs, rc = exchange(cols, offs, group, max_message_bytes)
        k2, sums2, _ = backend.groupby_sum(rk, [rs, rc], count=False)
        return k2, sums2[0], sums2[1]
    raise ValueError(f"unknown mode {mode!r}")


def distributed_inner_join(left_keys, right_keys, stream=None, backend=None, group=None, max_message_bytes=None):
    """Inner join of two tables sharded by rows over the ranks: both sides are hash-partitioned
    by key with the same hash, every partition meets on its owner rank (one all-to-all per side, the GLOBAL row id
    = rank offset + local index travels as a payload column), and the owner joins locally. Every matching
    (left row, right row) pair of the whole tables is returned exactly once, by the rank that owns the key:
    -> (global left row ids, global right row ids) as int64 tensors.
    Default: cudf::distributed::shuffle_join inside the library (hash-range ownership, RCCL). With a `backend` (the CPU tests'
    host backend) or a message limit, the same data flow through torch.distributed."""
    if backend is None and max_message_bytes is None:
        # inside the library: cudf::distributed::shuffle_join over the library's own communicator (RCCL Send / Recv)
        import cudf_amd
        comm = _native_comm(group)
        li, ri = shuffle_join(comm, cudf_amd.Table([cudf_amd.Column.from_torch(left_keys)]),
                              cudf_amd.Table([cudf_amd.Column.from_torch(right_keys)]), stream=stream)
        return li.to_torch(), ri.to_torch()
    backend = backend or GpuBackend(stream)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = left_keys.device

    def global_ids(n):
        counts = torch.zeros(world, dtype=torch.int64, device=dev)
        counts[rank] = n
        dist.all_reduce(counts, group=group)  # rows per rank; exclusive prefix = this rank's first global row id
        first = int(counts[:rank].sum().item())
        return torch.arange(first, first + n, dtype=torch.int64, device=dev)

    sides = []
    for keys in (left_keys, right_keys):
        cols, offs = backend.partition([keys, global_ids(keys.numel())], world)
        sides.append(exchange(cols, offs, group, max_message_bytes))
    (lk, lid), (rk, rid) = sides
    li, ri = backend.inner_join(lk, rk)
    return lid[li], rid[ri]
