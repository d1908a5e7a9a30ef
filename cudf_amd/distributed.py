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
import torch
import torch.distributed as dist


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


def distributed_groupby_sum_count(keys, vals, stream=None, mode="shuffle", backend=None, group=None,
                                  max_message_bytes=None):
    """Global SUM(vals) and COUNT per key over all ranks; every rank returns the groups it owns
    (keys, sums, counts). The union over ranks is the global result; ownership is by key hash."""
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
    -> (global left row ids, global right row ids) as int64 tensors."""
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
