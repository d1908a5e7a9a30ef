"""CPU, world_size 2, gloo: covers the N>1 path of cudf_amd.distributed (counts all-to-all, per-column
all_to_all_single with split sizes, both the raw-row shuffle and the pre-aggregated variant) with a host backend
built on the CPU oracle; checks that the union of the ranks' results equals a single-process groupby of all rows
and that ownership is disjoint."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class HostBackend:
    """Host stand-in for the local compute (test infrastructure): oracle hash partition + oracle groupby."""

    def partition(self, columns, num_partitions):
        from oracle import oracle as O
        k = columns[0].numpy()
        part, offs, order = O.hash_partition([k], num_partitions)
        return [c[torch.from_numpy(order.astype(np.int64))] for c in columns], [int(x) for x in offs]

    def groupby_sum(self, keys, value_columns, count=False):
        from oracle import oracle as O
        reqs = [(v.numpy(), ["sum"] + (["count_valid"] if (count and i == 0) else [])) for i, v in enumerate(value_columns)]
        kc, rc = O.groupby([keys.numpy()], reqs)
        sums = [torch.from_numpy(r[0][0].copy()) for r in rc]
        cnt = torch.from_numpy(rc[0][1][0].copy()) if count else None
        return torch.from_numpy(kc[0][0].copy()), sums, cnt


def _host_inner_join(self, left_keys, right_keys):
    from oracle import oracle as O
    li, ri = O.join([left_keys.numpy()], [right_keys.numpy()], nulls_equal=True, kind="inner")
    return torch.from_numpy(np.asarray(li, dtype=np.int64)), torch.from_numpy(np.asarray(ri, dtype=np.int64))


HostBackend.inner_join = _host_inner_join


def _join_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cudf_amd import distributed as D
        rng = np.random.default_rng(7 + rank)
        lk = torch.from_numpy(rng.integers(0, 4000, 9_000 + 500 * rank, dtype=np.int64))   # ragged shards, duplicates
        rk = torch.from_numpy(rng.integers(0, 4000, 2_000 + 300 * rank, dtype=np.int64))
        gl, gr = D.distributed_inner_join(lk, rk, backend=HostBackend())
        np.savez(os.path.join(out_dir, f"j{rank}.npz"), gl=gl.numpy(), gr=gr.numpy(), lk=lk.numpy(), rk=rk.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_inner_join(tmp_path):
    """Both sides sharded over two ranks: the union of the ranks' (global left id, global right id) pairs equals the
    inner join of the concatenated tables, each pair exactly once."""
    world = 2
    mp.spawn(_join_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"j{r}.npz") for r in range(world)]
    all_l = np.concatenate([p["lk"] for p in parts])
    all_r = np.concatenate([p["rk"] for p in parts])
    from oracle import oracle as O
    el, er = O.join([all_l], [all_r], nulls_equal=True, kind="inner")
    got = sorted(zip(np.concatenate([p["gl"] for p in parts]).tolist(), np.concatenate([p["gr"] for p in parts]).tolist()))
    assert got == sorted(zip(np.asarray(el).tolist(), np.asarray(er).tolist()))
    assert len(set(got)) == len(got)


def _worker(rank, world, port, mode, out_dir, max_message_bytes=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cudf_amd import distributed as D
        rng = np.random.default_rng(42 + rank)
        n = 20_000 + 1000 * rank  # ragged shards
        keys = torch.from_numpy(rng.integers(0, 3000, n, dtype=np.int64))
        vals = torch.from_numpy(rng.random(n))
        k, s, c = D.distributed_groupby_sum_count(keys, vals, mode=mode, backend=HostBackend(),
                                                  max_message_bytes=max_message_bytes)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), k=k.numpy(), s=s.numpy(), c=c.numpy(), keys=keys.numpy(), vals=vals.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode,max_message_bytes", [("shuffle", None), ("preaggregate", None), ("shuffle", 16 * 1024)])
def test_two_rank_groupby(tmp_path, mode, max_message_bytes):
    """max_message_bytes=16 KiB forces the multi-round exchange (RCCL truncates messages > 2^31 bytes)."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path), max_message_bytes), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    # ownership is disjoint
    assert len(np.intersect1d(parts[0]["k"], parts[1]["k"])) == 0
    allk = np.concatenate([p["keys"] for p in parts])
    allv = np.concatenate([p["vals"] for p in parts])
    import pandas as pd
    ref = pd.DataFrame({"k": allk, "v": allv}).groupby("k")["v"].agg(["sum", "count"]).sort_index()
    got_k = np.concatenate([p["k"] for p in parts])
    got_s = np.concatenate([p["s"] for p in parts])
    got_c = np.concatenate([p["c"] for p in parts])
    o = np.argsort(got_k)
    assert np.array_equal(got_k[o], ref.index.to_numpy())
    assert np.array_equal(got_c[o].astype(np.int64), ref["count"].to_numpy())
    assert np.allclose(got_s[o], ref["sum"].to_numpy(), rtol=1e-12)


def test_exchange_handles_empty_slices(tmp_path):
    """A rank that sends nothing to a peer (offsets with an empty slice) must still complete the collective."""
    world = 2
    mp.spawn(_empty_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "e0.npy"), np.load(tmp_path / "e1.npy")
    assert list(a) == [0, 1, 2, 10, 11] and list(b) == []


def _empty_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cudf_amd import distributed as D
        col = torch.arange(3, dtype=torch.int64) + 10 * rank if rank == 0 else torch.tensor([10, 11], dtype=torch.int64)
        # everything goes to rank 0: offsets [0, n]
        out = D.exchange([col], [0, col.numel()])
        np.save(os.path.join(out_dir, f"e{rank}.npy"), out[0].numpy())
    finally:
        dist.destroy_process_group()
