"""GPU parity tests for the pieces either side of the hot path: cudf::hashing::murmurhash3_x86_32 (bit-exact vs
the oracle's reference-compatible row hash), cudf::hash_partition, cudf::gather, and the single-rank run of the
distributed groupby over RCCL."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(gpu):
    import gpu_backend
    return gpu_backend


def _cols(rng, n):
    from oracle.oracle import HostColumn
    return [
        HostColumn(rng.integers(-50, 50, n, dtype=np.int64), rng.random(n) > 0.1, "int64"),
        HostColumn(rng.integers(0, 7, n).astype(np.int32), None, "int32"),
        HostColumn(np.where(rng.random(n) < 0.05, np.nan, rng.integers(-3, 3, n) * 0.5), None, "float64"),
        HostColumn(rng.integers(0, 3, n).astype(np.uint8), None, "bool"),
        HostColumn(rng.integers(-100, 100, n).astype(np.int16), rng.random(n) > 0.5, "int16"),
        HostColumn((rng.integers(-2, 3, n) * 0.0).astype(np.float32), None, "float32"),  # +0.0 / -0.0
    ]


@pytest.mark.parametrize("seed", [0, 31])
def test_murmur3_row_hash_bit_exact(G, oracle, seed):
    import cudf_amd
    from cudf_amd import partitioning
    rng = np.random.default_rng(1)
    cols = _cols(rng, 10_000)
    for k in range(1, len(cols) + 1):
        t = cudf_amd.Table([G.to_device(c) for c in cols[:k]])
        got = partitioning.murmurhash3_x86_32(t, seed).to_numpy()[0]
        assert np.array_equal(got, oracle.row_hash(cols[:k], seed)), f"{k} columns"


@pytest.mark.parametrize("nparts", [1, 2, 3, 8, 64, 1000])
@pytest.mark.parametrize("n", [0, 1, 5000, 300_000])
def test_hash_partition(G, oracle, nparts, n):
    import cudf_amd
    from cudf_amd import partitioning
    rng = np.random.default_rng(2)
    cols = _cols(rng, n)
    t = cudf_amd.Table([G.to_device(c) for c in cols])
    out, offs = partitioning.hash_partition(t, [0, 1], nparts)
    assert len(offs) == nparts + 1 and offs[-1] == n  # reference partitioning.hpp:84-101
    offs = offs[:nparts]
    if n == 0:
        assert out.num_rows() == 0 and all(o == 0 for o in offs)
        return
    part, eoffs, order = oracle.hash_partition(cols[:2], nparts)
    assert list(offs) == [int(x) for x in eoffs]
    got = [c.to_numpy() for c in out.columns()]
    bounds = list(offs) + [n]
    # membership: every output row of partition p is a source row of partition p (compare as sorted row tuples)
    def rows(data_valid_list, idx):
        out_rows = []
        for i in idx:
            out_rows.append(tuple(None if (v is not None and not v[i]) else (d[i].item() if not (isinstance(d[i].item(), float) and np.isnan(d[i])) else "nan")
                                  for d, v in data_valid_list))
        return sorted(out_rows, key=lambda r: tuple((x is None, str(x)) for x in r))
    src = [(c.data if c.type_id != 11 else c.data != 0, c.valid) for c in cols]
    for p in range(nparts):
        if bounds[p + 1] - bounds[p] > 2000:
            continue  # checked by the count; keep the python loop small
        src_idx = np.nonzero(part == p)[0]
        assert bounds[p + 1] - bounds[p] == len(src_idx)
        assert rows(got, range(bounds[p], bounds[p + 1])) == rows(src, src_idx)
    # null counts preserved
    for c, o in zip(cols, out.columns()):
        assert o.null_count() == c.null_count


def test_hash_partition_edge_cases(G):
    """The reference's edge KATs (cpp/tests/partitioning/hash_partition_test.cpp:48-141,341-368): a column index outside
    the table, zero partitions, zero rows, zero hash columns, a custom seed."""
    import cudf_amd
    from cudf_amd import partitioning
    floats = np.arange(1, 9, dtype=np.float32)
    ints = np.arange(1, 9, dtype=np.int16)
    codes = np.array([0, 1, 2, 3, 4, 5, 6, 7], dtype=np.int32)  # (the reference's string column, as codes)
    t = cudf_amd.Table([G.to_device(floats), G.to_device(ints), G.to_device(codes)])
    with pytest.raises(IndexError):  # InvalidColumnsToHash: std::out_of_range
        partitioning.hash_partition(t, [-1], 3)
    out, offs = partitioning.hash_partition(t, [2], 0)  # ZeroPartitions
    assert out.num_columns() == 3 and out.num_rows() == 0 and len(offs) == 1
    empty = cudf_amd.Table([G.to_device(floats[:0]), G.to_device(ints[:0]), G.to_device(codes[:0])])
    out, offs = partitioning.hash_partition(empty, [2], 3)  # ZeroRows
    assert out.num_columns() == 3 and out.num_rows() == 0 and offs == [0, 0, 0, 0]
    out, offs = partitioning.hash_partition(cudf_amd.Table([]), [], 3)  # ZeroColumns
    assert out.num_columns() == 0 and out.num_rows() == 0 and len(offs) == 4
    out, offs = partitioning.hash_partition(t, [], 3)  # ZeroColumnsNonEmptyTable
    assert out.num_columns() == 3 and out.num_rows() == 0 and len(offs) == 4
    assert [int(c.type().id()) for c in out.columns()] == [int(c.type().id()) for c in t.columns()]
    o1, f1 = partitioning.hash_partition(t, [0, 2], 3, seed=12345)  # CustomSeedValue: deterministic, same shape
    o2, f2 = partitioning.hash_partition(t, [0, 2], 3, seed=12345)
    assert len(f1) == 4 and f1 == f2 and o1.num_rows() == 8
    for a, b in zip(o1.columns(), o2.columns()):
        assert np.array_equal(a.to_numpy()[0], b.to_numpy()[0])
    o3, f3 = partitioning.hash_partition(t, [0, 2], 3, seed=0)
    assert sorted(o3.columns()[1].to_numpy()[0].tolist()) == list(range(1, 9))


@pytest.mark.parametrize("nparts", [2048, 4096])
def test_hash_partition_many_partitions(G, oracle, nparts):
    """More than 1024 partitions need more than the default 64 KiB of dynamic LDS (139 KiB at 4096): the kernel opts in."""
    import cudf_amd
    from cudf_amd import partitioning
    rng = np.random.default_rng(9)
    n = 200_000
    k = rng.integers(0, 1 << 40, n, dtype=np.int64)
    v = rng.random(n)
    out, offs = partitioning.hash_partition(cudf_amd.Table([G.to_device(k), G.to_device(v)]), [0], nparts)
    part, eoffs, order = oracle.hash_partition([k], nparts)
    assert offs[:nparts] == [int(x) for x in eoffs] and offs[-1] == n
    ok, ov = out.columns()[0].to_numpy()[0], out.columns()[1].to_numpy()[0]
    assert np.array_equal(oracle.hash_partition([ok], nparts)[0], np.repeat(np.arange(nparts), np.diff(offs)))
    assert sorted(zip(ok.tolist(), ov.tolist())) == sorted(zip(k.tolist(), v.tolist()))


def test_gather_negative_indices(G):
    """Negative indices of a signed gather map count from the end (reference copying/gather.cu:81-82); JoinNoMatch
    (INT32_MIN) stays out of bounds."""
    import cudf_amd
    from cudf_amd import partitioning
    from cudf_amd.types import OutOfBoundsPolicy
    data = np.arange(100, 110, dtype=np.int64)
    valid = np.array([1, 1, 0, 1, 1, 1, 1, 1, 1, 0], bool)
    t = cudf_amd.Table([G.to_device((data, valid)), G.to_device(np.arange(10, dtype=np.float64))])
    idx = np.array([-1, -10, 0, 9, -2**31, 10, -3, -11], np.int32)
    out = partitioning.gather(t, G.to_device(idx), OutOfBoundsPolicy.NULLIFY)
    d0, v0 = out.columns()[0].to_numpy()
    d1, v1 = out.columns()[1].to_numpy()
    assert v1.tolist() == [True, True, True, True, False, False, True, False]
    assert d1[v1].tolist() == [9.0, 0.0, 0.0, 9.0, 7.0]
    assert v0.tolist() == [False, True, True, False, False, False, True, False]
    assert d0[v0].tolist() == [100, 100, 107]


def test_gather_with_nullify(G):
    import cudf_amd
    from cudf_amd import partitioning
    from cudf_amd.types import OutOfBoundsPolicy
    rng = np.random.default_rng(3)
    n = 1000
    data = rng.integers(0, 1 << 40, n, dtype=np.int64)
    valid = rng.random(n) > 0.2
    t = cudf_amd.Table([G.to_device((data, valid)), G.to_device(rng.random(n).astype(np.float32))])
    idx = rng.integers(0, n, 5000).astype(np.int32)
    idx[::7] = -2**31  # JoinNoMatch
    out = partitioning.gather(t, G.to_device(idx), OutOfBoundsPolicy.NULLIFY)
    d0, v0 = out.columns()[0].to_numpy()
    d1, v1 = out.columns()[1].to_numpy()
    ok = idx >= 0
    assert np.array_equal(v0, np.where(ok, valid[np.where(ok, idx, 0)], False))
    assert np.array_equal(d0[v0], data[idx[v0]])
    assert np.array_equal(v1, ok)
    assert out.columns()[1].null_count() == int((~ok).sum())


def test_join_then_gather_payload(G, oracle):
    """C3 end to end at small size: inner join on int64 keys with 5% nulls, then gather 2 float64 payload columns
    per side by the returned indices (the step every caller runs after inner_join)."""
    import cudf_amd
    from cudf_amd import join as J, partitioning
    from cudf_amd.types import NullEquality
    rng = np.random.default_rng(44)
    nl, nr = 50_000, 5_000
    rk = rng.permutation(2 * nr)[:nr].astype(np.int64)
    lk = rng.integers(0, 4 * nr, nl, dtype=np.int64)
    lv, rv = rng.random(nl) > 0.05, rng.random(nr) > 0.05
    lp = [rng.random(nl), rng.random(nl)]
    rp = [rng.random(nr), rng.random(nr)]
    li, ri = J.inner_join(cudf_amd.Table([G.to_device((lk, lv))]), cudf_amd.Table([G.to_device((rk, rv))]), NullEquality.UNEQUAL)
    gl = partitioning.gather(cudf_amd.Table([G.to_device(x) for x in lp]), li)
    gr = partitioning.gather(cudf_amd.Table([G.to_device(x) for x in rp]), ri)
    el, er = oracle.join([(lk, lv)], [(rk, rv)], nulls_equal=False)
    got = sorted(zip(li.to_numpy()[0].tolist(), ri.to_numpy()[0].tolist(), gl.columns()[0].to_numpy()[0].tolist(),
                     gl.columns()[1].to_numpy()[0].tolist(), gr.columns()[0].to_numpy()[0].tolist(), gr.columns()[1].to_numpy()[0].tolist()))
    exp = sorted(zip(el.tolist(), er.tolist(), lp[0][el].tolist(), lp[1][el].tolist(), rp[0][er].tolist(), rp[1][er].tolist()))
    assert got == exp


@pytest.mark.parametrize("mode", ["shuffle", "preaggregate"])
def test_distributed_single_rank_rccl(G, oracle, mode):
    """world_size 1 over the nccl (RCCL) backend on the GPU: same code path the multi-GPU bench runs."""
    import torch
    import torch.distributed as dist
    import kat
    from cudf_amd import distributed as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(5)
        n = 400_000
        k = rng.integers(0, 30_000, n, dtype=np.int64)
        v = rng.random(n)
        gk, gs, gc = D.distributed_groupby_sum_count(torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda(), mode=mode)
        kc, rc = kat.sort_groups(*oracle.groupby([k], [(v, ["sum", "count_valid"])]))
        o = torch.argsort(gk)
        assert np.array_equal(gk[o].cpu().numpy(), kc[0][0])
        assert np.array_equal(gc[o].cpu().numpy().astype(np.int64), rc[0][1][0].astype(np.int64))
        assert np.allclose(gs[o].cpu().numpy(), rc[0][0][0], rtol=1e-12)
    finally:
        dist.destroy_process_group()


def test_distributed_inner_join_single_rank_rccl(G, oracle):
    """distributed_inner_join over the nccl (RCCL) backend at world_size 1: hash partition of (key, global row id) on both
    sides -> all-to-all -> local cudf::inner_join -> global row ids."""
    import torch
    import torch.distributed as dist
    import kat
    from cudf_amd import distributed as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(6)
        lk = rng.integers(0, 50_000, 300_000, dtype=np.int64)
        rk = rng.integers(0, 50_000, 40_000, dtype=np.int64)
        gl, gr = D.distributed_inner_join(torch.from_numpy(lk).cuda(), torch.from_numpy(rk).cuda())
        el, er = oracle.join([lk], [rk], nulls_equal=True, kind="inner")
        assert kat.sorted_pairs(gl.cpu().numpy(), gr.cpu().numpy()) == kat.sorted_pairs(el, er)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ndest", [2, 3, 8])
def test_range_partition_hash_range_ownership(G, oracle, ndest):
    """cudf::distributed::range_partition: destination = (murmur3 row hash * N) >> 32 on the high hash bits (SURVEY.md
    section 8e), rows of one destination contiguous, nulls and narrow types carried, num_destinations + 1 offsets."""
    import cudf_amd
    from cudf_amd import distributed as D
    rng = np.random.default_rng(12)
    n = 300_000
    cols = _cols(rng, n)
    t = cudf_amd.Table([G.to_device(c) for c in cols])
    out, offs = D.range_partition(t, [0, 1], ndest)
    assert len(offs) == ndest + 1 and offs[0] == 0 and offs[-1] == n
    h = oracle.row_hash(cols[:2], 0).astype(np.uint64)
    dest = ((h * np.uint64(ndest)) >> np.uint64(32)).astype(np.int64)
    assert np.array_equal(np.diff(offs), np.bincount(dest, minlength=ndest))
    got = [c.to_numpy() for c in out.columns()]
    for c, o in zip(cols, out.columns()):
        assert o.null_count() == c.null_count
    # every output row of destination p hashes to p, and the multiset of rows is preserved
    from oracle.oracle import HostColumn
    oh = oracle.row_hash([HostColumn(got[0][0], got[0][1], "int64"), HostColumn(got[1][0], None, "int32")], 0).astype(np.uint64)
    odest = ((oh * np.uint64(ndest)) >> np.uint64(32)).astype(np.int64)
    assert np.array_equal(odest, np.repeat(np.arange(ndest), np.diff(offs)))
    key = lambda d, v: sorted(zip(np.where(v, d[0], -999).tolist() if v is not None else d[0].tolist()))
    for (d, v), c in zip(got, cols):
        src = c.data if c.type_id != 11 else c.data != 0
        a = np.where(v, d, 0) if v is not None else d
        b = np.where(c.valid, src, 0) if c.valid is not None else src
        assert sorted(np.nan_to_num(a.astype(np.float64), nan=-7.5).tolist()) == sorted(np.nan_to_num(b.astype(np.float64), nan=-7.5).tolist())


def test_native_shuffle_groupby_single_rank_rccl(G, oracle):
    """The literal config-5 form through the C++ entry points at world_size 1: the library creates its own RCCL
    communicator (ncclCommInitRank), allgathers the counts, copies its own slice, and runs the local hash groupby on the
    received rows; nullable values travel with validity bytes."""
    import cudf_amd
    import kat
    from cudf_amd import aggregation as agg, distributed as D, groupby as gb
    from cudf_amd.types import NullPolicy
    rng = np.random.default_rng(15)
    n = 500_000
    k = rng.integers(-40_000, 40_000, n, dtype=np.int64)
    v = rng.random(n)
    vv = rng.random(n) > 0.1
    comm = D.Communicator(world_size=1, rank=0)
    t = cudf_amd.Table([G.to_device(k), G.to_device((v, vv))])
    mine = D.shuffle(comm, t, [0])
    assert mine.num_rows() == n and mine.columns()[1].null_count() == int((~vv).sum())
    gk, gv = mine.columns()[0].to_numpy()[0], mine.columns()[1].to_numpy()
    assert sorted(zip(gk.tolist(), np.where(gv[1], gv[0], -1.0).tolist())) == sorted(zip(k.tolist(), np.where(vv, v, -1.0).tolist()))
    req = gb.GroupByRequest(G.to_device((v, vv)), [agg.sum(), agg.count(NullPolicy.EXCLUDE), agg.max()])
    uk, res = D.shuffle_groupby(comm, cudf_amd.Table([G.to_device(k)]), [req])
    got = kat.sort_groups([G.from_device(c) for c in uk.columns()], [[G.from_device(c) for c in res[0].columns()]])
    exp = kat.sort_groups(*oracle.groupby([k], [((v, vv), ["sum", "count_valid", "max"])]))
    kat.compare_columns(got[0][0], exp[0][0], "keys")
    for a, e, name in zip(got[1][0], exp[1][0], ["sum", "count", "max"]):
        kat.compare_columns(a, e, name, atol=kat.sum_atol(64, 1.0) if name == "sum" else 0.0)
