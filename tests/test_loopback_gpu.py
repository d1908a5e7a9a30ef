"""The in-library exchange (cudf_amd/csrc/distributed/shuffle.hip) run WITH PEERS on one GPU: V virtual ranks of a loopback
communicator (include/cudf/distributed.hpp `transport`), one host thread and one HIP stream per rank. The same code drives RCCL on
a multi-GPU node; here sends and receives are matched into device copies at group_end, so the per-peer Send / Recv loop, the
count matrix, the receive offsets and the rounds of bounded messages all execute with world sizes 2, 4 and 8.

Checked against the CPU oracle (test infrastructure): union of the ranks' results == the single-process result, ownership
disjoint, destination of every row == (murmur3 row hash * V) >> 32.
Also here: concurrent groupby::aggregate calls on distinct objects and streams (the reference's threading model: any number of
host threads, each with its own stream - SURVEY.md section 8b).
"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_ranks(world, fn, timeout=300):
    """fn(rank, comm, stream) on one thread per rank of a fresh loopback world; returns the per-rank results."""
    import torch
    from cudf_amd import distributed as D
    comms = D.Communicator.loopback(world)
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            torch.cuda.set_device(0)
            out[r] = fn(r, comms[r], torch.cuda.Stream())
        except BaseException as e:  # noqa: BLE001 - re-raised on the main thread
            err[r] = e

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout)
    assert not any(t.is_alive() for t in threads), "a rank did not finish (a collective is waiting for a peer)"
    for e in err:
        if e is not None:
            raise e
    return out


def _shards(rng, world, rows, groups, empty_rank=None):
    """Ragged shards: rank r holds rows[r] rows of (int64 key, float64 value with ~10% nulls, int32 payload)."""
    shards = []
    for r in range(world):
        n = 0 if r == empty_rank else rows[r % len(rows)]
        k = rng.integers(-groups, groups, n, dtype=np.int64)
        v = rng.random(n)
        vv = rng.random(n) > 0.1
        p = rng.integers(0, 1 << 30, n).astype(np.int32)
        shards.append((k, v, vv, p))
    return shards


@pytest.mark.parametrize("world,limit", [(2, None), (4, None), (8, None), (4, 16 * 1024), (2, 16 * 1024)])
def test_shuffle_with_peers(gpu, oracle, world, limit):
    """Every rank receives exactly the rows it owns; a forced 16 KiB message limit walks the multi-round path; one rank holds no
    rows at all (empty slices to and from it); the nullable column travels with its validity."""
    import cudf_amd
    import gpu_backend as G
    from cudf_amd import distributed as D
    rng = np.random.default_rng(100 + world)
    shards = _shards(rng, world, [70_000, 31_111, 5, 120_000], 20_000, empty_rank=1 if world > 2 else None)
    tables = [cudf_amd.Table([G.to_device(k), G.to_device((v, vv)), G.to_device(p)]) for k, v, vv, p in shards]

    def rank_fn(r, comm, stream):
        if limit:
            comm.set_max_message_bytes(limit)
        mine = D.shuffle(comm, tables[r], [0], stream=stream)
        return [c.to_numpy() for c in mine.columns()], [c.null_count() for c in mine.columns()]

    res = _run_ranks(world, rank_fn)
    allk = np.concatenate([s[0] for s in shards])
    allv = np.concatenate([np.where(s[2], s[1], -1.0) for s in shards])
    allp = np.concatenate([s[3] for s in shards])
    got_rows = []
    for r, (cols, nulls) in enumerate(res):
        (k, _), (v, vv), (p, _) = cols
        if len(k):
            h = oracle.row_hash([k], 0).astype(np.uint64)
            dest = ((h * np.uint64(world)) >> np.uint64(32)).astype(np.int64)
            assert (dest == r).all(), f"rank {r} received a row it does not own"
            assert nulls[1] == int((~vv).sum()) if vv is not None else nulls[1] == 0
            vals = np.where(vv, v, -1.0) if vv is not None else v
            got_rows += list(zip(k.tolist(), vals.tolist(), p.tolist()))
    assert sorted(got_rows) == sorted(zip(allk.tolist(), allv.tolist(), allp.tolist()))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_shuffle_groupby_with_peers(gpu, oracle, world):
    """BASELINE config 5 inside the library with peers: union of the ranks' groups == the single-process groupby of all rows,
    no group on two ranks."""
    import cudf_amd
    import gpu_backend as G
    import kat
    from cudf_amd import aggregation as agg, distributed as D, groupby as gb
    from cudf_amd.types import NullPolicy
    rng = np.random.default_rng(200 + world)
    shards = _shards(rng, world, [150_000, 40_000, 99_999], 15_000, empty_rank=world - 1 if world > 2 else None)
    keys = [cudf_amd.Table([G.to_device(k)]) for k, _, _, _ in shards]
    vals = [G.to_device((v, vv)) for _, v, vv, _ in shards]

    def rank_fn(r, comm, stream):
        req = gb.GroupByRequest(vals[r], [agg.sum(), agg.count(NullPolicy.EXCLUDE), agg.max()])
        uk, out = D.shuffle_groupby(comm, keys[r], [req], stream=stream)
        return [G.from_device(c) for c in uk.columns()], [G.from_device(c) for c in out[0].columns()]

    res = _run_ranks(world, rank_fn)
    allk = np.concatenate([s[0] for s in shards])
    allv = np.concatenate([s[1] for s in shards])
    allvv = np.concatenate([s[2] for s in shards])
    exp = kat.sort_groups(*oracle.groupby([allk], [((allv, allvv), ["sum", "count_valid", "max"])]))
    seen = np.concatenate([kc[0][0] for kc, _ in res])
    assert len(np.unique(seen)) == len(seen), "a group came back from two ranks"
    for r, (kc, _) in enumerate(res):
        if len(kc[0][0]):
            h = oracle.row_hash([kc[0][0]], 0).astype(np.uint64)
            assert (((h * np.uint64(world)) >> np.uint64(32)).astype(np.int64) == r).all()

    def cat(cols):  # the ranks' result columns, one after the other
        data = np.concatenate([c[0] for c in cols])
        valid = None if all(c[1] is None for c in cols) else np.concatenate([np.ones(len(c[0]), bool) if c[1] is None else c[1] for c in cols])
        return (data, valid, cols[0][2])

    got = kat.sort_groups([cat([kc[0] for kc, _ in res])], [[cat([rc[j] for _, rc in res]) for j in range(3)]])
    kat.compare_columns(got[0][0], exp[0][0], "keys")
    for a, e, name in zip(got[1][0], exp[1][0], ["sum", "count", "max"]):
        kat.compare_columns(a, e, name, atol=kat.sum_atol(64, 1.0) if name == "sum" else 0.0)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("vt", ["float64", "int32"])
def test_combine_groupby_with_peers(gpu, oracle, world, vt):
    """cudf::distributed::combine_groupby (the decomposable form of config 5; reference streaming_groupby/merge.cu:91-144) with peers:
    local partials -> exchange of the partial groups -> merge -> MEAN finalised after the merge. Union of the ranks' groups == the
    single-process groupby of all rows for SUM / COUNT_VALID / COUNT_ALL / MIN / MAX / MEAN over a nullable value column (a key
    whose values are NULL on every rank comes back NULL for SUM / MIN / MAX / MEAN and 0 for COUNT_VALID), no group on two ranks."""
    import cudf_amd
    import gpu_backend as G
    import kat
    from oracle.oracle import HostColumn
    from cudf_amd import aggregation as agg, distributed as D, groupby as gb
    from cudf_amd.types import NullPolicy
    rng = np.random.default_rng(400 + world)
    shards = _shards(rng, world, [150_000, 40_000, 99_999], 15_000, empty_rank=world - 1 if world > 2 else None)
    names = ["sum", "count_valid", "count_all", "min", "max", "mean"]
    hv = []
    for k, v, vv, _ in shards:
        vv = vv & (k % 97 != 0)  # every value of some keys is NULL on every rank
        data = v if vt == "float64" else (v * 2000 - 1000).astype(np.int32)
        hv.append(HostColumn(data, vv, vt))
    keys = [cudf_amd.Table([G.to_device(k)]) for k, _, _, _ in shards]
    vals = [G.to_device(h) for h in hv]

    def rank_fn(r, comm, stream):
        req = gb.GroupByRequest(vals[r], [agg.sum(), agg.count(NullPolicy.EXCLUDE), agg.count(NullPolicy.INCLUDE), agg.min(), agg.max(), agg.mean()])
        uk, out = D.combine_groupby(comm, keys[r], [req], stream=stream)
        return [G.from_device(c) for c in uk.columns()], [G.from_device(c) for c in out[0].columns()]

    res = _run_ranks(world, rank_fn)
    allk = np.concatenate([s[0] for s in shards])
    allv = HostColumn(np.concatenate([h.data for h in hv]), np.concatenate([h.valid for h in hv]), vt)
    exp = kat.sort_groups(*oracle.groupby([allk], [(allv, names)]))
    seen = np.concatenate([kc[0][0] for kc, _ in res])
    assert len(np.unique(seen)) == len(seen), "a group came back from two ranks"
    for r, (kc, _) in enumerate(res):
        if len(kc[0][0]):
            h = oracle.row_hash([kc[0][0]], 0).astype(np.uint64)
            assert (((h * np.uint64(world)) >> np.uint64(32)).astype(np.int64) == r).all()

    def cat(cols):
        data = np.concatenate([c[0] for c in cols])
        valid = None if all(c[1] is None for c in cols) else np.concatenate([np.ones(len(c[0]), bool) if c[1] is None else c[1] for c in cols])
        return (data, valid, cols[0][2])

    got = kat.sort_groups([cat([kc[0] for kc, _ in res])], [[cat([rc[j] for _, rc in res]) for j in range(len(names))]])
    kat.compare_columns(got[0][0], exp[0][0], "keys")
    for a, e, name in zip(got[1][0], exp[1][0], names):
        kat.compare_columns(a, e, name, atol=kat.sum_atol(64, 1.0) if (name in ("sum", "mean") and vt == "float64") else (1e-9 if name == "mean" else 0.0))


def test_combine_groupby_rejects_what_does_not_decompose(gpu):
    """VARIANCE / ARGMAX do not merge by SUM / MIN / MAX: std::invalid_argument on every rank BEFORE the exchange (nobody is left waiting)."""
    import cudf_amd
    import gpu_backend as G
    from cudf_amd import aggregation as agg, distributed as D, groupby as gb
    k = [cudf_amd.Table([G.to_device(np.arange(100, dtype=np.int64) % 7)]) for _ in range(2)]
    v = [G.to_device(np.arange(100, dtype=np.float64)) for _ in range(2)]

    def rank_fn(r, comm, stream):
        for bad in (agg.variance(), agg.argmax()):
            with pytest.raises(ValueError):
                D.combine_groupby(comm, k[r], [gb.GroupByRequest(v[r], [agg.sum(), bad])], stream=stream)
        uk, out = D.combine_groupby(comm, k[r], [gb.GroupByRequest(v[r], [agg.sum()])], stream=stream)  # the communicator still works
        return uk.num_rows()

    assert sum(_run_ranks(2, rank_fn)) == 7


@pytest.mark.parametrize("world,limit", [(2, None), (4, 16 * 1024)])
def test_shuffle_join_with_peers(gpu, oracle, world, limit):
    """cudf::distributed::shuffle_join: every matching pair of the WHOLE tables comes back exactly once, as global row ids."""
    import cudf_amd
    import gpu_backend as G
    import kat
    from cudf_amd import distributed as D
    rng = np.random.default_rng(300 + world)
    left = [rng.integers(0, 30_000, n, dtype=np.int64) for n in [90_000, 10, 50_000, 0, 33_333, 7, 1, 20_000][:world]]
    right = [rng.integers(0, 30_000, n, dtype=np.int64) for n in [8_000, 12_000, 0, 3, 9_999, 1_000, 5, 2_000][:world]]
    lt = [cudf_amd.Table([G.to_device(k)]) for k in left]
    rt = [cudf_amd.Table([G.to_device(k)]) for k in right]

    def rank_fn(r, comm, stream):
        if limit:
            comm.set_max_message_bytes(limit)
        li, ri = D.shuffle_join(comm, lt[r], rt[r], stream=stream)
        return li.to_numpy()[0], ri.to_numpy()[0]

    res = _run_ranks(world, rank_fn)
    gl = np.concatenate([a for a, _ in res])
    gr = np.concatenate([b for _, b in res])
    el, er = oracle.join([np.concatenate(left)], [np.concatenate(right)], nulls_equal=True, kind="inner")
    assert kat.sorted_pairs(gl, gr) == kat.sorted_pairs(el, er)


def test_concurrent_aggregate_on_distinct_objects_and_streams(gpu, oracle):
    """Two host threads, each with its own groupby object and its own stream, aggregate at the same time (the first launches of
    several kernels race for their one-time attribute set-up: std::call_once); results against the oracle."""
    import torch
    import gpu_backend as G
    import kat
    import cudf_amd
    from cudf_amd import aggregation as agg, groupby as gb
    from cudf_amd.types import NullPolicy
    rng = np.random.default_rng(77)
    cases = []
    for groups, n in [(50, 300_000), (40_000, 400_000), (3, 250_000), (150_000, 350_000)]:
        k = rng.integers(0, groups, n, dtype=np.int64) * 1_000_003
        v = rng.random(n)
        cases.append((k, v, cudf_amd.Table([G.to_device(k)]), G.to_device(v)))
    out, err = [None] * len(cases), [None] * len(cases)
    start = threading.Barrier(len(cases))

    def body(i):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            start.wait()
            for _ in range(3):
                g = gb.GroupBy(cases[i][2])
                uk, res = g.aggregate([gb.GroupByRequest(cases[i][3], [agg.sum(), agg.count(NullPolicy.EXCLUDE), agg.min()])], stream=stream)
                out[i] = ([G.from_device(c) for c in uk.columns()], [G.from_device(c) for c in res[0].columns()])
        except BaseException as e:  # noqa: BLE001
            err[i] = e

    threads = [threading.Thread(target=body, args=(i,), daemon=True) for i in range(len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not any(t.is_alive() for t in threads)
    for e in err:
        if e is not None:
            raise e
    for (k, v, _, _), (kc, rc) in zip(cases, out):
        got = kat.sort_groups(kc, [rc])
        exp = kat.sort_groups(*oracle.groupby([k], [(v, ["sum", "count_valid", "min"])]))
        kat.compare_columns(got[0][0], exp[0][0], "keys")
        for a, e, name in zip(got[1][0], exp[1][0], ["sum", "count", "min"]):
            kat.compare_columns(a, e, name, atol=kat.sum_atol(200_000, 1.0) if name == "sum" else 0.0)


def test_result_read_back_on_the_producing_stream(gpu, oracle):
    """aggregate(stream=side stream) followed by to_numpy(): the host copy runs on the stream that produced the columns (a copy on
    the NULL stream does not order behind a non-blocking side stream and could read the output before k_finalize ran)."""
    import torch
    import gpu_backend as G
    import kat
    import cudf_amd
    from cudf_amd import aggregation as agg, groupby as gb
    from cudf_amd.types import NullPolicy
    rng = np.random.default_rng(91)
    n = 3_000_000
    k = rng.integers(0, 2_000, n, dtype=np.int64)
    v = rng.random(n)
    kt, vc = cudf_amd.Table([G.to_device(k)]), G.to_device(v)
    side = torch.cuda.Stream()
    exp = kat.sort_groups(*oracle.groupby([k], [(v, ["sum", "count_valid"])]))
    for _ in range(5):
        g = gb.GroupBy(kt)
        uk, res = g.aggregate([gb.GroupByRequest(vc, [agg.sum(), agg.count(NullPolicy.EXCLUDE)])], stream=side)
        assert uk.columns()[0].stream() is side
        got = kat.sort_groups([G.from_device(c) for c in uk.columns()], [[G.from_device(c) for c in res[0].columns()]])
        kat.compare_columns(got[0][0], exp[0][0], "keys")
        kat.compare_columns(got[1][0][1], exp[1][0][1], "count")
        kat.compare_columns(got[1][0][0], exp[1][0][0], "sum", atol=kat.sum_atol(3000, 1.0))
