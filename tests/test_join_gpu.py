"""GPU parity tests for cudf::inner_join / left_join / full_join and cudf::hash_join through the C ABI: the
reference's own KATs (tests/golden/kat_join.json), then seeded random inputs against the CPU oracle, then
size-independent properties at large sizes."""
import os

import numpy as np
import pytest

import kat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(gpu):
    import gpu_backend
    return gpu_backend


@pytest.mark.parametrize("c", kat.load("kat_join.json")["table_cases"], ids=lambda c: c["name"])
def test_join_table_kat(G, c):
    kat.run_join_table_case(G, c)


@pytest.mark.parametrize("c", kat.load("kat_join.json")["hash_join_cases"], ids=lambda c: c["name"])
def test_hash_join_kat(G, c):
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    right = cudf_amd.Table([G.to_device(x) for x in kat.table_cols(c["right"])])
    has_nulls = None if c["nullable"] is None else (c["nullable"] == "YES")
    kw = {} if c["load_factor"] is None else {"load_factor": c["load_factor"]}
    hj = HashJoin(right, NullEquality.EQUAL if c["nulls"] == "equal" else NullEquality.UNEQUAL, has_nulls=has_nulls, **kw)
    for p in c["probes"]:
        left = cudf_amd.Table([G.to_device(x) for x in kat.table_cols(p["left"])])
        size = getattr(hj, p["kind"] + "_join_size")(left)
        assert size == p["size"]
        li, ri = getattr(hj, p["kind"] + "_join")(left, output_size=size)
        assert kat.sorted_pairs(li.to_numpy()[0], ri.to_numpy()[0]) == kat.sorted_pairs(p["gold_left"], p["gold_right"])
        li, ri = getattr(hj, p["kind"] + "_join")(left)  # without the size hint
        assert kat.sorted_pairs(li.to_numpy()[0], ri.to_numpy()[0]) == kat.sorted_pairs(p["gold_left"], p["gold_right"])


@pytest.mark.parametrize("c", kat.load("kat_join.json")["match_context_cases"], ids=lambda c: c["name"])
def test_match_context_golden(G, c):
    """Every match-context vector of the reference (join_tests.cpp:2418-2651), from tests/golden/kat_join.json."""
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    lt, rt = kat.table_cols(c["left"]), kat.table_cols(c["right"])
    right = cudf_amd.Table([G.to_device(rt[i]) for i in c["on"]])
    left = cudf_amd.Table([G.to_device(lt[i]) for i in c["on"]])
    hj = HashJoin(right, NullEquality.EQUAL if c["nulls"] == "equal" else NullEquality.UNEQUAL)
    ctx = getattr(hj, f"{c['kind']}_join_match_context")(left)
    counts = ctx._match_counts.to_numpy()[0]
    assert counts.dtype == np.int32 and counts.tolist() == c["match_counts"]
    if c["size_equals_sum"]:
        assert getattr(hj, f"{c['kind']}_join_size")(left) == sum(c["match_counts"])


@pytest.mark.parametrize("c", kat.load("kat_join.json")["sorted_index_cases"], ids=lambda c: c["name"])
def test_sorted_index_golden(G, c):
    """HashJoinWithNullsOneSide (join_tests.cpp:2271-2378): sizes, then the two index columns sorted independently."""
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    right = cudf_amd.Table([G.to_device(x) for x in kat.table_cols(c["right"])])
    left = cudf_amd.Table([G.to_device(x) for x in kat.table_cols(c["left"])])
    hj = HashJoin(right, NullEquality.EQUAL if c["nulls"] == "equal" else NullEquality.UNEQUAL)
    for p in c["probes"]:
        size = getattr(hj, p["kind"] + "_join_size")(left)
        assert size == p["size"]
        li, ri = getattr(hj, p["kind"] + "_join")(left, output_size=size)
        assert sorted(li.to_numpy()[0].tolist()) == p["sorted_left"] and sorted(ri.to_numpy()[0].tolist()) == p["sorted_right"]


@pytest.mark.parametrize("c", kat.load("kat_join.json")["partitioned_cases"], ids=lambda c: c["name"])
def test_partitioned_golden(G, oracle, c):
    """The reference's partitioned-join tests (join_tests.cpp:3347-3709): partitioned_*_join over the stated row ranges, concatenated
    (a full join through finalize_partitioned_full_join), equals the whole join; an empty range yields nothing."""
    import cudf_amd
    from cudf_amd.join import HashJoin, JoinPartitionContext
    from cudf_amd.types import NullEquality
    left_all, right_all = kat.partitioned_case_tables(c)
    lcols, rcols = [left_all[i] for i in c["on"]], [right_all[i] for i in c["on"]]
    right = cudf_amd.Table([G.to_device(x) for x in rcols])
    left = cudf_amd.Table([G.to_device(x) for x in lcols])
    hj = HashJoin(right, NullEquality.EQUAL if c["nulls"] == "equal" else NullEquality.UNEQUAL)
    ctx = getattr(hj, f"{c['kind']}_join_match_context")(left)
    lparts, rparts = [], []
    for a, b in c["ranges"]:
        li, ri = getattr(hj, f"partitioned_{c['kind']}_join")(JoinPartitionContext(ctx, a, b))
        lparts.append(li)
        rparts.append(ri)
    if "expect_pairs" in c:
        assert sum(x.size() for x in lparts) == c["expect_pairs"]
    if c["kind"] == "full":
        fl, fr = HashJoin.finalize_partitioned_full_join(lparts, rparts, len(lcols[0].data), len(rcols[0].data))
        got_l, got_r = fl.to_numpy()[0], fr.to_numpy()[0]
    else:
        got_l = np.concatenate([x.to_numpy()[0] for x in lparts])
        got_r = np.concatenate([x.to_numpy()[0] for x in rparts])
    covered = sorted(x for a, b in c["ranges"] for x in range(a, b))
    if covered == list(range(len(lcols[0].data))):
        wl, wr = getattr(hj, f"{c['kind']}_join")(left)
        assert kat.sorted_pairs(got_l, got_r) == kat.sorted_pairs(wl.to_numpy()[0], wr.to_numpy()[0])
        el, er = oracle.join(lcols, rcols, nulls_equal=(c["nulls"] == "equal"), kind=c["kind"])
        assert kat.sorted_pairs(got_l, got_r) == kat.sorted_pairs(el, er)


def test_join_generated_kats(G):
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    z = cudf_amd.Table([G.to_device(np.zeros(65567, np.int32))])  # join_tests.cpp:2379-2394
    hj = HashJoin(z, NullEquality.UNEQUAL, has_nulls=False, load_factor=0.5)
    assert hj.inner_join_size(z) == 65567 * 65567
    a = np.array([1197], np.int32)
    with pytest.raises(ValueError):  # join_tests.cpp:432-450 -> std::invalid_argument
        G.join([a, a], [a, a, a])
    with pytest.raises(ValueError):
        G.join([], [a, a, a])
    with pytest.raises(TypeError):  # hash_join.cu:56-58 -> cudf::data_type_error
        G.join([a], [a.astype(np.int64)])
    with pytest.raises(ValueError):  # join_tests.cpp:347-367: load factor outside (0, 1]
        HashJoin(z, NullEquality.EQUAL, has_nulls=False, load_factor=0.0)
    with pytest.raises(ValueError):
        HashJoin(z, NullEquality.EQUAL, has_nulls=False, load_factor=1.5)
    # left table has nulls but the table was built without null check (hash_join.cu:52-54)
    hj2 = HashJoin(cudf_amd.Table([G.to_device(a)]), NullEquality.EQUAL, has_nulls=False)
    with pytest.raises(ValueError):
        hj2.inner_join(cudf_amd.Table([G.to_device((a, np.array([False])))]))


def _check(G, O, left, right, nulls_equal, kind):
    li, ri = G.join(left, right, nulls_equal=nulls_equal, kind=kind)
    el, er = O.join(left, right, nulls_equal=nulls_equal, kind=kind)
    assert len(li) == len(el)
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)


@pytest.mark.parametrize("kind", ["inner", "left", "full"])
@pytest.mark.parametrize("nulls_equal", [True, False])
def test_random_int64_with_nulls(G, oracle, kind, nulls_equal):
    """C3 shape at small size: int64 keys, 5% nulls on both sides, selectivity 0.3, duplicates on both sides."""
    rng = np.random.default_rng(44)
    nl, nr = 60_000, 7_000
    rk = rng.integers(0, 5_000, nr, dtype=np.int64)
    lk = np.where(rng.random(nl) < 0.3, rng.integers(0, 5_000, nl), rng.integers(5_000, 10_000, nl)).astype(np.int64)
    lv, rv = rng.random(nl) > 0.05, rng.random(nr) > 0.05
    _check(G, oracle, [(lk, lv)], [(rk, rv)], nulls_equal, kind)


@pytest.mark.parametrize("kind", ["inner", "left", "full"])
def test_random_no_nulls_fast_path(G, oracle, kind):
    rng = np.random.default_rng(12345)
    nl, nr = 200_000, 20_000
    rk = rng.permutation(40_000)[:nr].astype(np.int64)  # unique build keys
    lk = rng.integers(0, 80_000, nl, dtype=np.int64)
    _check(G, oracle, [lk], [rk], True, kind)
    _check(G, oracle, [rk], [lk], True, kind)  # right bigger than left: inner join builds on left and swaps


@pytest.mark.parametrize("nulls_equal", [True, False])
def test_random_multi_column_mixed_types(G, oracle, nulls_equal):
    from oracle.oracle import HostColumn
    rng = np.random.default_rng(7)
    nl, nr = 30_000, 9_000
    l0, r0 = rng.integers(0, 50, nl, dtype=np.int32), rng.integers(0, 50, nr, dtype=np.int32)
    l1 = rng.integers(0, 4, nl).astype(np.float64) * 0.5
    r1 = rng.integers(0, 4, nr).astype(np.float64) * 0.5
    l1[::97] = np.nan
    r1[::89] = np.nan
    l1[1::97] = -0.0
    r1[1::89] = 0.0
    l2, r2 = rng.integers(0, 2, nl).astype(np.uint8), rng.integers(0, 2, nr).astype(np.uint8)
    lv, rv = rng.random(nl) > 0.1, rng.random(nr) > 0.1
    left = [HostColumn(l0, lv, "int32"), HostColumn(l1, None, "float64"), HostColumn(l2, None, "bool")]
    right = [HostColumn(r0, rv, "int32"), HostColumn(r1, None, "float64"), HostColumn(r2, None, "bool")]
    for kind in ("inner", "left", "full"):
        _check(G, oracle, left, right, nulls_equal, kind)


@pytest.mark.parametrize("kind", ["inner", "left", "full"])
@pytest.mark.parametrize("two_columns", [False, True])
def test_heavily_duplicated_build_key(G, oracle, kind, two_columns):
    """One build key with 40,000 duplicates among 120,000 build rows: its entries fill the linear window of its probe
    sequence and continue on the key's own stride trail (kernels.hip seq_next); build, count and retrieve passes must
    walk the same sequence, neighbours of the heavy key's home slot must still be found, and the inserts must share
    their skip hints. Checked against the oracle, with a time bound (the quadratic walk took seconds)."""
    import time
    rng = np.random.default_rng(77)
    nr, nl = 120_000, 90_000
    rk = rng.permutation(400_000)[:nr].astype(np.int64)
    rk[rng.permutation(nr)[:40_000]] = 31337
    rk[rng.permutation(nr)[:3_000]] = 4242          # a second, lighter heavy key
    lk = rng.integers(0, 400_000, nl, dtype=np.int64)
    lk[:25] = 31337                                  # 25 x 40,000 pairs
    lk[25:40] = 4242
    left, right = [lk], [rk]
    if two_columns:  # generic (multi-column) row path
        left.append((lk % 7).astype(np.int32))
        right.append((rk % 7).astype(np.int32))
    t0 = time.perf_counter()
    _check(G, oracle, left, right, True, kind)
    assert time.perf_counter() - t0 < 30.0


def test_empty_sides(G, oracle):
    e = np.zeros(0, np.int64)
    k = np.array([1, 2, 3], np.int64)
    for kind in ("inner", "left", "full"):
        _check(G, oracle, [e], [k], True, kind)
        _check(G, oracle, [k], [e], True, kind)
        _check(G, oracle, [e], [e], True, kind)


def test_sliced_key_columns(G, oracle):
    from oracle.oracle import HostColumn
    rng = np.random.default_rng(5)
    lk, rk = rng.integers(0, 100, 5000, dtype=np.int64), rng.integers(0, 100, 700, dtype=np.int64)
    lv, rv = rng.random(5000) > 0.1, rng.random(700) > 0.1
    _check(G, oracle, [HostColumn(lk, lv, "int64", offset=13)], [HostColumn(rk, rv, "int64", offset=5)], False, "inner")
    _check(G, oracle, [HostColumn(lk, lv, "int64", offset=13)], [HostColumn(rk, rv, "int64", offset=5)], True, "full")


def test_large_properties(G):
    """50M x 5M inner join (unique build keys): pair count == number of probe keys present in the build side;
    every pair joins equal keys; every probe row appears at most once."""
    import torch
    import cudf_amd
    from cudf_amd import join as J
    nl, nr = 50_000_000, 5_000_000
    g = torch.Generator(device="cuda").manual_seed(12345)
    rk = torch.randperm(2 * nr, generator=g, device="cuda")[:nr].to(torch.int64)
    lk = torch.randint(0, 4 * nr, (nl,), generator=g, device="cuda", dtype=torch.int64)
    li, ri = J.inner_join(cudf_amd.Table([cudf_amd.Column.from_torch(lk)]), cudf_amd.Table([cudf_amd.Column.from_torch(rk)]))
    li_t = torch.from_numpy(li.to_numpy()[0]).cuda().long()
    ri_t = torch.from_numpy(ri.to_numpy()[0]).cuda().long()
    present = torch.zeros(4 * nr, dtype=torch.bool, device="cuda")
    present[rk] = True
    assert li_t.numel() == int(present[lk].sum())
    assert bool((lk[li_t] == rk[ri_t]).all())
    assert torch.unique(li_t).numel() == li_t.numel()


def test_hash_join_object_reuse(G, oracle):
    """One hash_join object, several probes of different sizes and the size API."""
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    rng = np.random.default_rng(101)
    rk = rng.integers(0, 30_000, 40_000, dtype=np.int64)
    hj = HashJoin(cudf_amd.Table([G.to_device(rk)]), NullEquality.EQUAL)
    for nl in (5, 70_000, 300_000):
        lk = rng.integers(0, 60_000, nl, dtype=np.int64)
        t = cudf_amd.Table([G.to_device(lk)])
        li, ri = hj.inner_join(t)
        el, er = oracle.join([lk], [rk], nulls_equal=True, kind="inner")
        assert hj.inner_join_size(t) == len(el)
        assert kat.sorted_pairs(li.to_numpy()[0], ri.to_numpy()[0]) == kat.sorted_pairs(el, er)


# ---------------------------------------------------------------- match contexts / partitioned probes
_T0 = [(np.array([3, 1, 2, 0, 2], np.int32), None), (np.array([1, 1, 0, 4, 0], np.int32), np.array([1, 1, 0, 1, 1], bool))]
_T1 = [(np.array([2, 2, 0, 4, 3], np.int32), None), (np.array([1, 0, 1, 2, 1], np.int32), np.array([1, 0, 1, 1, 1], bool))]
# reference join_tests.cpp:2418-2482 (inner), :2484-2534 (left); the string key column is encoded as int32 codes
_MATCH_KATS = [
    ("inner", [0], True, [1, 0, 2, 1, 2]), ("inner", [0, 1], True, [1, 0, 1, 0, 0]),
    ("inner", [0], False, [1, 0, 2, 1, 2]), ("inner", [0, 1], False, [1, 0, 0, 0, 0]),
    ("left", [0], True, [1, 1, 2, 1, 2]), ("left", [0, 1], True, [1, 1, 1, 1, 1]), ("left", [0, 1], False, [1, 1, 1, 1, 1]),
    ("full", [0], True, [1, 1, 2, 1, 2]),
]


def _hash_join(G, right_cols, nulls_equal):
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    t = cudf_amd.Table([G.to_device(c if c[1] is not None else c[0]) for c in right_cols])
    return HashJoin(t, NullEquality.EQUAL if nulls_equal else NullEquality.UNEQUAL), t


@pytest.mark.parametrize("kind,on,nulls_equal,expect", _MATCH_KATS)
def test_match_context_kat(G, kind, on, nulls_equal, expect):
    import cudf_amd
    hj, _keep = _hash_join(G, [_T1[i] for i in on], nulls_equal)
    left = cudf_amd.Table([G.to_device(_T0[i] if _T0[i][1] is not None else _T0[i][0]) for i in on])
    ctx = getattr(hj, f"{kind}_join_match_context")(left)
    counts = ctx._match_counts.to_numpy()[0]
    assert counts.dtype == np.int32 and counts.tolist() == expect
    assert int(counts.sum()) == getattr(hj, f"{kind}_join_size")(left) or kind == "full"


@pytest.mark.parametrize("kind", ["inner", "left", "full"])
@pytest.mark.parametrize("nulls_equal", [True, False])
def test_partitioned_join_equals_whole_join(G, oracle, kind, nulls_equal):
    """Chunked probing (reference join_tests.cpp:1120-1180 shape): the union of partitioned_*_join over row ranges of
    the left table equals the whole join; per-range sizes equal the sums of the match counts; left indices refer to
    the complete left table. Full join: finalize_partitioned_full_join appends the unmatched right rows."""
    import cudf_amd
    from cudf_amd.join import HashJoin, JoinPartitionContext
    rng = np.random.default_rng(77)
    nl, nr = 50_000, 6_000
    rk = rng.integers(0, 4_000, nr, dtype=np.int64)
    lk = rng.integers(0, 8_000, nl, dtype=np.int64)
    lv, rv = rng.random(nl) > 0.05, rng.random(nr) > 0.05
    hj, _keep = _hash_join(G, [(rk, rv)], nulls_equal)
    left = cudf_amd.Table([G.to_device((lk, lv))])
    ctx = getattr(hj, f"{kind}_join_match_context")(left)
    counts = ctx._match_counts.to_numpy()[0]
    bounds = [0, 1, 7, 20_000, 20_000, 49_999, nl]  # includes an empty and two single-row ranges
    ls, rs, lparts, rparts = [], [], [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        li, ri = getattr(hj, f"partitioned_{kind}_join")(JoinPartitionContext(ctx, a, b))
        lparts.append(li)
        rparts.append(ri)
        l, r = li.to_numpy()[0], ri.to_numpy()[0]
        assert len(l) == int(counts[a:b].sum())
        assert len(l) == 0 or (l.min() >= a and l.max() < b)
        ls.append(l)
        rs.append(r)
    if kind == "full":
        fl, fr = HashJoin.finalize_partitioned_full_join(lparts, rparts, nl, nr)
        got_l, got_r = fl.to_numpy()[0], fr.to_numpy()[0]
    else:
        got_l, got_r = np.concatenate(ls), np.concatenate(rs)
    el, er = oracle.join([(lk, lv)], [(rk, rv)], nulls_equal=nulls_equal, kind=kind)
    assert kat.sorted_pairs(got_l, got_r) == kat.sorted_pairs(el, er)


def test_partitioned_join_invalid_context(G):
    import cudf_amd
    from cudf_amd.join import JoinMatchContext, JoinPartitionContext
    hj, _keep = _hash_join(G, [(np.array([1, 2, 3], np.int64), None)], True)
    left = cudf_amd.Table([G.to_device(np.array([1, 2], np.int64))])
    ctx = hj.inner_join_match_context(left)
    for a, b in ((-1, 1), (1, 0), (0, 3)):  # out of bounds (reference hash_join.hpp:341-343)
        with pytest.raises(ValueError):
            hj.partitioned_inner_join(JoinPartitionContext(ctx, a, b))
    with pytest.raises(ValueError):
        hj.partitioned_inner_join(JoinPartitionContext(JoinMatchContext(left, None), 0, 1))


_JOIN_FUZZ_TYPES = ["int8", "int16", "int32", "int64", "uint32", "uint64", "float32", "float64", "bool"]


@pytest.mark.parametrize("seed", range(int(os.environ.get("CUDF_AMD_FUZZ_SEEDS", "60"))))
def test_fuzz_joins_against_oracle(G, oracle, seed, monkeypatch):
    """Seeded random joins: 1-3 key columns of mixed types, nulls on either side, both null equalities, all three kinds,
    duplicate keys on both sides, sliced inputs, empty sides."""
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(5000 + seed)
    if seed % 4 in (1, 2):
        # the round-3 paths at fuzz sizes: whenever a case is ONE integer key column of 4 or 8 bytes whose NULLs never match, it goes
        # through the LDS radix join (seed % 4 == 1: dense tables off) or the dense table's ordered / partitioned / one-pass-left
        # probes (seed % 4 == 2), sliced columns and validity offsets included
        for k, v in (("RADIX_MIN_BUILD", "0"), ("RADIX_MIN_PROBE", "0"), ("DENSE_MIN_ROWS", "1"), ("DENSE_ORDERED_MIN_PROBE", "0"),
                     ("DENSE_PART_MIN_BUILD", "0"), ("DENSE_PART_MIN_RANGE", "0"), ("DENSE_PART_MIN_PROBE", "0"), ("DENSE_PART_SLICE_LOG2", "8")):
            monkeypatch.setenv("CUDF_AMD_JOIN_" + k, v)
        if seed % 4 == 1:
            monkeypatch.setenv("CUDF_AMD_JOIN_DENSE", "0")
        else:
            monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_PROBE", str(1 + (seed // 4) % 2))
    nl = int(rng.choice([0, 1, 50, 3_000, 120_000]))
    nr = int(rng.choice([0, 1, 40, 2_000, 30_000]))
    ncols = int(rng.integers(1, 4))
    types = [str(rng.choice(_JOIN_FUZZ_TYPES)) for _ in range(ncols)]
    spread = int(rng.choice([4, 60, 5000]))
    # keep the result enumerable on the host: about nl * nr / (distinct keys) pairs
    while nl * nr / float(np.prod([2 if t == "bool" else spread for t in types])) > 2e6:
        if all(t == "bool" for t in types):  # the key space cannot grow: shrink the left side instead
            nl //= 4
        else:
            spread *= 4

    def side(n):
        cols = []
        for t in types:
            npt = NP_OF_TYPE_ID[TYPE_ID[t]]
            off = int(rng.choice([0, 0, 4]))
            m = n + off
            if t == "bool":
                data = rng.integers(0, 2, m).astype(npt)
            elif np.dtype(npt).kind == "f":
                data = (rng.integers(0, spread, m) - spread // 2).astype(npt) * npt(0.5)
                if m:
                    data[rng.integers(0, m, max(1, m // 50))] = npt("nan")
            else:
                data = rng.integers(0, min(spread, int(np.iinfo(npt).max) - 1), m).astype(npt)
            valid = (rng.random(m) > 0.1) if rng.random() < 0.4 else None
            cols.append(HostColumn(data, valid, t, offset=off) if off else HostColumn(data, valid, t))
        return cols

    left, right = side(nl), side(nr)
    nulls_equal = bool(rng.random() < 0.5)
    kind = str(rng.choice(["inner", "left", "full"]))
    _check(G, oracle, left, right, nulls_equal, kind)


@pytest.mark.parametrize("kind", ["inner", "left", "full"])
@pytest.mark.parametrize("shape", ["unique", "duplicates", "negative", "nulls"])
def test_dense_key_join_against_oracle(G, oracle, monkeypatch, kind, shape):
    """One int64 key column whose build-side values span a small range: the direct-address table (head + chain through the
    build rows) replaces the hash table. Unique and repeated build keys, negative keys, NULLs that never match (UNEQUAL),
    probe keys outside the build side's range, all three join kinds; the same call with CUDF_AMD_JOIN_DENSE=0 agrees."""
    rng = np.random.default_rng(91)
    nl, nr = 120_000, 70_000
    lo = -35_000 if shape == "negative" else 1_000_000
    if shape == "duplicates":
        rk = rng.integers(0, 9_000, nr, dtype=np.int64) + lo  # ~8 build rows per key
    else:
        rk = rng.permutation(nr).astype(np.int64) + lo
    lk = rng.integers(-20_000, nr + 40_000, nl, dtype=np.int64) + lo  # a third of the probe keys lie outside the range
    lv = rv = None
    if shape == "nulls":
        lv, rv = rng.random(nl) > 0.1, rng.random(nr) > 0.1
    left, right = [(lk, lv)] if lv is not None else [lk], [(rk, rv)] if rv is not None else [rk]
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_MIN_ROWS", "1")
    li, ri = G.join(left, right, nulls_equal=False, kind=kind)
    el, er = oracle.join(left, right, nulls_equal=False, kind=kind)
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE", "0")
    li2, ri2 = G.join(left, right, nulls_equal=False, kind=kind)
    assert kat.sorted_pairs(li2, ri2) == kat.sorted_pairs(el, er)


@pytest.mark.parametrize("kind", ["inner", "left", "full"])
def test_dense_key_join_hot_build_key(G, oracle, monkeypatch, kind):
    """A build key with thousands of rows that hundreds of probe rows hit: the keys' ROW LISTS (offsets + rows grouped by key,
    launch_dense_csr) let the whole wave emit a hot key's pairs 64 at a time; a chain walked link by link by one lane took
    149 ms for 1000 probes of a key with 100,000 build rows (profiles/r2_join_matrix.txt). Lists of 1, 2, 64, 65 and 5000 rows;
    NULL build rows never match; the match counts of the join_match_context agree."""
    import cudf_amd
    from cudf_amd import join as J
    from cudf_amd.types import NullEquality
    rng = np.random.default_rng(93)
    nr, nl = 60_000, 40_000
    rk = rng.permutation(nr).astype(np.int64) + 500
    rk[:5000] = 777            # the hot key
    rk[5000:5064] = 1_111      # a list of exactly 64 rows (+ possibly the permutation's own 1111)
    rk[5064:5129] = 2_222      # 65 rows
    rk[5129:5131] = 3_333
    rv = rng.random(nr) > 0.05
    lk = rng.integers(0, nr + 20_000, nl, dtype=np.int64)
    lk[:300] = 777
    lk[300:340] = 1_111
    lk[340:380] = 2_222
    lk[380:400] = 3_333
    lv = rng.random(nl) > 0.05
    left, right = [(lk, lv)], [(rk, rv)]
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_MIN_ROWS", "1")
    li, ri = G.join(left, right, nulls_equal=False, kind=kind)
    el, er = oracle.join(left, right, nulls_equal=False, kind=kind)
    assert len(li) == len(el)
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)
    if kind == "inner":
        # (cudf::inner_join builds on the smaller side: probe the hot BUILD key through a hash_join object as well)
        hj = J.HashJoin(cudf_amd.Table([G.to_device((rk, rv))]), NullEquality.UNEQUAL)
        pl, pr = hj.inner_join(cudf_amd.Table([G.to_device((lk, lv))]))
        assert kat.sorted_pairs(pl.to_numpy()[0], pr.to_numpy()[0]) == kat.sorted_pairs(el, er)
        hj = J.HashJoin(cudf_amd.Table([G.to_device((rk, rv))]), NullEquality.UNEQUAL)
        ctx = hj.inner_join_match_context(cudf_amd.Table([G.to_device((lk, lv))]))
        counts = ctx._match_counts.to_numpy()[0]
        assert np.array_equal(counts, np.bincount(el, minlength=nl))


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_dense_joins_against_oracle(G, oracle, monkeypatch, seed):
    """Seeded shapes on the dense path (CUDF_AMD_JOIN_DENSE_MIN_ROWS=1): sizes 0 .. 60K, key ranges from a handful of values to
    sparse (where the planner must fall back to the hash table), heavy duplication, nulls under both null equalities."""
    rng = np.random.default_rng(7000 + seed)
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_MIN_ROWS", "1")
    nl, nr = int(rng.choice([0, 1, 300, 60_000])), int(rng.choice([1, 17, 5_000, 40_000]))
    spread = int(rng.choice([3, 500, 50_000, 10**12]))
    while nl * nr / spread > 2e6:  # keep the result enumerable on the host
        nl, nr = max(1, nl // 2), max(1, nr // 2)
    base = int(rng.choice([0, -10**9, 2**62]))
    rk = rng.integers(0, spread, nr, dtype=np.int64) + base
    lk = rng.integers(0, max(2, int(spread * 1.3)), nl, dtype=np.int64) + base - int(0.1 * min(spread, 10**6))
    lv = (rng.random(nl) > 0.2) if rng.random() < 0.5 else None
    rv = (rng.random(nr) > 0.2) if rng.random() < 0.5 else None
    left, right = [(lk, lv)] if lv is not None else [lk], [(rk, rv)] if rv is not None else [rk]
    for kind in ("inner", "left", "full"):
        for eq in (False, True):
            li, ri = G.join(left, right, nulls_equal=eq, kind=kind)
            el, er = oracle.join(left, right, nulls_equal=eq, kind=kind)
            assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er), (kind, eq)


# ---------------------------------------------------------------- LDS radix join (big sparse single-int64-key inner joins)
_RJ_EMPTY = np.array([0x7f4a7c159e3779b9], dtype=np.uint64).view(np.int64)[0]  # the radix join's empty-slot marker (radix_kernels.hip)


@pytest.fixture
def force_radix_join(monkeypatch):
    """Makes small inputs take the LDS radix join (engine.hpp radix_join_args) that big sparse-key inner joins use."""
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE", "0")
    monkeypatch.setenv("CUDF_AMD_JOIN_RADIX_MIN_BUILD", "0")
    monkeypatch.setenv("CUDF_AMD_JOIN_RADIX_MIN_PROBE", "0")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("kind", ["inner", "left"])
def test_radix_join_takes_float_keys(G, oracle, force_radix_join, dtype, kind):
    """One FLOAT key column goes through the LDS radix join as well (round 4): the scatter's key is the value's NORMALISED bits, so
    -0.0 meets +0.0 and every NaN meets every NaN, as the row equality of the table probes has it (reference
    row_operator/equality.cuh:59-89)."""
    rng = np.random.default_rng(5 if dtype == "float64" else 6)
    nl, nr = 300_000, 40_000
    npt = np.dtype(dtype)
    pool = (rng.integers(-30_000, 30_000, 50_000) / 8.0).astype(npt)
    pool[:6] = [0.0, -0.0, np.nan, np.inf, -np.inf, npt.type(1e-30)]
    rk = pool[rng.integers(0, len(pool), nr)]
    rk[:3] = [-0.0, np.nan, 0.0]
    lk = pool[rng.integers(0, len(pool), nl)]
    lk[:4] = [0.0, np.array(np.nan).astype(npt).view(np.uint32 if dtype == "float32" else np.uint64).__or__(1).view(npt), -0.0, np.nan]  # a second NaN pattern
    lk = np.concatenate([lk, (rng.integers(40_000, 90_000, 50_000) / 8.0).astype(npt)])  # keys the build side does not hold
    (li, ri), kernels = _kernels_of(lambda: G.join([lk], [rk], nulls_equal=True, kind=kind))
    assert kernels.get("join_partition", 0) >= 2, kernels  # (both sides were radix-partitioned)
    el, er = oracle.join([lk], [rk], nulls_equal=True, kind=kind)
    assert len(li) == len(el)
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)


@pytest.mark.parametrize("kind", ["inner", "left", "full"])
@pytest.mark.parametrize("shape", ["narrow_ranges", "signed_narrow_first", "one_value_columns", "wide_second_column"])
def test_two_integer_key_columns_in_one_word(G, oracle, force_radix_join, shape, kind):
    """Two INTEGER key columns whose build-side ranges fit 63 bits together go through the single-word radix join as
    (c0 - lo0) << bits1 | (c1 - lo1) (round 4, join.cpp decide_packed_words): probe rows below / above the build side's value box
    in either column join nothing (a left / full join keeps them); ranges that do not fit stay on the two-word key."""
    from oracle.oracle import HostColumn
    rng = np.random.default_rng({"narrow_ranges": 1, "signed_narrow_first": 2, "one_value_columns": 3, "wide_second_column": 4}[shape])
    nl, nr = 300_000, 60_000
    t0, t1 = {"narrow_ranges": ("int64", "int32"), "signed_narrow_first": ("int32", "int64"), "one_value_columns": ("uint32", "int64"),
              "wide_second_column": ("int64", "int64")}[shape]
    if shape == "narrow_ranges":
        r0, r1 = rng.integers(10**12, 10**12 + 5000, nr), rng.integers(-300, 300, nr)
        l0, l1 = rng.integers(10**12 - 500, 10**12 + 5500, nl), rng.integers(-350, 350, nl)   # outside the box on all four sides
    elif shape == "signed_narrow_first":
        r0, r1 = rng.integers(-2**31, -2**31 + 700, nr), rng.integers(-2**40, 2**40, nr) // 2**24 * 2**24
        l0, l1 = rng.integers(-2**31, -2**31 + 800, nl), rng.integers(-2**40, 2**40, nl) // 2**24 * 2**24
        l1[:100] = np.iinfo(np.int64).min   # the difference to the box's corner wraps around
        l1[100:200] = np.iinfo(np.int64).max
    elif shape == "one_value_columns":
        r0, r1 = np.full(nr, 4_000_000_000), rng.integers(0, 30_000, nr)   # a column of one value takes no bits
        l0, l1 = np.where(rng.random(nl) < 0.1, 7, 4_000_000_000), rng.integers(0, 40_000, nl)
    else:
        r0, r1 = rng.integers(0, 1000, nr), rng.integers(-2**62, 2**62, nr)   # 10 + 63 bits: two words
        pick = rng.integers(0, nr, nl)
        l0, l1 = np.where(rng.random(nl) < 0.5, r0[pick], r0[pick] + 1), r1[pick]
    right = [HostColumn(r0.astype(np.dtype(t0)), None, t0), HostColumn(r1.astype(np.dtype(t1)), None, t1)]
    left = [HostColumn(l0.astype(np.dtype(t0)), None, t0), HostColumn(l1.astype(np.dtype(t1)), None, t1)]
    (li, rj), kernels = _kernels_of(lambda: G.join(left, right, nulls_equal=True, kind=kind))
    assert kernels.get("join_partition", 0) >= 2, kernels  # (both sides were radix-partitioned)
    el, er = oracle.join(left, right, nulls_equal=True, kind=kind)
    assert len(li) == len(el)
    assert kat.sorted_pairs(li, rj) == kat.sorted_pairs(el, er)


@pytest.mark.parametrize("pack", ["one_word_if_it_fits", "two_words"])
@pytest.mark.parametrize("kind", ["inner", "left"])
@pytest.mark.parametrize("shape", ["i64_i64", "i64_i32", "f64_i64", "u32_f32", "low_cardinality_first", "nulls_unequal"])
def test_radix_join_takes_two_key_columns(G, oracle, force_radix_join, monkeypatch, shape, kind, pack):
    """TWO key columns of 4 or 8 bytes each go through the LDS radix join as a two-word key (round 4): the partition digit and the LDS
    table's slot state are a 64-bit fold of the two words, every candidate is verified against the build record's two words. A first
    column with a handful of distinct values (all the identity is in the second), duplicated tuples on both sides, 5 % NULLs in either
    column (UNEQUAL: such rows join nothing; a left join keeps them... on the table path: the radix left join leaves NULLs to it)."""
    from oracle.oracle import HostColumn
    if pack == "two_words":
        monkeypatch.setenv("CUDF_AMD_JOIN_PACK_RANGE", "0")  # (integer columns of small ranges would travel as one word)
    rng = np.random.default_rng({"i64_i64": 1, "i64_i32": 2, "f64_i64": 3, "u32_f32": 4, "low_cardinality_first": 5, "nulls_unequal": 6}[shape])
    nl, nr = 260_000, 50_000
    t0, t1 = {"i64_i64": ("int64", "int64"), "i64_i32": ("int64", "int32"), "f64_i64": ("float64", "int64"), "u32_f32": ("uint32", "float32"),
              "low_cardinality_first": ("int64", "int64"), "nulls_unequal": ("int64", "int32")}[shape]
    card0 = 5 if shape == "low_cardinality_first" else 3_000

    def col(n, card, t):
        v = rng.integers(0, card, n)
        if t.startswith("float"):
            v = v / 4.0 - 100.0
        elif t.startswith("int"):
            v = v - card // 2
        return v.astype(np.dtype(t))
    pool0, pool1 = col(40_000, card0, t0), col(40_000, 20_000, t1)          # 40K tuples, some of them equal
    ri, li_ = rng.integers(0, 30_000, nr), rng.integers(0, 40_000, nl)       # the probe side also draws tuples the build side lacks
    right = [HostColumn(pool0[ri], None, t0), HostColumn(pool1[ri], None, t1)]
    left = [HostColumn(pool0[li_], None, t0), HostColumn(pool1[li_], None, t1)]
    nulls_equal = True
    if shape == "nulls_unequal":
        nulls_equal = False
        right = [HostColumn(c.data, rng.random(nr) > 0.05, t) for c, t in zip(right, (t0, t1))]
        if kind == "inner":
            left = [HostColumn(c.data, rng.random(nl) > 0.05, t) for c, t in zip(left, (t0, t1))]
    (li, rj), kernels = _kernels_of(lambda: G.join(left, right, nulls_equal=nulls_equal, kind=kind))
    assert kernels.get("join_partition", 0) >= 2, kernels  # (both sides were radix-partitioned)
    el, er = oracle.join(left, right, nulls_equal=nulls_equal, kind=kind)
    assert len(li) == len(el)
    assert kat.sorted_pairs(li, rj) == kat.sorted_pairs(el, er)


def _kernels_of(fn):
    from cudf_amd import _lib
    _lib.profile_reset()
    _lib.profile_enable(True)
    try:
        out = fn()
    finally:
        _lib.profile_enable(False)
    return out, {k: v[0] for k, v in _lib.profile_report().items()}  # {kernel: launches}


@pytest.mark.parametrize("kind", ["inner", "left"])
@pytest.mark.parametrize("shape", ["unique", "duplicates", "nulls", "marker_key", "four_pairs_per_row"])
def test_radix_join_matches_oracle(G, oracle, force_radix_join, shape, kind):
    """Both sides through the two scatter levels into 2048 LDS-sized partitions, then the per-partition LDS table: unique build
    keys; duplicates on both sides (several pairs per probe row); 5 % NULLs with null_equality::UNEQUAL; rows whose key is the
    table's empty-slot marker (they take the side list); every probe row matching four build rows (more pairs than the
    partitions' stages hold: the retrieve pass joins those partitions again)."""
    rng = np.random.default_rng({"unique": 1, "duplicates": 2, "nulls": 3, "marker_key": 4, "four_pairs_per_row": 5}[shape])
    nl, nr = 700_000, 90_000
    if shape == "four_pairs_per_row":
        nr = 80_000
        rk = np.repeat(np.arange(20_000, dtype=np.int64), 4) * 1_000_003
        lk = rng.integers(0, 20_000, nl, dtype=np.int64) * 1_000_003
    elif shape == "unique":
        rk = rng.permutation(400_000)[:nr].astype(np.int64) * 1_000_003
        lk = rng.integers(0, 800_000, nl, dtype=np.int64) * 1_000_003
    else:
        rk = rng.integers(0, 60_000, nr, dtype=np.int64) * 1_000_003 - 7
        lk = np.where(rng.random(nl) < 0.3, rng.integers(0, 60_000, nl), rng.integers(60_000, 200_000, nl)).astype(np.int64) * 1_000_003 - 7
    left, right = [lk], [rk]
    nulls_equal = True
    if shape == "nulls":
        left, right, nulls_equal = [(lk, rng.random(nl) > 0.05)], [(rk, rng.random(nr) > 0.05)], False
    if shape == "marker_key":
        rk[[5, 77, 4000]] = _RJ_EMPTY
        lk[rng.integers(0, nl, 50)] = _RJ_EMPTY
    # (left joins: a probe record without a partner yields {row, JoinNoMatch} inside its partition, the probe rows with a NULL key
    # get theirs from k_radix_null_rows)
    (li, ri), kernels = _kernels_of(lambda: G.join(left, right, nulls_equal=nulls_equal, kind=kind))
    assert kernels.get("join_partition_level2") == 2, kernels
    el, er = oracle.join(left, right, nulls_equal=nulls_equal, kind=kind)
    assert len(li) == len(el)
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)


def test_radix_join_object_other_kinds_and_sizes(G, oracle, force_radix_join):
    """A hash_join object whose build side took the radix partitions: inner joins go through them; left / full joins, the size
    API and the match context build the open-addressing table on first need and agree with the oracle."""
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    rng = np.random.default_rng(202)
    rk = rng.integers(0, 30_000, 40_000, dtype=np.int64) * 1_000_003
    hj = HashJoin(cudf_amd.Table([G.to_device(rk)]), NullEquality.EQUAL)
    for nl in (5, 70_000, 300_000):
        lk = rng.integers(0, 60_000, nl, dtype=np.int64) * 1_000_003
        t = cudf_amd.Table([G.to_device(lk)])
        (li, ri), kernels = _kernels_of(lambda: hj.inner_join(t))
        assert "join_partition_level2" in kernels
        el, er = oracle.join([lk], [rk], nulls_equal=True, kind="inner")
        assert kat.sorted_pairs(li.to_numpy()[0], ri.to_numpy()[0]) == kat.sorted_pairs(el, er)
        assert hj.inner_join_size(t) == len(el)
        for kind in ("left", "full"):
            gl, gr = getattr(hj, kind + "_join")(t)
            xl, xr = oracle.join([lk], [rk], nulls_equal=True, kind=kind)
            assert kat.sorted_pairs(gl.to_numpy()[0], gr.to_numpy()[0]) == kat.sorted_pairs(xl, xr)


def test_radix_join_hot_key_falls_back(G, oracle, force_radix_join):
    """A build key with 20,000 rows: its partition does not fit the LDS table, the constructor keeps the open-addressing table.
    A probe side whose rows mostly carry ONE key overflows that partition's regions: that call probes the table instead."""
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    rng = np.random.default_rng(203)
    nl, nr = 300_000, 80_000
    rk = rng.permutation(200_000)[:nr].astype(np.int64) * 1_000_003
    lk = rng.integers(0, 200_000, nl, dtype=np.int64) * 1_000_003
    hot = rk.copy()
    hot[:20_000] = rk[7]
    (li, ri), kernels = _kernels_of(lambda: G.join([lk], [hot], nulls_equal=True, kind="inner"))
    assert kernels.get("join_partition_level2") == 1 and "join_build" in kernels, kernels  # (the build side tried, the probe side did not)
    el, er = oracle.join([lk], [hot], nulls_equal=True, kind="inner")
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)
    hj = HashJoin(cudf_amd.Table([G.to_device(rk)]), NullEquality.EQUAL)
    lhot = lk.copy()
    lhot[::2] = rk[11]
    li, ri = hj.inner_join(cudf_amd.Table([G.to_device(lhot)]))
    el, er = oracle.join([lhot], [rk], nulls_equal=True, kind="inner")
    assert kat.sorted_pairs(li.to_numpy()[0], ri.to_numpy()[0]) == kat.sorted_pairs(el, er)


# ---------------------------------------------------------------- partitioned dense join (big inner joins on a dense unique key)
@pytest.fixture
def force_dense_part(monkeypatch):
    """Makes small inputs take the partitioned dense join (engine.hpp dense_part_args) that big dense-key inner joins use."""
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_MIN_ROWS", "1")
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_PART_MIN_BUILD", "0")
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_PART_MIN_RANGE", "0")
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_PART_MIN_PROBE", "0")
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_PART_SLICE_LOG2", "10")
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_PROBE", "2")  # (the default, 1, is the ordered direct probe: test_dense_ordered_probe_*)


@pytest.mark.parametrize("inrange,expect_partitioned", [(0.95, True), (0.30, False)])
def test_dense_probe_is_chosen_from_the_in_range_share_of_the_probe_keys(G, oracle, monkeypatch, inrange, expect_partitioned):
    """With no CUDF_AMD_JOIN_DENSE_PROBE the join samples the probe keys: mostly INSIDE the table's range (every row costs the direct
    probe a random access) -> probe rows partitioned by key range; mostly outside (the range test rejects them for free) -> the
    ordered direct probe. Misses inside the range: the build side holds the even keys, the misses are odd (bench.py --config c3inrange)."""
    monkeypatch.delenv("CUDF_AMD_JOIN_DENSE_PROBE", raising=False)
    for k, v in (("MIN_ROWS", "1"), ("PART_MIN_BUILD", "0"), ("PART_MIN_RANGE", "0"), ("PART_MIN_PROBE", "0"), ("PART_SLICE_LOG2", "10"), ("ORDERED_MIN_PROBE", "0")):
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_" + k, v)
    rng = np.random.default_rng(int(inrange * 100))
    nl, nr = 700_000, 150_000
    rk = rng.permutation(nr).astype(np.int64) * 2 + 10_000                                  # even keys of [10000, 10000 + 2 nr)
    inside = rng.random(nl) < inrange
    lk = np.where(inside, rng.integers(0, 2 * nr, nl) + 10_000, rng.integers(0, 2 * nr, nl) + 10_000 + 4 * nr).astype(np.int64)
    (li, ri), kernels = _kernels_of(lambda: G.join([lk], [rk], nulls_equal=True, kind="inner"))
    assert kernels.get("join_sample") == 1, kernels
    assert (kernels.get("join_partition", 0) == 2) == expect_partitioned, kernels  # (2: the build side's stores and the probe side)
    el, er = oracle.join([lk], [rk], nulls_equal=True, kind="inner")
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)


@pytest.mark.parametrize("shape", ["unique", "nulls", "negative", "one_partition"])
def test_dense_part_join_matches_oracle(G, oracle, force_dense_part, monkeypatch, shape):
    """Unique dense build keys, probe keys inside and outside the build range (the latter are dropped by the partition pass),
    5 % NULLs on both sides (UNEQUAL), negative keys, a range that is a single slice."""
    rng = np.random.default_rng({"unique": 11, "nulls": 12, "negative": 13, "one_partition": 14}[shape])
    nl, nr = 600_000, 150_000
    span = 200_000
    if shape == "one_partition":
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_PART_SLICE_LOG2", "18")
    base = -123_456_789 if shape == "negative" else 1_000
    rk = rng.permutation(span)[:nr].astype(np.int64) + base
    lk = rng.integers(-50_000, span + 50_000, nl, dtype=np.int64) + base
    left, right, nulls_equal = [lk], [rk], True
    if shape == "nulls":
        left, right, nulls_equal = [(lk, rng.random(nl) > 0.05)], [(rk, rng.random(nr) > 0.05)], False
    (li, ri), kernels = _kernels_of(lambda: G.join(left, right, nulls_equal=nulls_equal, kind="inner"))
    assert kernels.get("join_partition") == 2, kernels  # (the build side and the probe side)
    el, er = oracle.join(left, right, nulls_equal=nulls_equal, kind="inner")
    assert len(li) == len(el)
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)


def test_dense_part_join_fallbacks(G, oracle, force_dense_part):
    """A repeated build key: the count of filled table entries differs from the number of rows and the atomic-exchange build (then
    the row lists) takes over. Probe rows sorted by key fill one ring tile after tile: that call takes the direct pass."""
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    rng = np.random.default_rng(15)
    nl, nr = 500_000, 120_000
    rk = rng.permutation(160_000)[:nr].astype(np.int64)
    lk = rng.integers(0, 200_000, nl, dtype=np.int64)
    dup = rk.copy()
    dup[1000] = dup[5]
    (li, ri), kernels = _kernels_of(lambda: G.join([lk], [dup], nulls_equal=True, kind="inner"))
    assert kernels.get("join_partition") == 1, kernels
    el, er = oracle.join([lk], [dup], nulls_equal=True, kind="inner")
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)
    hj = HashJoin(cudf_amd.Table([G.to_device(rk)]), NullEquality.EQUAL)
    for probe in (np.sort(lk), lk):
        t = cudf_amd.Table([G.to_device(probe)])
        gl, gr = hj.inner_join(t)
        el, er = oracle.join([probe], [rk], nulls_equal=True, kind="inner")
        assert kat.sorted_pairs(gl.to_numpy()[0], gr.to_numpy()[0]) == kat.sorted_pairs(el, er)
        assert hj.inner_join_size(t) == len(el)
    for kind in ("left", "full"):
        gl, gr = getattr(hj, kind + "_join")(cudf_amd.Table([G.to_device(lk)]))
        xl, xr = oracle.join([lk], [rk], nulls_equal=True, kind=kind)
        assert kat.sorted_pairs(gl.to_numpy()[0], gr.to_numpy()[0]) == kat.sorted_pairs(xl, xr)


@pytest.mark.parametrize("dtype", ["int32", "uint32"])
@pytest.mark.parametrize("path", ["dense_part", "radix"])
def test_partitioned_joins_take_four_byte_keys(G, oracle, monkeypatch, dtype, path):
    """One 4-byte integer key column: the partitioned joins widen it in their first scatter level (sign-extended for signed
    types), so int32 / uint32 keys take the same paths as int64 keys; negative keys, 5 % NULLs (UNEQUAL); a hash_join object
    whose build side took the partitions still answers left joins and sizes through the generic hash table."""
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    from oracle.oracle import HostColumn
    rng = np.random.default_rng(21 if dtype == "int32" else 22)
    nl, nr = 500_000, 120_000
    npt = np.dtype(dtype)
    if path == "dense_part":
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_MIN_ROWS", "1")
        for k, v in (("MIN_BUILD", "0"), ("MIN_RANGE", "0"), ("MIN_PROBE", "0"), ("SLICE_LOG2", "10")):
            monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_PART_" + k, v)
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_PROBE", "2")
        base = -70_000 if dtype == "int32" else 3_000_000_000  # (unsigned keys above 2^31: zero-extended, not sign-extended)
        rk = (rng.permutation(200_000)[:nr].astype(np.int64) + base).astype(npt)
        lk = (rng.integers(-20_000, 240_000, nl, dtype=np.int64) + base).astype(npt)
    else:
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE", "0")
        monkeypatch.setenv("CUDF_AMD_JOIN_RADIX_MIN_BUILD", "0")
        monkeypatch.setenv("CUDF_AMD_JOIN_RADIX_MIN_PROBE", "0")
        lo, hi = (-2**31, 2**31 - 1) if dtype == "int32" else (0, 2**32 - 1)
        pool = rng.integers(lo, hi, 150_000, dtype=np.int64)
        rk = pool[rng.integers(0, 100_000, nr)].astype(npt)           # duplicates on the build side
        lk = pool[rng.integers(0, 150_000, nl)].astype(npt)           # a third of the probe keys have no partner
    lv, rv = rng.random(nl) > 0.05, rng.random(nr) > 0.05
    left, right = [HostColumn(lk, lv, dtype)], [HostColumn(rk, rv, dtype)]
    (li, ri), kernels = _kernels_of(lambda: G.join(left, right, nulls_equal=False, kind="inner"))
    assert kernels.get("join_partition", 0) >= 2, kernels
    el, er = oracle.join(left, right, nulls_equal=False, kind="inner")
    assert len(li) == len(el)
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)
    hj = HashJoin(cudf_amd.Table([G.to_device(right[0])]), NullEquality.UNEQUAL)
    t = cudf_amd.Table([G.to_device(left[0])])
    gl, gr = hj.left_join(t)
    xl, xr = oracle.join(left, right, nulls_equal=False, kind="left")
    assert kat.sorted_pairs(gl.to_numpy()[0], gr.to_numpy()[0]) == kat.sorted_pairs(xl, xr)
    assert hj.inner_join_size(t) == len(el)
    pl, pr = hj.inner_join(t)
    assert kat.sorted_pairs(pl.to_numpy()[0], pr.to_numpy()[0]) == kat.sorted_pairs(el, er)


@pytest.mark.parametrize("shape", ["int64", "int64_nulls", "int32_negative", "uint32_high", "atomic_build"])
def test_dense_ordered_probe_yields_pairs_in_probe_row_order(G, oracle, monkeypatch, shape):
    """Inner joins on a dense unique build key take the ordered direct probe (engine.hpp dense_stage_args): every wave walks its own
    contiguous range of probe rows and appends its pairs to its own stage, so the pairs come out sorted by probe row - what a
    payload gather by them wants. 8-byte and 4-byte keys, NULLs (UNEQUAL), probe keys outside the build range; a table built by the
    partitioned stores and one built by the atomic exchanges (`atomic_build`); a probe of rows [a, b) of a bigger table through the
    match context keeps the row base."""
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    from oracle.oracle import HostColumn
    rng = np.random.default_rng({"int64": 31, "int64_nulls": 32, "int32_negative": 33, "uint32_high": 34, "atomic_build": 35}[shape])
    nl, nr = 900_001, 150_000
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_MIN_ROWS", "1")
    monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_ORDERED_MIN_PROBE", "0")
    if shape != "atomic_build":
        for k, v in (("MIN_BUILD", "0"), ("MIN_RANGE", "0"), ("SLICE_LOG2", "10")):
            monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_PART_" + k, v)
    dtype = {"int32_negative": "int32", "uint32_high": "uint32"}.get(shape, "int64")
    base = {"int32_negative": -90_000, "uint32_high": 4_000_000_000}.get(shape, 10**12)
    npt = np.dtype(dtype)
    rk = (rng.permutation(220_000)[:nr].astype(np.int64) + base).astype(npt)
    lk = (rng.integers(-30_000, 260_000, nl, dtype=np.int64) + base).astype(npt)
    lv = rng.random(nl) > 0.05 if shape == "int64_nulls" else None
    rv = rng.random(nr) > 0.05 if shape == "int64_nulls" else None
    left, right = [HostColumn(lk, lv, dtype)], [HostColumn(rk, rv, dtype)]
    hj = HashJoin(cudf_amd.Table([G.to_device(right[0])]), NullEquality.UNEQUAL)
    t = cudf_amd.Table([G.to_device(left[0])])
    (pl, pr), kernels = _kernels_of(lambda: hj.inner_join(t))
    assert "join_retrieve" in kernels and kernels.get("join_partition", 0) == 0, kernels  # (count + copy, no probe-side partition pass)
    li, ri = pl.to_numpy()[0], pr.to_numpy()[0]
    assert np.all(np.diff(li.astype(np.int64)) >= 0), "pairs are not in probe-row order"
    el, er = oracle.join(left, right, nulls_equal=False, kind="inner")
    assert len(li) == len(el)
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)
    assert hj.inner_join_size(t) == len(el)
    # LEFT join against the same table: one pair per probe row, in order, from a single pass (launch_dense_left_direct)
    (gl, gr), kernels = _kernels_of(lambda: hj.left_join(t))
    assert "join_count" not in kernels, kernels
    gl, gr = gl.to_numpy()[0], gr.to_numpy()[0]
    assert np.array_equal(gl, np.arange(nl, dtype=gl.dtype))
    xl, xr = oracle.join(left, right, nulls_equal=False, kind="left")
    assert kat.sorted_pairs(gl, gr) == kat.sorted_pairs(xl, xr)


@pytest.mark.parametrize("path", ["dense_ordered", "dense_partitioned", "radix"])
@pytest.mark.parametrize("kind", ["inner", "left"])
def test_partitioned_join_ranges_through_the_round3_paths(G, oracle, monkeypatch, path, kind):
    """Chunked probing (partitioned_{inner,left}_join over row ranges of the left table, reference hash_join.hpp:276-441) with the
    range probes going through the ordered direct probe / the one-pass left join, the partitioned dense probe and the LDS radix
    join: the emitted left indices carry the range's first row, the union over the ranges is the whole join."""
    import cudf_amd
    from cudf_amd.join import HashJoin, JoinPartitionContext
    from cudf_amd.types import NullEquality
    rng = np.random.default_rng(88)
    nl, nr = 700_000, 90_000
    if path == "radix":
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE", "0")
        monkeypatch.setenv("CUDF_AMD_JOIN_RADIX_MIN_BUILD", "0")
        monkeypatch.setenv("CUDF_AMD_JOIN_RADIX_MIN_PROBE", "0")
        rk = rng.permutation(300_000)[:nr].astype(np.int64) * 1_000_003
        lk = rng.integers(0, 400_000, nl, dtype=np.int64) * 1_000_003
    else:
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_MIN_ROWS", "1")
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_ORDERED_MIN_PROBE", "0")
        for k, v in (("MIN_BUILD", "0"), ("MIN_RANGE", "0"), ("MIN_PROBE", "0"), ("SLICE_LOG2", "10")):
            monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_PART_" + k, v)
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_PROBE", "1" if path == "dense_ordered" else "2")
        rk = rng.permutation(200_000)[:nr].astype(np.int64) - 5_000
        lk = rng.integers(-30_000, 230_000, nl, dtype=np.int64)
    lv, rv = rng.random(nl) > 0.05, rng.random(nr) > 0.05
    hj = HashJoin(cudf_amd.Table([G.to_device((rk, rv))]), NullEquality.UNEQUAL)
    left = cudf_amd.Table([G.to_device((lk, lv))])
    ctx = getattr(hj, f"{kind}_join_match_context")(left)
    counts = ctx._match_counts.to_numpy()[0]
    bounds = [0, 3, 250_000, 250_000, 699_999, nl]
    ls, rs = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        li, ri = getattr(hj, f"partitioned_{kind}_join")(JoinPartitionContext(ctx, a, b))
        l, r = li.to_numpy()[0], ri.to_numpy()[0]
        assert len(l) == int(counts[a:b].sum())
        assert len(l) == 0 or (l.min() >= a and l.max() < b)
        ls.append(l)
        rs.append(r)
    el, er = oracle.join([(lk, lv)], [(rk, rv)], nulls_equal=False, kind=kind)
    assert kat.sorted_pairs(np.concatenate(ls), np.concatenate(rs)) == kat.sorted_pairs(el, er)


@pytest.mark.parametrize("kind", ["inner", "left"])
@pytest.mark.parametrize("nulls", [False, True])
def test_radix_join_packs_two_four_byte_key_columns(G, oracle, force_radix_join, kind, nulls):
    """TWO 4-byte integer key columns (int32, uint32) are packed into the radix join's 8-byte key by its first scatter level -
    rows are equal iff both columns are; a NULL in either column drops the row (UNEQUAL). Duplicates on both sides, negative
    values. Left joins take the path when no probe column has NULLs; with NULLs they fall back to the hash table (still right)."""
    from oracle.oracle import HostColumn
    rng = np.random.default_rng(55 + int(nulls))
    nl, nr = 600_000, 100_000
    a_pool = rng.integers(-2**31, 2**31 - 1, 3000, dtype=np.int64).astype(np.int32)
    b_pool = rng.integers(0, 2**32 - 1, 80, dtype=np.int64).astype(np.uint32)
    ra, rb = a_pool[rng.integers(0, 2500, nr)], b_pool[rng.integers(0, 60, nr)]
    la, lb = a_pool[rng.integers(0, 3000, nl)], b_pool[rng.integers(0, 80, nl)]
    def col(x, t):
        return HostColumn(x, (rng.random(len(x)) > 0.05) if nulls else None, t)
    left, right = [col(la, "int32"), col(lb, "uint32")], [col(ra, "int32"), col(rb, "uint32")]
    (li, ri), kernels = _kernels_of(lambda: G.join(left, right, nulls_equal=False, kind=kind))
    expect_radix_probe = not (kind == "left" and nulls)
    assert kernels.get("join_partition_level2") == (2 if expect_radix_probe else 1), kernels
    el, er = oracle.join(left, right, nulls_equal=False, kind=kind)
    assert len(li) == len(el)
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)


@pytest.mark.parametrize("path", ["dense", "radix"])
def test_round3_join_paths_edge_cases(G, oracle, monkeypatch, path):
    """No pair at all (every probe key outside the build keys), a probe side that is all NULL, a probe side of one row, on the
    ordered dense probe / one-pass left join and on the radix join: empty or all-JoinNoMatch results of the right sizes."""
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    rng = np.random.default_rng(99)
    nr = 70_000
    if path == "radix":
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE", "0")
        monkeypatch.setenv("CUDF_AMD_JOIN_RADIX_MIN_BUILD", "0")
        monkeypatch.setenv("CUDF_AMD_JOIN_RADIX_MIN_PROBE", "0")
        rk = rng.permutation(200_000)[:nr].astype(np.int64) * 1_000_003
        outside = (rng.integers(0, 100_000, 300_000).astype(np.int64) + 500_000) * 1_000_003
    else:
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_MIN_ROWS", "1")
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_ORDERED_MIN_PROBE", "0")
        rk = rng.permutation(100_000)[:nr].astype(np.int64) + 1_000
        outside = np.concatenate([rng.integers(-50_000, 1_000, 150_000), rng.integers(101_000, 200_000, 150_000)]).astype(np.int64)
    hj = HashJoin(cudf_amd.Table([G.to_device(rk)]), NullEquality.UNEQUAL)
    cases = [(outside, None), (outside[:5000], np.zeros(5000, dtype=bool)), (rk[:1].copy(), None), (outside[:1].copy(), None)]
    for lk, lv in cases:
        t = cudf_amd.Table([G.to_device((lk, lv) if lv is not None else lk)])
        for kind in ("inner", "left"):
            gl, gr = getattr(hj, kind + "_join")(t)
            el, er = oracle.join([(lk, lv) if lv is not None else lk], [rk], nulls_equal=False, kind=kind)
            assert gl.size() == len(el), (kind, len(lk))
            assert kat.sorted_pairs(gl.to_numpy()[0], gr.to_numpy()[0]) == kat.sorted_pairs(el, er)


def test_join_size_against_a_radix_build_runs_the_count_pass_only(G, oracle, force_radix_join):
    """inner_join_size / left_join_size on a hash_join whose build side took the radix partitions: the radix join's count pass answers
    (no open-addressing copy of the build side is built just to count - ADVICE r3), and the sizes equal the joins' pair counts."""
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    rng = np.random.default_rng(77)
    nl, nr = 400_000, 60_000
    rk = (rng.permutation(200_000)[:nr].astype(np.int64)) * 1_000_003
    lk = (rng.integers(0, 260_000, nl, dtype=np.int64)) * 1_000_003
    lv = rng.random(nl) > 0.05
    hj = HashJoin(cudf_amd.Table([G.to_device(rk)]), NullEquality.UNEQUAL)
    t = cudf_amd.Table([G.to_device((lk, lv))])
    n_inner, kernels = _kernels_of(lambda: hj.inner_join_size(t))
    assert kernels.get("join_build", 0) == 0, kernels  # (nothing built by the size call)
    n_left = hj.left_join_size(t)
    el, _ = oracle.join([(lk, lv)], [rk], nulls_equal=False, kind="inner")
    assert n_inner == len(el)
    assert n_left == len(el) + int(nl - len(np.unique(el)))
    li, _ = hj.inner_join(t)
    assert li.size() == n_inner
    del hj  # (destroyed right behind an unsynchronised probe: the mirror drains the probe stream first)


@pytest.mark.parametrize("path", ["radix", "radix_two_columns", "dense"])
def test_full_join_on_the_partitioned_paths(G, oracle, monkeypatch, path):
    """FULL joins no longer need the open-addressing table (round 4): the partitioned LEFT join + the build rows that appear in none of
    its pairs (marked from the build indices, appended by the complement kernel). Sizes agree with full_join_size."""
    import cudf_amd
    from cudf_amd.join import HashJoin
    from cudf_amd.types import NullEquality
    rng = np.random.default_rng({"radix": 21, "radix_two_columns": 22, "dense": 23}[path])
    nl, nr = 300_000, 60_000
    if path == "dense":
        for k, v in (("MIN_ROWS", "1"), ("ORDERED_MIN_PROBE", "0")):
            monkeypatch.setenv("CUDF_AMD_JOIN_DENSE_" + k, v)
        rk = rng.permutation(90_000)[:nr].astype(np.int64) + 5_000
        lk = rng.integers(0, 120_000, nl, dtype=np.int64)
        left, right = [lk], [rk]
    else:
        monkeypatch.setenv("CUDF_AMD_JOIN_DENSE", "0")
        monkeypatch.setenv("CUDF_AMD_JOIN_RADIX_MIN_BUILD", "0")
        monkeypatch.setenv("CUDF_AMD_JOIN_RADIX_MIN_PROBE", "0")
        rk = rng.integers(0, 80_000, nr, dtype=np.int64) * 1_000_003      # duplicates on the build side
        lk = rng.integers(0, 110_000, nl, dtype=np.int64) * 1_000_003
        left, right = [lk], [rk]
        if path == "radix_two_columns":
            left, right = [lk, (lk % 977).astype(np.int32)], [rk, (rk % 977).astype(np.int32)]
    hj = HashJoin(cudf_amd.Table([G.to_device(c) for c in right]), NullEquality.EQUAL)
    t = cudf_amd.Table([G.to_device(c) for c in left])
    (pl, pr), kernels = _kernels_of(lambda: hj.full_join(t))
    assert kernels.get("join_complement") == 1 and kernels.get("join_build", 0) == 0, kernels  # (no table was built for it)
    li, ri = pl.to_numpy()[0], pr.to_numpy()[0]
    el, er = oracle.join(left, right, nulls_equal=True, kind="full")
    assert len(li) == len(el) == hj.full_join_size(t)
    assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(el, er)
