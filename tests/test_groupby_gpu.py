"""GPU parity tests for cudf::groupby::groupby::aggregate (hash path) through the C ABI: the reference's own
KATs, then seeded random inputs against the CPU oracle on every kernel family (LDS single pass, one- and
two-level partitioned), then size-independent properties at large sizes."""
import os

import numpy as np
import pytest

import kat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(gpu):
    import gpu_backend
    return gpu_backend


@pytest.mark.parametrize("name,c,kt,vt", list(kat.groupby_cases()), ids=[x[0] for x in kat.groupby_cases()])
def test_groupby_kat(G, name, c, kt, vt):
    kat.run_groupby_case(G, c, kt, vt)


def test_groupby_generated_kats(G):
    k = np.arange(512, dtype=np.int32)  # max_tests.cpp:554-574
    kc, rc = kat.sort_groups(*G.groupby([k], [(k, ["max"])]))
    assert np.array_equal(kc[0][0], k) and np.array_equal(rc[0][0][0], k)
    rng = np.random.default_rng(0)  # max_tests.cpp:576-597
    k = np.tile(np.arange(128, dtype=np.int32), 10000)
    rng.shuffle(k)
    kc, rc = kat.sort_groups(*G.groupby([k], [(k, ["max"])]))
    assert np.array_equal(kc[0][0], np.arange(128)) and np.array_equal(rc[0][0][0], np.arange(128))
    K10 = np.array([1, 2, 3, 1, 2, 2, 1, 3, 3, 2], np.int32)  # keys_tests.cpp:355-409
    V10 = np.arange(10, dtype=np.int32)
    kc, rc = kat.sort_groups(*G.groupby([K10], [(V10, ["sum", "sum"])]))
    assert [list(c[0]) for c in rc[0]] == [[9, 19, 17], [9, 19, 17]]
    kc, rc = kat.sort_groups(*G.groupby([K10], [(V10, ["sum"]), (V10, ["sum"])]))
    assert list(rc[0][0][0]) == [9, 19, 17] and list(rc[1][0][0]) == [9, 19, 17]


def test_groupby_errors(G):
    from cudf_amd._lib import CudfAmdError
    K = np.array([1, 2, 3], np.int32)
    with pytest.raises(CudfAmdError, match="Size mismatch"):  # groupby.cu:225-229 -> cudf::logic_error
        G.groupby([K], [(np.arange(4, dtype=np.int32), ["sum"])])
    from oracle.oracle import HostColumn
    with pytest.raises(CudfAmdError, match="Invalid type/aggregation"):  # groupby.cu:186-201
        G.groupby([K], [(HostColumn(np.arange(3, dtype=np.int64), None, "timestamp_s"), ["sum"])])
    with pytest.raises(CudfAmdError, match="Only arithmetic types"):  # group_quantiles.cu:125-131
        G.groupby([K], [(HostColumn(np.arange(3, dtype=np.int64), None, "timestamp_s"), ["median"])])


def _check_against_oracle(G, O, keys, requests, include=False, expect_path=None):
    got = kat.sort_groups(*G.groupby(keys, requests, include_null_keys=include))
    if expect_path is not None:
        # dense integer keys may take the direct-address tables (DENSE_DIRECT) wherever the partitioned path is expected
        ok = (expect_path, "DENSE_DIRECT") if expect_path == "PARTITIONED_LDS" else (expect_path,)
        assert G.last_path.name in ok, G.last_path
    exp = kat.sort_groups(*O.groupby(keys, requests, include_null_keys=include))
    assert len(got[0]) == len(exp[0])
    for a, e in zip(got[0], exp[0]):
        kat.compare_columns(a, e, "keys")
    # order-dependent float sums: worst-case bound for the largest group (see kat.sum_atol)
    cnt = O.groupby(keys, [(requests[0][0], ["count_all"])], include_null_keys=include)[1][0][0][0]
    m = int(cnt.max()) if len(cnt) else 1
    for (vals, kinds), ra, re_ in zip(requests, got[1], exp[1]):
        h = G.to_host_column(vals)
        scale = float(np.max(np.abs(h.data.astype(np.float64)))) if h.size else 0.0
        for kind, a, e in zip(kinds, ra, re_):
            is_fsum = kind in ("sum", "mean", "sum_of_squares") and np.dtype(h.data.dtype).kind == "f"
            kat.compare_columns(a, e, f"result[{kind}]", atol=kat.sum_atol(m, scale) if is_fsum else 0.0)


ALL_AGGS = ["sum", "count_valid", "count_all", "min", "max", "mean"]


@pytest.mark.parametrize("n,groups,path", [(1000, 7, "LDS_SINGLE_PASS"), (200_000, 300, "LDS_SINGLE_PASS"),
                                           (300_000, 50_000, "PARTITIONED_LDS"),
                                           (2_000_000, 1_000_000, "PARTITIONED_LDS")])
def test_random_int64_key_f64_value(G, oracle, n, groups, path):
    rng = np.random.default_rng(42)
    k = rng.integers(0, groups, n, dtype=np.int64) * 7919 - 3  # arbitrary, non-dense int64 keys
    v = rng.random(n)
    _check_against_oracle(G, oracle, [k], [(v, ALL_AGGS)], expect_path=path)


@pytest.mark.parametrize("include", [False, True])
@pytest.mark.parametrize("n,g0,g1", [(5000, 10, 5), (400_000, 2000, 300)])
def test_random_multikey_nulls(G, oracle, n, g0, g1, include):
    """C4 shape at small size: keys (int64, int32) with 10% nulls on k1 and on the value column."""
    rng = np.random.default_rng(46)
    k0 = rng.integers(0, g0, n, dtype=np.int64)
    k1 = rng.integers(0, g1, n, dtype=np.int32)
    k1v = rng.random(n) > 0.1
    v = rng.random(n)
    vv = rng.random(n) > 0.1
    _check_against_oracle(G, oracle, [k0, (k1, k1v)], [((v, vv), ["mean", "min", "max", "sum", "count_valid", "count_all"])],
                          include=include)


@pytest.mark.parametrize("vt", ["int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float32",
                                "float64", "bool"])
def test_value_types(G, oracle, vt):
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(7)
    n = 50_000
    k = rng.integers(0, 97, n, dtype=np.int32)
    npt = NP_OF_TYPE_ID[TYPE_ID[vt]]
    if vt == "bool":
        v = rng.integers(0, 2, n).astype(np.uint8)
    elif np.dtype(npt).kind == "f":
        v = (rng.random(n) * 1000 - 500).astype(npt)
    else:
        info = np.iinfo(npt)
        v = rng.integers(max(info.min, -2**40), min(info.max, 2**40), n, dtype=np.int64).astype(npt)
    vv = rng.random(n) > 0.2
    aggs = ["sum", "min", "max", "count_valid", "mean"]
    _check_against_oracle(G, oracle, [k], [(HostColumn(v, vv, vt), aggs)])


@pytest.mark.parametrize("kt", ["int8", "int16", "uint32", "uint64", "bool", "timestamp_ms", "duration_days", "decimal64"])
def test_key_types(G, oracle, kt):
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(9)
    n = 30_000
    npt = NP_OF_TYPE_ID[TYPE_ID[kt]]
    k = rng.integers(0, 2 if kt == "bool" else 100, n).astype(np.uint8 if kt == "bool" else npt)
    kv = rng.random(n) > 0.05
    v = rng.integers(-1000, 1000, n, dtype=np.int64)
    for include in (False, True):
        _check_against_oracle(G, oracle, [HostColumn(k, kv, kt)], [(v, ["sum", "count_all"])], include=include)


@pytest.mark.parametrize("vt", ["int32", "int64", "float64", "float32"])
def test_variance_std_m2(G, oracle, vt):
    """M2 / VARIANCE / STD on the hash path = SUM_OF_SQUARES, SUM, COUNT_VALID combined as the reference does
    (groupby/common/m2_var_std.cu:48-60,152-187); groups of one row give null VARIANCE/STD (ddof = 1)."""
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(21)
    n = 60_000
    k = rng.integers(0, 400, n, dtype=np.int64)
    k[:50] = np.arange(1000, 1050)  # singleton groups
    npt = NP_OF_TYPE_ID[TYPE_ID[vt]]
    v = (rng.random(n) * 20 - 10).astype(npt) if np.dtype(npt).kind == "f" else rng.integers(-50, 50, n).astype(npt)
    vv = rng.random(n) > 0.1
    got = kat.sort_groups(*G.groupby([k], [(HostColumn(v, vv, vt), ["m2", "variance", "std", "sum_of_squares"])]))
    exp = kat.sort_groups(*oracle.groupby([k], [(HostColumn(v, vv, vt), ["m2", "variance", "std", "sum_of_squares"])]))
    kat.compare_columns(got[0][0], exp[0][0], "keys")
    m = 400  # rows per group upper bound (n / 400 groups ~ 150, generous)
    # sumsq - sum^2/n cancels: both sides carry rounding of order eps * m * max(v)^2 (float32 inputs: our float64
    # accumulation vs the float32 of the reference differs by float32 rounding of each square)
    tol = kat.sum_atol(m, 100.0) * (1e7 if vt == "float32" else 1.0)
    for name, a, e in zip(["m2", "variance", "std", "sum_of_squares"], got[1][0], exp[1][0]):
        kat.compare_columns(a, e, name, atol=tol)
    assert got[1][0][1][1] is not None and not got[1][0][1][1].all()  # singleton groups -> null variance


@pytest.mark.parametrize("vt", ["int8", "int64", "float64", "float32"])
def test_product(G, oracle, vt):
    """PRODUCT (integral -> int64 wrapping, float -> same type): LDS compare-and-swap loop; integer products are
    order-independent mod 2^64 and must be bit-exact, float products carry one rounding per factor."""
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(22)
    n = 40_000
    k = rng.integers(0, 500, n, dtype=np.int64)
    npt = NP_OF_TYPE_ID[TYPE_ID[vt]]
    v = (0.9 + 0.2 * rng.random(n)).astype(npt) if np.dtype(npt).kind == "f" else rng.integers(-3, 4, n).astype(npt)
    vv = rng.random(n) > 0.1
    got = kat.sort_groups(*G.groupby([k], [(HostColumn(v, vv, vt), ["product", "count_valid"])]))
    exp = kat.sort_groups(*oracle.groupby([k], [(HostColumn(v, vv, vt), ["product", "count_valid"])]))
    kat.compare_columns(got[0][0], exp[0][0], "keys")
    kat.compare_columns(got[1][0][1], exp[1][0][1], "count_valid")
    if np.dtype(npt).kind == "f":
        a, e = got[1][0][0], exp[1][0][0]
        assert a[2] == e[2] and np.array_equal(a[1], e[1])
        # ~80 factors per group in [0.9, 1.1]: relative error <= factors * eps of the result type
        rel = 200 * (np.finfo(np.float32).eps if vt == "float32" else np.finfo(np.float64).eps)
        ok = a[1] if a[1] is not None else np.ones(len(a[0]), bool)
        assert np.allclose(a[0][ok].astype(np.float64), e[0][ok].astype(np.float64), rtol=rel, atol=0.0)
    else:
        kat.compare_columns(got[1][0][0], exp[1][0][0], "product")


@pytest.mark.parametrize("vt", ["int32", "int64", "float64", "float32"])
@pytest.mark.parametrize("n,groups", [(30_000, 300), (6_000_000, 3_000), (5_000_000, 2_500_000)])
def test_argmin_argmax(G, oracle, vt, n, groups):
    """ARGMIN / ARGMAX = smallest row index among the rows that attain the group's extreme value (two sweeps over the
    rows; deterministic where the reference's CAS loop is arrival-order dependent). Values repeat, so ties are common;
    sizes cover the single-table, merged and partitioned (row index carried in the records) paths."""
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(23)
    k = rng.integers(0, groups, n, dtype=np.int64)
    npt = NP_OF_TYPE_ID[TYPE_ID[vt]]
    v = rng.integers(-20, 20, n).astype(npt)
    vv = rng.random(n) > 0.1
    col = HostColumn(v, vv, vt)
    aggs = ["argmin", "argmax", "min", "max"]
    got = kat.sort_groups(*G.groupby([k], [(col, aggs)]))
    exp = kat.sort_groups(*oracle.groupby([k], [(col, aggs)]))
    kat.compare_columns(got[0][0], exp[0][0], "keys")
    for name, a, e in zip(aggs, got[1][0], exp[1][0]):
        kat.compare_columns(a, e, name)


def test_sliced_columns_offset(G, oracle):
    """Arrow offset: element i at data[offset+i], validity at bit offset+i."""
    from oracle.oracle import HostColumn
    rng = np.random.default_rng(3)
    n, off = 10_000, 37
    k = rng.integers(0, 50, n, dtype=np.int64)
    kv = rng.random(n) > 0.1
    v = rng.random(n)
    vv = rng.random(n) > 0.1
    keys = [HostColumn(k, kv, "int64", offset=off)]
    reqs = [(HostColumn(v, vv, "float64", offset=off), ["sum", "count_valid", "max"])]
    _check_against_oracle(G, oracle, keys, reqs)
    _check_against_oracle(G, oracle, keys, reqs, include=True)


def test_all_rows_one_group_and_all_distinct(G, oracle):
    n = 100_000
    v = np.arange(n, dtype=np.float64)
    _check_against_oracle(G, oracle, [np.zeros(n, np.int64)], [(v, ALL_AGGS)])
    _check_against_oracle(G, oracle, [np.arange(n, dtype=np.int64)], [(v, ["sum", "count_all"])])


def test_skewed_keys_retry(G, oracle):
    """A sample-based cardinality estimate that is too low must be repaired by the overflow retry."""
    rng = np.random.default_rng(11)
    n = 1_500_000
    k = np.zeros(n, np.int64)
    k[-400_000:] = rng.integers(1, 300_000, 400_000)  # distinct tail the strided sample under-represents
    v = rng.random(n)
    _check_against_oracle(G, oracle, [k], [(v, ["sum", "count_all"])])


@pytest.mark.parametrize("kind", ["periodic", "zipf"])
def test_misjudged_cardinality_is_repaired_quickly(G, oracle, kind):
    """Key distributions whose sample misjudges the group count: `row % 2M` (a fixed sampling stride aliases with the
    period) and a log-uniform (Zipf-like) column whose tail the sample barely sees. The tables of the first attempt
    overflow; the workgroups must stop probing their saturated tables (this took seconds) and the planner must recount
    over all rows (HyperLogLog) instead of escalating blindly. Results checked against the oracle; bounded time."""
    import time
    rng = np.random.default_rng(23)
    n, groups = 8_000_000, 2_000_000
    if kind == "periodic":
        k = np.arange(n, dtype=np.int64) % groups
    else:
        k = np.clip((float(groups) ** rng.random(n)).astype(np.int64) - 1, 0, groups - 1)
    v = rng.random(n)
    G.groupby([k], [(v, ["sum"])])  # warm-up (allocator, module load)
    t0 = time.perf_counter()
    G.groupby([k], [(v, ["sum", "count_all"])])
    dt = time.perf_counter() - t0
    _check_against_oracle(G, oracle, [k], [(v, ["sum", "count_all"])], expect_path="PARTITIONED_LDS")
    assert dt < 1.0, f"{kind}: {dt:.3f} s for {n} rows (host copies included) - a saturated-table walk is back"


@pytest.mark.parametrize("kt", ["int32", "int16", "uint32", "float32"])
@pytest.mark.parametrize("vt", ["float64", "int8"])
def test_narrow_key_nullable_value_partitioned(G, oracle, kt, vt):
    """A key of at most 4 bytes carries the validity flags of a nullable value in its spare half, and the record is 16
    bytes - the aggregate's one-load record path. (It once dropped the flags: null values were summed and counted; found
    by fuzz seed 424.) Groups whose values are all null must come out null."""
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(31)
    n, groups = 300_000, 30_000 if kt == "int16" else 120_000
    k = rng.integers(0, groups, n).astype(NP_OF_TYPE_ID[TYPE_ID[kt]])
    v = rng.integers(-9, 9, n).astype(NP_OF_TYPE_ID[TYPE_ID[vt]])
    vv = rng.random(n) > 0.3
    _check_against_oracle(G, oracle, [HostColumn(k, None, kt)], [(HostColumn(v, vv, vt), ["sum", "count_valid", "mean", "max", "count_all"])],
                          expect_path="PARTITIONED_LDS")


def test_optimistic_partition_and_its_fallback(G, oracle):
    """n >= 4M rows takes the optimistic single-pass partition (no histogram pass) through the write-combining scatter + tagged
    tables. Uniform keys must stay on it; a heavy-hitter key overflows its fixed-capacity region and must be repaired by the
    exact pipeline."""
    rng = np.random.default_rng(17)
    n = 5_000_000
    k = rng.integers(0, 200_000, n, dtype=np.int64)
    v = rng.random(n)
    _check_against_oracle(G, oracle, [k], [(v, ["sum", "count_all", "min"])], expect_path="PARTITIONED_LDS")
    k2 = k.copy()
    k2[rng.random(n) < 0.5] = 7  # half of all rows carry one key
    _check_against_oracle(G, oracle, [k2], [(v, ["sum", "count_all", "max"])], expect_path="PARTITIONED_LDS")


@pytest.mark.parametrize("groups,vt,kinds", [(900_000, "float64", ["sum", "count_valid"]), (1_100_000, "int64", ["sum", "count_all", "mean"]),
                                             (40_000, "float64", ["min", "max", "mean"]), (300_000, "float64", ["sum_of_squares", "product", "count_valid", "max", "sum"]),
                                             (5_000, "uint64", ["sum"])])
def test_sparse_keys_with_edge_values(G, oracle, groups, vt, kinds):
    """Sparse int64 keys (no small range: never the direct-address tables), one plain 8-byte value column, n >= 4M: the
    write-combining scatter + tagged LDS tables. The keys include 0 / -1 / INT64_MIN / INT64_MAX and a golden-ratio pattern.
    (Round 3 also had a ring scatter + key-word tables for this shape; measured slower - profiles/r3_sparse_ring.txt - and removed.)"""
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(groups)
    n = 4_400_000
    pool = rng.integers(-2**62, 2**62, groups, dtype=np.int64)
    pool[:6] = [0, -1, np.iinfo(np.int64).min, np.iinfo(np.int64).max, np.int64(np.uint64(0x9e3779b97f4a7c15).astype(np.int64)), 1]
    k = pool[rng.integers(0, groups, n)]
    npt = NP_OF_TYPE_ID[TYPE_ID[vt]]
    v = (rng.random(n) * 2 - 1).astype(npt) if vt == "float64" else rng.integers(0, 1000, n).astype(npt)
    if "product" in kinds:
        v = np.where(rng.random(n) < 0.999, 1.0, 1.0 + 2.0 ** -20)
    _check_against_oracle(G, oracle, [k], [(HostColumn(v, None, vt), kinds)], expect_path="PARTITIONED_LDS")
    assert G.last_path.name == "PARTITIONED_LDS"


def test_sparse_keys_hidden_groups_and_clustered_rows(G, oracle):
    """The optimistic partition repairs itself: a skewed sample that hides most of the groups (the tables overflow), and keys
    clustered in the row order (a region overflows)."""
    rng = np.random.default_rng(909)
    n = 4_500_000
    # 95 % of the rows on 2000 keys, the rest on 1.5M more: the sample estimates far too few groups
    k = np.where(rng.random(n) < 0.95, rng.integers(0, 2000, n), rng.integers(0, 1_500_000, n)).astype(np.int64) * 1_000_003
    v = rng.random(n)
    _check_against_oracle(G, oracle, [k], [(v, ["sum", "count_valid"])], expect_path="PARTITIONED_LDS")
    # runs of 300 equal keys: whole keys land in one workgroup's tile
    k2 = (np.arange(n, dtype=np.int64) // 300 % 120_000) * 1_000_003
    _check_against_oracle(G, oracle, [k2], [(v, ["sum", "max"])], expect_path="PARTITIONED_LDS")


@pytest.mark.parametrize("ring", ["1", "0"])
@pytest.mark.parametrize("lo,groups,vt", [(0, 1_000_000, "float64"), (-5_000_000_000, 300_000, "int64"), (2**62, 60_000, "float64")])
def test_dense_keys_direct_address(G, oracle, monkeypatch, lo, groups, vt, ring):
    """One plain int64 key column spanning a small range, n >= 4M: direct-address LDS tables (no hash, no probe, keys
    rebuilt from the slot index), through the ring scatter (12-byte records in two streams) and through the write-combining
    scatter (16-byte records). Same call with CUDF_AMD_GB_DENSE=0 must take the hash tables and agree."""
    monkeypatch.setenv("CUDF_AMD_GB_DENSE_RING", ring)
    rng = np.random.default_rng(71)
    n = 4_500_000
    k = rng.integers(0, groups, n, dtype=np.int64) + lo
    v = rng.random(n) if vt == "float64" else rng.integers(-10**6, 10**6, n, dtype=np.int64)
    _check_against_oracle(G, oracle, [k], [(v, ["sum", "count_valid", "min", "max", "mean"])], expect_path="DENSE_DIRECT")
    monkeypatch.setenv("CUDF_AMD_GB_DENSE", "0")
    _check_against_oracle(G, oracle, [k], [(v, ["sum", "count_valid"])], expect_path="PARTITIONED_LDS")
    assert G.last_path.name == "PARTITIONED_LDS"


def test_dense_keys_write_combining_and_fallbacks(G, oracle, monkeypatch):
    """The write-combining form of the dense path (CUDF_AMD_GB_DENSE_RING=0: 16-byte records); a key outside the sampled range and a
    sparse key column must fall back to the hash tables."""
    rng = np.random.default_rng(72)
    n = 6_000_000
    k = rng.integers(0, 500_000, n, dtype=np.int64)
    v = rng.random(n)
    monkeypatch.setenv("CUDF_AMD_GB_DENSE_RING", "0")
    _check_against_oracle(G, oracle, [k], [(v, ["sum", "count_all", "max"])], expect_path="DENSE_DIRECT")
    monkeypatch.delenv("CUDF_AMD_GB_DENSE_RING")
    k2 = k.copy()
    k2[n // 2 + 12345] = 10**12  # one outlier the sample will not see: the attempt is void, the call is redone by hash
    _check_against_oracle(G, oracle, [k2], [(v, ["sum", "count_all"])], expect_path="PARTITIONED_LDS")
    assert G.last_path.name == "PARTITIONED_LDS"
    k3 = k * 1_000_003  # sparse keys: never dense
    _check_against_oracle(G, oracle, [k3], [(v, ["sum", "count_all"])], expect_path="PARTITIONED_LDS")
    assert G.last_path.name == "PARTITIONED_LDS"


@pytest.mark.parametrize("log2p,nsplit,groups", [(4, 16, 40_000), (5, 8, 60_000), (6, 1, 200_000), (7, 2, 500_000), (8, 4, 1_000_000),
                                                 (9, 1, 1_000_000), (12, 1, 1_000_000), (16, 1, 4_000_000)])
def test_dense_ring_geometries(G, oracle, monkeypatch, log2p, nsplit, groups):
    """Ring scatter: every fan-out of one level (16 ... 256 partitions, a partition's regions shared out to 1 ... 16 aggregate
    workgroups whose table images are merged) and two levels (32 x 16, 64 x 64, 256 x 256)."""
    rng = np.random.default_rng(1000 + log2p)
    n = 4_400_000
    k = rng.integers(0, groups, n, dtype=np.int64) - 123_456_789
    v = rng.random(n)
    monkeypatch.setenv("CUDF_AMD_GB_DENSE_LOG2P", str(log2p))
    monkeypatch.setenv("CUDF_AMD_GB_DENSE_NSPLIT", str(nsplit))
    _check_against_oracle(G, oracle, [k], [(v, ["sum", "count_valid", "min", "max", "mean"])], expect_path="DENSE_DIRECT")


def test_dense_ring_key_outside_the_sampled_range(G, oracle):
    """One key the sample will not see lies outside the dense range: the ring scatter voids the attempt (overflow bit 2) and the
    call is redone by hash."""
    rng = np.random.default_rng(73)
    n = 5_000_000
    k = rng.integers(0, 400_000, n, dtype=np.int64)
    k[n // 3 + 777] = -(10**13)
    v = rng.random(n)
    _check_against_oracle(G, oracle, [k], [(v, ["sum", "count_all"])], expect_path="PARTITIONED_LDS")
    assert G.last_path.name == "PARTITIONED_LDS"


@pytest.mark.parametrize("ring", ["1", "0"])
@pytest.mark.parametrize("two_level", [False, True])
@pytest.mark.parametrize("shape", ["c4", "narrow", "three_keys"])
def test_dense_composite_keys(G, oracle, monkeypatch, shape, two_level, ring):
    """Composite dense keys: 1-4 integer key columns of any width (rows with a NULL key dropped under EXCLUDE), one value
    column of any type with nulls; the record is {mixed-radix index | validity, value}. One partition level, and two levels
    forced (CUDF_AMD_GB_DENSE_LOG2P=11: level 1 on the top 6 bits, level 2 on the next 5). Same call by hash agrees."""
    from oracle.oracle import HostColumn
    rng = np.random.default_rng(83)
    n = 4_300_000
    if shape == "c4":  # BASELINE config 4 in small: (int64, int32 with nulls) keys, float64 value with nulls, MEAN + MIN + MAX
        keys = [HostColumn(rng.integers(0, 700, n, dtype=np.int64), None, "int64"),
                HostColumn(rng.integers(-50, 50, n).astype(np.int32), rng.random(n) > 0.1, "int32")]
        vals, kinds = HostColumn(rng.random(n), rng.random(n) > 0.1, "float64"), ["mean", "min", "max"]
    elif shape == "narrow":
        keys = [HostColumn(rng.integers(0, 200, n).astype(np.uint8), None, "uint8"),
                HostColumn(rng.integers(-300, 300, n).astype(np.int16), None, "int16")]
        vals, kinds = HostColumn(rng.integers(-100, 100, n).astype(np.int32), rng.random(n) > 0.3, "int32"), ["sum", "count_valid", "count_all", "max"]
    else:
        keys = [HostColumn(rng.integers(10**12, 10**12 + 400, n, dtype=np.int64), None, "int64"),
                HostColumn(rng.integers(0, 30, n).astype(np.int8), rng.random(n) > 0.05, "int8"),
                HostColumn(rng.integers(0, 25, n).astype(np.uint32), None, "uint32")]
        vals, kinds = HostColumn(rng.random(n).astype(np.float32), None, "float32"), ["sum", "mean", "min"]
    monkeypatch.setenv("CUDF_AMD_GB_DENSE_RING", ring)
    if two_level:
        monkeypatch.setenv("CUDF_AMD_GB_DENSE_LOG2P", "11")
    _check_against_oracle(G, oracle, keys, [(vals, kinds)], expect_path="DENSE_DIRECT")
    if not two_level:
        monkeypatch.setenv("CUDF_AMD_GB_DENSE", "0")
        _check_against_oracle(G, oracle, keys, [(vals, kinds)])
        assert G.last_path.name != "DENSE_DIRECT"


@pytest.mark.parametrize("vt", ["float64", "int64"])
def test_skewed_dense_keys_on_two_levels(G, oracle, monkeypatch, vt):
    """Log-uniform ("Zipf") keys over 400K dense values with two ring levels forced: the 256 heaviest keys are aggregated inside
    the first level's workgroups (the tag32 instantiation of the heavy-hitter table), the regions of both levels are sized from
    the squared row shares of the remaining keys (estimate.cpp skew_m2), and the call stays on the direct-address tables."""
    from oracle.oracle import HostColumn
    rng = np.random.default_rng(71)
    n, groups = 6_000_000, 400_000
    k = np.clip((float(groups) ** rng.random(n)).astype(np.int64) - 1, 0, groups - 1) - 777
    v = rng.random(n) if vt == "float64" else rng.integers(-1000, 1000, n).astype(np.int64)
    monkeypatch.setenv("CUDF_AMD_GB_DENSE_LOG2P", "11")
    _check_against_oracle(G, oracle, [k], [(HostColumn(v, None, vt), ["sum", "count_valid"])], expect_path="DENSE_DIRECT")
    assert G.last_path.name == "DENSE_DIRECT"


def test_two_level_partition_forced(G, oracle, monkeypatch):
    """Forces the two-level radix partition (C4's regime) at a size the oracle can check."""
    rng = np.random.default_rng(13)
    n = 3_000_000
    k = rng.integers(0, 2_500_000, n, dtype=np.int64)
    v = rng.random(n)
    monkeypatch.setenv("CUDF_AMD_GB_LDS_KB", "4")  # tiny tables -> needs > 2048 partitions -> two levels
    try:
        _check_against_oracle(G, oracle, [k], [(v, ["sum", "count_all", "max"])], expect_path="PARTITIONED_LDS")
    finally:
        monkeypatch.delenv("CUDF_AMD_GB_LDS_KB")


def test_large_properties(G):
    """Size-independent properties at 100M rows (the oracle is too slow here): count sums to n, sum of sums
    matches a float64 reduction, every key appears exactly once, min <= mean <= max."""
    import torch
    import cudf_amd
    from cudf_amd import aggregation as agg, groupby as gb
    from cudf_amd.types import NullPolicy
    n, groups = 100_000_000, 1_000_000
    g = torch.Generator(device="cuda").manual_seed(42)
    k = torch.randint(0, groups, (n,), generator=g, device="cuda", dtype=torch.int64)
    v = torch.rand(n, generator=g, device="cuda", dtype=torch.float64)
    req = gb.GroupByRequest(cudf_amd.Column.from_torch(v), [agg.sum(), agg.count(NullPolicy.EXCLUDE), agg.min(), agg.max(), agg.mean()])
    grp = gb.GroupBy(cudf_amd.Table([cudf_amd.Column.from_torch(k)]))
    keys, res = grp.aggregate([req], stream=torch.cuda.current_stream())
    assert grp.last_path.name in ("PARTITIONED_LDS", "DENSE_DIRECT")
    kk = keys.columns()[0].to_numpy()[0]
    s, c, mn, mx, me = [x.to_numpy()[0] for x in res[0].columns()]
    assert len(kk) == len(np.unique(kk)) == int(torch.unique(k).numel())
    assert int(c.sum()) == n
    total = float(v.sum(dtype=torch.float64))
    assert abs(float(s.sum()) - total) <= 1e-9 * total
    assert np.all(mn <= me + 1e-15) and np.all(me <= mx + 1e-15)
    # spot-check 50 groups exactly against torch
    order = np.argsort(kk)
    for gi in order[:: max(1, len(order) // 50)][:50]:
        sel = v[k == int(kk[gi])]
        assert int(c[gi]) == sel.numel()
        assert abs(float(sel.sum(dtype=torch.float64)) - s[gi]) <= 4 * np.finfo(np.float64).eps * abs(float(sel.sum()) + s[gi]) * 8
        assert float(sel.min()) == mn[gi] and float(sel.max()) == mx[gi]


@pytest.mark.parametrize("shape", ["int64_key", "two_keys_nulls"])
def test_two_level_optimistic_partition(G, oracle, shape, monkeypatch):
    """More groups than 1024 LDS tables hold: two optimistic partition levels (level 2 reads level-1 regions as a strided
    list). 6M rows / ~3.8M groups against the oracle, for the 16-byte-record (write-combining) and the generic kernel;
    the exact two-level pipeline must give the same groups."""
    from oracle.oracle import HostColumn
    rng = np.random.default_rng(31)
    n = 6_000_000
    if shape == "int64_key":
        keys = [rng.integers(0, 6_000_000, n, dtype=np.int64)]
        vals = (rng.random(n), None)
        aggs = ["sum", "count_valid"]
    else:
        keys = [HostColumn(rng.integers(0, 3000, n, dtype=np.int64), None, "int64"),
                HostColumn(rng.integers(0, 2000, n).astype(np.int32), rng.random(n) > 0.1, "int32")]
        vals = (rng.random(n), rng.random(n) > 0.1)
        aggs = ["mean", "min", "max"]
    v = HostColumn(vals[0], vals[1], "float64")
    exp = kat.sort_groups(*oracle.groupby(keys, [(v, aggs)]))
    for opt2 in ("1", "0"):
        monkeypatch.setenv("CUDF_AMD_GB_OPTIMISTIC2", opt2)
        got = kat.sort_groups(*G.groupby(keys, [(v, aggs)]))
        for a, e in zip(got[0], exp[0]):
            kat.compare_columns(a, e, "keys")
        for name, a, e in zip(aggs, got[1][0], exp[1][0]):
            kat.compare_columns(a, e, name, atol=kat.sum_atol(16, 1.0))


def test_two_level_optimistic_overflow_falls_back(G, oracle):
    """Two optimistic levels with a heavy hitter: a third of the rows carry one key, its level-1 (or level-2) region
    overflows and the exact two-level pipeline repairs the call."""
    rng = np.random.default_rng(32)
    n = 6_000_000
    k = rng.integers(0, 6_000_000, n, dtype=np.int64)
    k[rng.random(n) < 0.33] = 123_456
    v = rng.random(n)
    _check_against_oracle(G, oracle, [k], [(v, ["sum", "count_all", "max"])], expect_path="PARTITIONED_LDS")


@pytest.mark.parametrize("vt", ["float64", "float32"])
def test_float_keys_output_a_representative_input_row(G, oracle, vt):
    """-0.0 == +0.0 and NaN == NaN group together (row equality, equality.cuh:59-89), but the OUTPUT key is one of the
    group's input rows, bit for bit (compute_groupby.cu:104-111): a group of only -0.0 comes back as -0.0, a group
    of NaNs as a NaN, a mixed +-0.0 group as either zero."""
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    npt = NP_OF_TYPE_ID[TYPE_ID[vt]]
    rng = np.random.default_rng(41)
    n = 50_000
    k = rng.integers(1, 200, n).astype(npt) * npt(0.5)
    k[::7] = npt(-0.0)              # a group of negative zeros only
    k[1::11] = npt("nan")
    second = rng.integers(0, 2, n).astype(np.int32)  # second key: (-0.0, 0) and (-0.0, 1) groups
    v = rng.random(n)
    kc, rc = G.groupby([HostColumn(k, None, vt), HostColumn(second, None, "int32")], [(HostColumn(v, None, "float64"), ["count_all", "sum"])])
    out_k = kc[0][0]
    zeros = out_k[out_k == 0]
    assert len(zeros) == 2 and np.signbit(zeros).all()                 # both zero groups hold only -0.0 rows
    assert np.isnan(out_k).sum() == 2                                   # the two NaN groups come back as NaN
    ek, er = oracle.groupby([HostColumn(k, None, vt), HostColumn(second, None, "int32")], [(HostColumn(v, None, "float64"), ["count_all", "sum"])])
    got, exp = kat.sort_groups(kc, rc), kat.sort_groups(ek, er)
    for a, e in zip(got[0], exp[0]):
        kat.compare_columns(a, e, "keys")
    kat.compare_columns(got[1][0][0], exp[1][0][0], "count_all")
    # a mixed group may return either zero
    k2 = np.array([0.0, -0.0, 0.0, 1.5], dtype=npt)
    kc2, rc2 = G.groupby([HostColumn(k2, None, vt)], [(HostColumn(np.ones(4), None, "float64"), ["count_all"])])
    o = np.argsort(kc2[0][0])
    assert kc2[0][0][o].tolist() == [0.0, 1.5] and rc2[0][0][0][o].tolist() == [3, 1]


_FUZZ_KEY_TYPES = ["int8", "int16", "int32", "int64", "uint32", "uint64", "float32", "float64", "bool"]
_FUZZ_VAL_TYPES = ["int8", "int32", "int64", "uint16", "uint64", "float32", "float64"]
_FUZZ_AGGS = ["sum", "count_valid", "count_all", "min", "max", "mean", "sum_of_squares", "product", "argmin", "argmax",
              "variance", "std", "m2"]


@pytest.mark.parametrize("seed", range(int(os.environ.get("CUDF_AMD_FUZZ_SEEDS", "60"))))
def test_fuzz_against_oracle(G, oracle, seed, monkeypatch):
    """Seeded random shapes: 1-3 key columns of mixed types (nullable or not), 1-2 value columns, a random subset of
    every engine aggregation, both null policies, sliced inputs, and sizes / table sizes that land on every path
    (single pass, merges, one and two partition levels)."""
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([0, 1, 37, 5_000, 80_000, 700_000]))
    if seed % 4 == 3:
        monkeypatch.setenv("CUDF_AMD_GB_LDS_KB", str(int(rng.choice([16, 24, 32]))))  # small tables: partition paths at small n
    # a third of the seeds: the big-input paths (dense keys through the ring scatter, heavy hitters, local pre-aggregation of
    # sorted keys) at sizes the oracle can check; half of those with the rows sorted by key
    big_paths = seed % 3 == 1
    if big_paths:
        monkeypatch.setenv("CUDF_AMD_GB_BIG_MIN_ROWS", "1000")
        n = int(rng.choice([80_000, 300_000, 700_000]))
        if seed % 4 != 3:
            monkeypatch.setenv("CUDF_AMD_GB_LDS_KB", str(int(rng.choice([16, 32, 159]))))  # (the groups must not fit one table)

    def column(tname, distinct, nullable, offset):
        npt = NP_OF_TYPE_ID[TYPE_ID[tname]]
        m = n + offset
        if tname == "bool":
            data = rng.integers(0, 2, m).astype(npt)
        elif np.dtype(npt).kind == "f":
            data = (rng.integers(0, distinct, m) - distinct // 2).astype(npt) * npt(0.25)
        else:
            info = np.iinfo(npt)
            data = rng.integers(0, min(distinct, int(info.max) - 1), m).astype(npt)
        valid = (rng.random(m) > 0.15) if nullable else None
        return HostColumn(data, valid, tname, offset=offset) if offset else HostColumn(data[:n] if offset == 0 else data, valid, tname)

    if os.environ.get("CUDF_AMD_FUZZ_LDS_KB"):  # (reproduce a seed under another table size)
        monkeypatch.setenv("CUDF_AMD_GB_LDS_KB", os.environ["CUDF_AMD_FUZZ_LDS_KB"])
    # (half of the big-path seeds keep to what the dense tables take: integer keys, one value column, no ARGMIN / ARGMAX, EXCLUDE)
    dense_shape = big_paths and seed % 6 == 1
    # (and half of THOSE draw 2-3 plain 8-byte value columns behind one plain 8-byte integer key: one value stream per column)
    dense_multi = dense_shape and seed % 12 == 1
    nkeys = 1 if dense_multi else (int(rng.integers(1, 3)) if dense_shape else int(rng.integers(1, 4)))
    spread = int(rng.choice([40, 3000, 60_000])) if big_paths else int(rng.choice([3, 40, 3000]))
    key_types = [t for t in _FUZZ_KEY_TYPES if not dense_shape or t.startswith(("int", "uint"))]
    if dense_multi:
        keys = [column(str(rng.choice(["int64", "uint64"])), spread, False, 0)]
    else:
        keys = [column(str(rng.choice(key_types)), spread, bool(rng.random() < 0.4), int(rng.choice([0, 0, 5]))) for _ in range(nkeys)]
    requests = []
    for _ in range(int(rng.integers(2, 4)) if dense_multi else (int(rng.integers(1, 4)) if dense_shape else int(rng.integers(1, 3)))):
        vt = str(rng.choice(["int64", "uint64", "float64"] if dense_multi else _FUZZ_VAL_TYPES))
        vals = column(vt, 9, False, 0) if dense_multi else column(vt, 9, bool(rng.random() < 0.5), int(rng.choice([0, 0, 3])))
        aggs = [a for a in _FUZZ_AGGS if not dense_shape or not a.startswith("arg")]
        kinds = [str(k) for k in rng.choice(aggs, size=int(rng.integers(1, 5)), replace=False)]
        requests.append((vals, kinds))
    include = bool(rng.random() < 0.5) and not dense_shape
    if big_paths and seed % 2 == 0 and n > 1:  # rows sorted by key (runs of equal keys): the pre-aggregation path
        order = np.lexsort([k.data[k.offset:k.offset + n] for k in keys][::-1], axis=0)
        type_name = {v: k for k, v in TYPE_ID.items()}

        def take(c):
            d = c.data[c.offset:c.offset + n][order]
            v = None if c.valid is None else c.valid[c.offset:c.offset + n][order]
            return HostColumn(d, v, type_name[c.type_id])
        keys = [take(k) for k in keys]
        requests = [(take(v), kinds) for v, kinds in requests]
    try:
        got = kat.sort_groups(*G.groupby(keys, requests, include_null_keys=include))
    except Exception as e:  # the engine's documented per-call limits are not what is fuzzed here
        if any(m in str(e) for m in ("Too many distinct accumulators", "Record too wide", "Key too wide", "Too many ARGMIN")):
            pytest.skip(str(e))
        raise
    exp = kat.sort_groups(*oracle.groupby(keys, requests, include_null_keys=include))
    assert len(got[0]) == len(exp[0])
    for a, e in zip(got[0], exp[0]):
        kat.compare_columns(a, e, "keys")
    eps = np.finfo(np.float64).eps
    for (vals, kinds), ra, re_ in zip(requests, got[1], exp[1]):
        is_f = np.dtype(vals.data.dtype).kind == "f"
        # per-group bounds from the rows actually present: m = valid rows of the group, vmax = largest magnitude
        cnt = kat.sort_groups(*oracle.groupby(keys, [(vals, ["count_valid"])], include_null_keys=include))[1][0][0][0]
        m = np.maximum(cnt.astype(np.float64), 1.0)
        vmax = float(np.max(np.abs(vals.data.astype(np.float64)))) if vals.size else 0.0
        for kind, a, e in zip(kinds, ra, re_):
            # The reference sums with unordered atomics, so two correct results differ by at most the worst-case bound of any
            # summation order, m^2 * eps * max|v| (kat.sum_atol) - per GROUP here, not per input size:
            if not is_f:
                atol = 0.0 if kind not in ("mean", "variance", "std", "m2") else None
            if kind in ("sum", "sum_of_squares") and is_f:
                atol = m * m * eps * max(vmax, vmax * vmax)
            elif kind == "mean":
                # sum error / count; an integer sum is exact and the division is one correctly rounded operation
                atol = (m * m * eps * vmax / m) if is_f else 0.0
            elif kind in ("m2", "variance", "std"):
                # M2 = ssq - s*s/n: the subtraction cancels, the absolute error stays within a few ulps of ssq <= m * vmax^2
                # (the bound test_variance_std_m2 uses); VAR divides it by (m - 1); |sqrt(a) - sqrt(b)| <= sqrt(|a - b|) for STD
                m2_atol = 8.0 * eps * m * vmax * vmax + (m * m * eps * vmax * vmax if is_f else 0.0)
                atol = m2_atol if kind == "m2" else m2_atol / np.maximum(m - 1.0, 1.0)
                if kind == "std":
                    atol = np.sqrt(atol)
            elif kind == "product" and is_f:
                atol = 2.0 * m * eps * np.maximum(1.0, vmax) ** np.minimum(m, 64.0)  # m roundings of a product of magnitude <= vmax^m
            elif is_f or kind in ("min", "max", "count_valid", "count_all", "argmin", "argmax"):
                atol = 0.0  # exact (float32 results: compare_columns allows 4 float32 ulps of the float64 reference)
            kat.compare_columns(a, e, f"{kind}({vals.type_id})", atol=atol)


@pytest.mark.parametrize("vt", ["float64", "int64"])
@pytest.mark.parametrize("hot_fraction", [0.01, 0.3, 0.9])
def test_heavy_hitters(G, oracle, vt, hot_fraction):
    """A few keys carry a large share of the rows: they are found in the sample and aggregated inside the scatter
    workgroups (an LDS table) instead of overflowing their regions / being aggregated by one workgroup. SUM, COUNT and
    MEAN must still be exact (integers) or within the usual float bound; the remaining keys are untouched."""
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(51)
    n = 6_000_000
    k = rng.integers(0, 900_000, n, dtype=np.int64) * 31 + 7
    hot = rng.random(n) < hot_fraction
    k[hot] = rng.integers(0, 12, int(hot.sum())) * 1_000_003 - 5  # 12 hot keys outside the cold range, one negative
    npt = NP_OF_TYPE_ID[TYPE_ID[vt]]
    v = rng.random(n).astype(npt) if vt == "float64" else rng.integers(-1000, 1000, n).astype(npt)
    _check_against_oracle(G, oracle, [k], [(HostColumn(v, None, vt), ["sum", "count_valid", "mean"])], expect_path="PARTITIONED_LDS")


def test_two_value_columns_at_1024_partitions(G, oracle):
    """Two value columns (24-byte records) over enough groups for 1024 partitions: the write-combining scatter does not fit there
    (granule 0), and the heavy-hitter set-up divided by that granule - SIGFPE on 1B rows x 1M groups. Also with three columns."""
    rng = np.random.default_rng(131)
    n = 4_500_000
    k = rng.integers(0, 1_200_000, n, dtype=np.int64) * 977 + 13  # sparse keys: the hash tables
    vals = [rng.random(n), rng.integers(-100, 100, n, dtype=np.int64), rng.random(n)]
    _check_against_oracle(G, oracle, [k], [(vals[0], ["sum", "mean"]), (vals[1], ["sum", "mean"])], expect_path="PARTITIONED_LDS")
    _check_against_oracle(G, oracle, [k], [(vals[0], ["sum"]), (vals[1], ["max", "count_valid"]), (vals[2], ["min"])], expect_path="PARTITIONED_LDS")


@pytest.mark.parametrize("groups", [700, 6_000, 250_000, 1_000_000])
@pytest.mark.parametrize("plan", ["sum_count_sum", "two_sums", "three_means", "mixed_types", "minmax_var"])
def test_dense_keys_several_value_columns(G, oracle, monkeypatch, plan, groups):
    """One plain int64 key column spanning a small range and TWO or THREE plain 8-byte value columns, n >= 4M: the dense path carries
    one value stream per column (ring scatter: dense_multi_kernels.hip; few groups: one direct-address table per workgroup with
    every column's accumulators) instead of falling to 24- / 32-byte hash records. The same call with CUDF_AMD_GB_DENSE_MULTI=0
    must take the hash tables and agree."""
    rng = np.random.default_rng(500 + groups % 97)
    n = 4_300_000
    k = rng.integers(0, groups, n, dtype=np.int64) - 77_000
    a, b, c = rng.random(n), rng.random(n) * 3.0 - 1.0, rng.integers(-1000, 1000, n, dtype=np.int64)
    u = rng.integers(0, 1 << 40, n, dtype=np.uint64)
    requests = {"sum_count_sum": [(a, ["sum", "count_valid"]), (b, ["sum"])],
                "two_sums": [(a, ["sum"]), (b, ["sum"])],
                "three_means": [(a, ["mean"]), (b, ["mean"]), (c, ["mean", "sum"])],
                "mixed_types": [(c, ["sum", "min"]), (u, ["max", "count_all"]), (a, ["sum", "max"])],
                "minmax_var": [(a, ["min", "max", "sum_of_squares"]), (b, ["mean", "count_all", "max"])]}[plan]
    # (three 8-byte accumulators of ONE column over 8192 slots do not fit a table: that plan keeps the hash tables at 1M groups)
    too_wide = plan == "minmax_var" and groups >= 1_000_000
    _check_against_oracle(G, oracle, [k], requests, expect_path=None if too_wide else "DENSE_DIRECT")
    assert too_wide or G.last_path.name == "DENSE_DIRECT"
    monkeypatch.setenv("CUDF_AMD_GB_DENSE_MULTI", "0")
    _check_against_oracle(G, oracle, [k], requests)
    assert G.last_path.name != "DENSE_DIRECT"


def test_dense_keys_several_value_columns_fall_back(G, oracle):
    """Two value columns on the dense path: a key outside the sampled range voids the ring attempt (the call is redone by hash);
    rows sorted by key take the local pre-aggregation; a repeated value column is ONE column (the single-column dense path)."""
    rng = np.random.default_rng(577)
    n = 4_300_000
    k = rng.integers(0, 400_000, n, dtype=np.int64)
    a, b = rng.random(n), rng.random(n)
    k2 = k.copy()
    k2[n // 3 + 777] = 10**13
    _check_against_oracle(G, oracle, [k2], [(a, ["sum", "count_valid"]), (b, ["sum"])])
    assert G.last_path.name == "PARTITIONED_LDS"
    ks = np.sort(k)
    _check_against_oracle(G, oracle, [ks], [(a, ["sum", "count_valid"]), (b, ["sum", "max"])])
    _check_against_oracle(G, oracle, [k], [(a, ["sum"]), (a, ["count_valid", "max"])], expect_path="DENSE_DIRECT")


@pytest.mark.parametrize("shape", ["plain", "int32_key_nullable_value", "two_keys", "sorted", "one_hot_key", "outlier"])
def test_dense_keys_one_table(G, oracle, monkeypatch, shape):
    """Few groups over a small key range, n >= 4M: every workgroup aggregates its rows straight from the columns into a direct-address
    table of its own (path T of groupby.cpp), the images are merged. A wave that meets in one slot (sorted rows, one hot key) is
    reduced across the wave; a key outside the sampled range sends the call to the hash tables."""
    from oracle.oracle import HostColumn
    rng = np.random.default_rng(111)
    n = 4_400_000
    kinds = ["sum", "count_valid", "count_all", "min", "max", "mean"]
    expect = "DENSE_DIRECT"
    if shape == "plain":
        keys, vals, kinds = [rng.integers(-300, 300, n, dtype=np.int64)], HostColumn(rng.random(n), None, "float64"), ["sum", "count_valid"]
    elif shape == "int32_key_nullable_value":
        keys = [HostColumn(rng.integers(0, 900, n).astype(np.int32), None, "int32")]
        vals = HostColumn(rng.integers(-50, 50, n).astype(np.int16), rng.random(n) > 0.3, "int16")
    elif shape == "two_keys":
        keys = [HostColumn(rng.integers(0, 40, n).astype(np.uint8), None, "uint8"), HostColumn(rng.integers(-20, 20, n).astype(np.int64), rng.random(n) > 0.1, "int64")]
        vals = HostColumn(rng.random(n), rng.random(n) > 0.1, "float64")
    elif shape == "sorted":
        keys, vals = [np.arange(n, dtype=np.int64) // 9000], HostColumn(rng.random(n), None, "float64")
    elif shape == "one_hot_key":
        k = rng.integers(0, 2000, n, dtype=np.int64)
        k[rng.random(n) < 0.7] = 1234
        keys, vals, kinds = [k], HostColumn(rng.integers(-9, 9, n, dtype=np.int64), None, "int64"), ["sum", "count_all"]
    else:
        k = rng.integers(0, 700, n, dtype=np.int64)
        k[n // 2 + 5] = 10**9  # one key the sample will not see
        keys, vals, kinds, expect = [k], HostColumn(rng.random(n), None, "float64"), ["sum", "count_valid"], None
    _check_against_oracle(G, oracle, keys, [(vals, kinds)], expect_path=expect)
    if shape == "outlier":
        assert G.last_path.name != "DENSE_DIRECT"
    monkeypatch.setenv("CUDF_AMD_GB_DENSE_ONE_TABLE", "0")
    _check_against_oracle(G, oracle, keys, [(vals, kinds)])


@pytest.mark.parametrize("shape", ["sorted", "clustered", "sorted_nulls_two_keys", "float_keys"])
def test_sorted_and_clustered_keys_are_preaggregated(G, oracle, monkeypatch, shape):
    """Most rows are followed by a row of the same key (the estimate pass measures it): row chunks are aggregated locally first and
    their partial records take the exact partition pipeline (path A of groupby.cpp). Same call with CUDF_AMD_GB_PREAGG=0 agrees."""
    from oracle.oracle import HostColumn
    rng = np.random.default_rng(91)
    n = 4_600_000
    i = np.arange(n, dtype=np.int64)
    if shape == "sorted":
        keys = [i // 37 - 1000]
        vals = HostColumn(rng.random(n), None, "float64")
    elif shape == "clustered":  # runs of 24 equal keys, every key recurs about 20 times
        keys = [((i // 24) * 2654435761) % 9_973]
        vals = HostColumn(rng.integers(-1000, 1000, n).astype(np.int32), rng.random(n) > 0.2, "int32")
    elif shape == "sorted_nulls_two_keys":
        keys = [HostColumn(i // 4000, None, "int64"), HostColumn(((i // 40) % 100).astype(np.int16), (i // 40) % 7 != 0, "int16")]
        vals = HostColumn(rng.random(n), rng.random(n) > 0.1, "float64")
    else:  # float keys: +0.0 / -0.0 and NaNs meet, the output key is a representative input row
        f = (i // 50).astype(np.float64)
        f[(i // 50) % 11 == 0] = np.nan
        f[(i // 50) % 13 == 0] = -0.0
        keys = [HostColumn(f, None, "float64")]
        vals = HostColumn(rng.random(n), None, "float64")
    kinds = ["sum", "count_valid", "count_all", "min", "max", "mean"]
    _check_against_oracle(G, oracle, keys, [(vals, kinds)], expect_path="PARTITIONED_LDS")
    monkeypatch.setenv("CUDF_AMD_GB_PREAGG", "0")
    _check_against_oracle(G, oracle, keys, [(vals, kinds)])


@pytest.mark.parametrize("shape", ["runs_of_64", "ragged_runs", "int_values_sum_of_squares", "two_value_columns", "few_long_runs"])
def test_plain_runs_collapse_without_a_table(G, oracle, monkeypatch, shape):
    """Plain shapes (one int64 key, 8-byte values, no NULLs) with clustered rows take the run-collapsing front end of path A
    (collapse_runs.hip): a segmented scan over the lanes, one partial record per run and row set, no LDS table. Runs of exactly 64
    rows whose keys recur; runs of 1..200 rows (unaligned, a row count that is not a multiple of the batch); int64 values with
    SUM of squares / MIN / MAX; two value columns; a handful of very long runs. The same call with the front end switched off
    (CUDF_AMD_GB_COLLAPSE_RUNS=0: one LDS table per row chunk) agrees."""
    from oracle.oracle import HostColumn
    rng = np.random.default_rng({"runs_of_64": 1, "ragged_runs": 2, "int_values_sum_of_squares": 3, "two_value_columns": 4, "few_long_runs": 5}[shape])
    n = 4_700_003
    i = np.arange(n, dtype=np.int64)
    requests = None
    if shape == "runs_of_64":
        keys = [((i // 64) * 2654435761) % 99_991 - 50_000]
        requests = [(HostColumn(rng.random(n), None, "float64"), ["sum", "count_valid", "mean"])]
    elif shape == "ragged_runs":
        lens = rng.integers(1, 200, n // 50)
        k = np.repeat(rng.integers(-2**40, 2**40, len(lens)), lens)[:n]
        k = np.concatenate([k, np.full(n - len(k), 7, dtype=np.int64)]) if len(k) < n else k
        keys = [k.astype(np.int64)]
        requests = [(HostColumn(rng.random(n), None, "float64"), ["sum", "min", "max", "count_all"])]
    elif shape == "int_values_sum_of_squares":
        keys = [i // 90 * 3]
        requests = [(HostColumn(rng.integers(-3000, 3000, n).astype(np.int64), None, "int64"), ["sum", "sum_of_squares", "min", "max"])]
    elif shape == "two_value_columns":
        keys = [((i // 48) * 40503) % 20_011]
        requests = [(HostColumn(rng.random(n), None, "float64"), ["sum", "count_valid"]),
                    (HostColumn(rng.integers(-10**6, 10**6, n).astype(np.int64), None, "int64"), ["sum", "max"])]
    else:
        keys = [np.minimum(i // 1_000_000, 3)]
        requests = [(HostColumn(rng.random(n), None, "float64"), ["sum", "count_valid", "min"])]
    from cudf_amd import _lib
    _lib.profile_reset()
    _lib.profile_enable(True)
    try:
        _check_against_oracle(G, oracle, keys, requests, expect_path=None if shape == "few_long_runs" else "PARTITIONED_LDS")
    finally:
        _lib.profile_enable(False)
    if shape != "few_long_runs":  # (four groups fit one table per workgroup: no pre-aggregation)
        assert "collapse_runs" in _lib.profile_report(), _lib.profile_report()
    monkeypatch.setenv("CUDF_AMD_GB_COLLAPSE_RUNS", "0")
    _check_against_oracle(G, oracle, keys, requests)


@pytest.mark.parametrize("vt", ["float64", "int64"])
@pytest.mark.parametrize("hot_fraction", [0.01, 0.3, 0.9])
def test_heavy_hitters_dense_keys(G, oracle, vt, hot_fraction):
    """The same over dense keys: the call stays on the direct-address tables, the ring scatter keeps the hot keys in its own LDS
    table and their merged partials come back as one more item behind the tables'."""
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(52)
    n = 6_000_000
    k = rng.integers(0, 900_000, n, dtype=np.int64) - 450_000
    hot = rng.random(n) < hot_fraction
    k[hot] = rng.integers(0, 12, int(hot.sum())) * 70_001 - 400_000  # 12 hot keys inside the range
    npt = NP_OF_TYPE_ID[TYPE_ID[vt]]
    v = rng.random(n).astype(npt) if vt == "float64" else rng.integers(-1000, 1000, n).astype(npt)
    # (with 90 % of the rows on 12 keys the sample shows few of the cold keys, the key range looks sparse against the estimated
    # group count and the call takes the hash tables: either path is right there)
    _check_against_oracle(G, oracle, [k], [(HostColumn(v, None, vt), ["sum", "count_valid", "mean"])],
                          expect_path="DENSE_DIRECT" if hot_fraction < 0.5 else "PARTITIONED_LDS")


@pytest.mark.parametrize("vt", ["int8", "int16", "int32", "int64", "decimal32", "decimal64"])
@pytest.mark.parametrize("n,groups", [(20_000, 37), (600_000, 90_000)])
def test_sum_overflow_against_oracle(G, oracle, vt, n, groups):
    """SUM_OVERFLOW -> struct {sum: source type, overflow: bool} (reference device_aggregators.cuh:136-160). Values of one
    sign per group, so that a group overflows in every order of the additions or in none: the reference's flag depends on
    the arrival order of its atomics otherwise. Sums compared where the group did not overflow, flags everywhere."""
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(61)
    npt = NP_OF_TYPE_ID[TYPE_ID[vt]]
    info = np.iinfo(npt)
    k = rng.integers(0, groups, n).astype(np.int32)
    sign = np.where((k % 3) == 0, -1, 1)
    per_group = max(1, n // groups)
    mag = rng.integers(0, max(2, int(info.max // per_group) * 2), n, dtype=np.int64)  # about half the groups overflow
    v = (sign * mag).astype(np.int64).clip(info.min, info.max).astype(npt)
    vv = rng.random(n) > 0.2
    _check_against_oracle(G, oracle, [k], [(HostColumn(v, vv, vt), ["sum_overflow", "count_valid"])])
    _check_against_oracle(G, oracle, [k], [(HostColumn(v, None, vt), ["sum_overflow"])])


@pytest.mark.parametrize("ddof", [0, 1, 3])
def test_variance_std_ddof(G, ddof):
    """ddof travels through the C ABI (cudf_amd_aggregation_request.params; make_variance_aggregation(ddof),
    reference aggregation.hpp:259-266): VAR = M2 / (count - ddof), null when count - ddof <= 0."""
    import cudf_amd
    from cudf_amd import aggregation as agg, groupby as gb
    rng = np.random.default_rng(5)
    n = 50_000
    k = rng.integers(0, 500, n).astype(np.int32)
    k[:3] = [10_000, 10_000, 10_001]  # groups of 2 rows and 1 row
    v = rng.normal(size=n)
    req = gb.GroupByRequest(cudf_amd.Column.from_numpy(v), [agg.variance(ddof), agg.std(ddof), agg.count()])
    keys, res = gb.GroupBy(cudf_amd.Table([cudf_amd.Column.from_numpy(k)])).aggregate([req])
    kk = keys.columns()[0].to_numpy()[0]
    (var, var_valid), (std, std_valid), (cnt, _) = [c.to_numpy() for c in res[0].columns()]
    for g, key in enumerate(kk):
        x = v[k == key]
        if len(x) - ddof <= 0:
            assert not var_valid[g] and not std_valid[g]
            continue
        assert var_valid[g] and std_valid[g]
        np.testing.assert_allclose(var[g], np.var(x, ddof=ddof), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(std[g], np.std(x, ddof=ddof), rtol=1e-9, atol=1e-12)


def test_all_nan_group_min_max_argmin_argmax_pinned(G):
    """The chosen behaviour for a group whose valid values are all NaN (the reference pins only
    termination, max_tests.cpp:526-549, and its CAS loops leave the outcome to the arrival order): a NaN never replaces a
    number, so MIN / MAX stay at their identities (+inf / -inf), and ARGMIN / ARGMAX, which look for a row that attains the
    extreme, find none and return -1. A group with one number among NaNs returns that number and its row."""
    import cudf_amd
    from cudf_amd import aggregation as agg, groupby as gb
    k = np.array([1, 1, 2, 2, 2, 3], np.int32)
    v = np.array([np.nan, np.nan, np.nan, 5.0, np.nan, 7.0])
    req = gb.GroupByRequest(cudf_amd.Column.from_numpy(v), [agg.min(), agg.max(), agg.argmin(), agg.argmax()])
    keys, res = gb.GroupBy(cudf_amd.Table([cudf_amd.Column.from_numpy(k)])).aggregate([req])
    kk = keys.columns()[0].to_numpy()[0]
    order = np.argsort(kk)
    mn, mx, amn, amx = [c.to_numpy()[0][order] for c in res[0].columns()]
    assert kk[order].tolist() == [1, 2, 3]
    assert mn[0] == np.inf and mx[0] == -np.inf and amn[0] == -1 and amx[0] == -1
    assert mn[1:].tolist() == [5.0, 7.0] and mx[1:].tolist() == [5.0, 7.0]
    assert amn[1:].tolist() == [3, 5] and amx[1:].tolist() == [3, 5]


@pytest.mark.parametrize("include", [False, True])
@pytest.mark.parametrize("ktype,vtype", [("int64", "float64"), ("int32", "int32"), ("uint8", "float32"), ("int16", "int64"), ("int64", "bool"),
                                         ("int32", "timestamp_ms")])
def test_argmin_argmax_by_lookup(G, oracle, monkeypatch, ktype, vtype, include):
    """ARGMIN / ARGMAX over one integer key column of a small range are answered as MIN / MAX + one lookup pass over the rows (round 4,
    arg_lookup.hip) from CUDF_AMD_GB_BIG_MIN_ROWS rows on: the smallest row among those that attain the extreme (ties on purpose: values from
    a small pool), NULL keys (their own group under INCLUDE), NULL values, a group without a valid value (null), NaN / -0.0 / +0.0 among the
    floats, negative keys, other kinds and a repeated ARGMIN in the same call - against the oracle and against the engine's own ARGMIN / ARGMAX."""
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    rng = np.random.default_rng(["int64", "int32", "uint8", "int16"].index(ktype) * 16 + len(vtype) + (7 if include else 0))
    n = 60_000
    kdt, vdt = NP_OF_TYPE_ID[TYPE_ID[ktype]], NP_OF_TYPE_ID[TYPE_ID[vtype]]
    klo = -40 if np.dtype(kdt).kind == "i" else 0
    k = rng.integers(klo, klo + 97, n).astype(kdt)
    kvalid = rng.random(n) > 0.05
    vvalid = rng.random(n) > 0.2
    vvalid[k == k[0]] = False  # one group without any valid value
    keys = [HostColumn(k, kvalid, ktype)]
    monkeypatch.setenv("CUDF_AMD_GB_BIG_MIN_ROWS", "1000")
    # with_nan: a NaN among a group's values makes the REFERENCE's answer depend on the arrival order of its atomics (the oracle follows row
    # order: a group whose first valid value is NaN keeps that row) - such data is compared with the engine's own ARGMIN / ARGMAX only
    for with_nan in (False, True):
        if np.dtype(vdt).kind == "f":
            pool = np.concatenate([rng.normal(size=20), [-0.0, 0.0, np.inf, -np.inf] + ([np.nan] if with_nan else [])]).astype(vdt)
            v = pool[rng.integers(0, len(pool), n)]
        elif np.dtype(vdt).kind == "b":
            v = rng.integers(0, 2, n).astype(np.uint8)
        else:
            v = rng.integers(-7, 8, n).astype(vdt)
        vals = HostColumn(v, vvalid, vtype)
        requests = [(vals, ["argmin", "argmax", "min", "count_valid", "argmin"]),
                    (HostColumn(v, None, vtype), ["argmax", "sum" if vtype != "timestamp_ms" else "max"])]
        monkeypatch.delenv("CUDF_AMD_GB_ARG_LOOKUP", raising=False)
        if not with_nan:
            _check_against_oracle(G, oracle, keys, requests, include=include)
        by_lookup = kat.sort_groups(*G.groupby(keys, requests, include_null_keys=include))
        monkeypatch.setenv("CUDF_AMD_GB_ARG_LOOKUP", "0")
        by_engine = kat.sort_groups(*G.groupby(keys, requests, include_null_keys=include))
        for (_, kinds), ra, rb in zip(requests, by_lookup[1], by_engine[1]):
            for kind, a, b in zip(kinds, ra, rb):
                if kind != "sum":  # (float sums: order-dependent; the oracle comparison above bounds them)
                    kat.compare_columns(a, b, f"lookup against the engine [{kind}]")
