"""GPU parity tests for the sort-based groupby (cudf_amd/csrc/groupby/sort_groupby.hip) through the C ABI: the kinds the hash
tables cannot serve (NTH_ELEMENT, NUNIQUE, MEDIAN, QUANTILE; reference cpp/src/groupby/sort/) - the reference's own KATs, then
seeded random inputs against the CPU restatement (oracle/sort_groupby.py), then properties at sizes the restatement is too slow
for."""
import numpy as np
import pytest

import kat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(gpu):
    import gpu_backend
    return gpu_backend


@pytest.fixture(scope="module")
def SG():
    from oracle import sort_groupby
    return sort_groupby


@pytest.mark.parametrize("name,c,kt,vt", list(kat.sort_groupby_cases()), ids=[x[0] for x in kat.sort_groupby_cases()])
def test_sort_groupby_kat(G, name, c, kt, vt):
    kat.run_sort_groupby_case(G, c, kt, vt)
    if len(c["keys"]) and isinstance(c["agg"], dict):  # (a pre-sorted-keys case with a hash kind is answered by the hash path where runs == keys)
        assert G.last_path.name == "SORT"
    if c["name"] == "pre_sorted_keys_nulls_before_include_nulls":  # the claim is wrong there: the call is redone on the runs
        assert G.last_path.name == "SORT"


SORT_KINDS = [{"kind": "nth_element", "n": 0}, {"kind": "nth_element", "n": 1, "null_policy": "exclude"},
              {"kind": "nth_element", "n": -1}, {"kind": "nth_element", "n": -2, "null_policy": "exclude"},
              {"kind": "nunique"}, {"kind": "nunique", "null_policy": "include"}, {"kind": "median"},
              {"kind": "quantile", "quantiles": [0.0, 0.1, 0.5, 0.9, 1.0]},
              {"kind": "quantile", "quantiles": [0.33], "interpolation": "lower"},
              {"kind": "quantile", "quantiles": [0.33], "interpolation": "higher"},
              {"kind": "quantile", "quantiles": [0.5, 0.37], "interpolation": "midpoint"},
              {"kind": "quantile", "quantiles": [0.5, 0.125], "interpolation": "nearest"},
              {"kind": "quantile", "quantiles": [0.5, 0.125], "interpolation": "nearest_half_up"}]


def _compare(got, exp, float_rtol=0.0):
    """float_rtol: order-dependent float results of the hash kinds (sums of squares, variance ... accumulated by unordered atomics on the
    device, in row order by oracle.c) are compared to this relative error of the column's largest magnitude instead of 4 ulps."""
    (gk, gr), (ek, er) = got, exp
    assert len(gk) == len(ek)
    for a, e in zip(gk, ek):
        kat.compare_columns(a, e, "keys")
    for r, (ga, ea) in enumerate(zip(gr, er)):
        assert len(ga) == len(ea)
        for j, (a, e) in enumerate(zip(ga, ea)):
            atol = 0.0
            if float_rtol and not isinstance(e[0], tuple) and np.asarray(e[0]).dtype.kind == "f" and len(e[0]):
                finite = np.asarray(e[0])[np.isfinite(e[0])]
                atol = float_rtol * (float(np.abs(finite).max()) if len(finite) else 0.0)
            kat.compare_columns(a, e, f"request {r} aggregation {j}", atol)


def _random_column(rng, n, type_name, distinct, null_share):
    from oracle.oracle import HostColumn, NP_OF_TYPE_ID, TYPE_ID
    npt = NP_OF_TYPE_ID[TYPE_ID[type_name]]
    if np.dtype(npt).kind == "f":
        pool = rng.normal(size=distinct).astype(npt)
        pool[: min(4, distinct)] = np.array([np.nan, -0.0, 0.0, np.inf], npt)[: min(4, distinct)]
        if distinct > 5:
            pool[4] = -np.inf
            pool[5] = np.float64(np.nan).astype(npt)  # a second NaN: all NaNs are one key
    elif np.dtype(npt).kind == "b":
        pool = np.array([False, True])
    else:
        info = np.iinfo(npt)
        pool = rng.integers(info.min, info.max, size=distinct, dtype=np.int64 if np.dtype(npt).kind == "i" else np.uint64,
                            endpoint=True).astype(npt)
        pool[: min(2, distinct)] = np.array([info.min, info.max], npt)[: min(2, distinct)]
    data = pool[rng.integers(0, len(pool), n)]
    valid = None if null_share == 0 else rng.random(n) >= null_share
    return HostColumn(data, valid, type_name)


@pytest.mark.parametrize("key_types", [("int32",), ("int64",), ("float64",), ("float32", "int8"), ("uint16", "int64", "bool")])
@pytest.mark.parametrize("include", [False, True])
def test_random_against_the_restatement(G, SG, key_types, include):
    rng = np.random.default_rng(sum(ord(c) for c in "".join(key_types)) + (5000 if include else 0))  # (not hash(): strings hash per process)
    n = 20011
    keys = [_random_column(rng, n, t, 9 if len(key_types) > 1 else 301, 0.07) for t in key_types]
    v1 = _random_column(rng, n, "int64", 50, 0.2)
    v2 = _random_column(rng, n, "float64", 1000, 0.1)
    v3 = _random_column(rng, n, "int16", 40, 0.0)
    requests = [(v1, SORT_KINDS + ["sum", "min", "count_valid"]), (v2, SORT_KINDS[:8] + ["max", "count_all"]),
                (v3, ["mean", {"kind": "median"}, {"kind": "nunique"}, {"kind": "nth_element", "n": 2}])]
    got = G.groupby(keys, requests, include_null_keys=include)
    assert G.last_path.name == "SORT"
    _compare(got, SG.groupby(keys, requests, include_null_keys=include))


def test_every_value_type(G, SG):
    rng = np.random.default_rng(5)
    n = 5003
    keys = [_random_column(rng, n, "int32", 37, 0.05)]
    for vt in ["int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64", "bool"]:
        v = _random_column(rng, n, vt, 23, 0.15)
        requests = [(v, SORT_KINDS)]
        _compare(G.groupby(keys, requests), SG.groupby(keys, requests))
    for vt in ["timestamp_ms", "duration_s", "decimal64", "timestamp_days"]:  # no quantiles of these (group_quantiles.cu:125-131)
        v = _random_column(rng, n, vt, 23, 0.15)
        requests = [(v, SORT_KINDS[:6])]
        _compare(G.groupby(keys, requests), SG.groupby(keys, requests))


def test_presorted_keys_are_not_sorted_again(G, SG):
    """keys_are_sorted = YES: groups are the runs of equal adjacent keys (sort_helper.cu:81-88), whatever their order."""
    from oracle.oracle import HostColumn
    k = HostColumn(np.repeat(np.array([5, 3, 9, 3, 1], np.int32), [4, 1, 7, 2, 3]))
    v = HostColumn(np.arange(17, dtype=np.float64))
    requests = [(v, [{"kind": "median"}, {"kind": "nth_element", "n": -1}, "sum"])]
    got = G.groupby([k], requests, keys_are_sorted=True)
    exp = SG.groupby([k], requests, keys_are_sorted=True)
    assert list(got[0][0][0]) == [5, 3, 9, 3, 1]
    _compare(got, exp)
    # with nulls to drop the keys are sorted after all (sort_helper.cu:51-54)
    k = HostColumn(np.array([4, 4, 2, 2, 7], np.int32), np.array([1, 1, 0, 1, 1], bool))
    v = HostColumn(np.arange(5, dtype=np.int32))
    requests = [(v, [{"kind": "nth_element", "n": 0}])]
    got = G.groupby([k], requests, keys_are_sorted=True)
    assert list(got[0][0][0]) == [2, 4, 7] and list(got[1][0][0][0]) == [3, 0, 4]


def test_sorted_path_skips_the_digits_all_keys_share(G, SG):
    """Small-range int64 keys, one group, all-equal values, one row: the passes that would move nothing are skipped."""
    from oracle.oracle import HostColumn
    rng = np.random.default_rng(11)
    for n, hi in [(1, 1), (2, 1), (4097, 1), (4096, 3), (70001, 70000)]:
        k = HostColumn(rng.integers(0, hi, n).astype(np.int64) + (1 << 40))
        v = HostColumn(np.full(n, 7, np.int32) if hi == 1 else rng.integers(-5, 5, n).astype(np.int32))
        requests = [(v, [{"kind": "median"}, {"kind": "nunique"}, {"kind": "nth_element", "n": -1}, "count_all"])]
        _compare(G.groupby([k], requests), SG.groupby([k], requests))


def test_all_keys_excluded(G):
    from oracle.oracle import HostColumn
    k = HostColumn(np.arange(100, dtype=np.int32), np.zeros(100, bool))
    v = HostColumn(np.arange(100, dtype=np.int32))
    kc, rc = G.groupby([k], [(v, [{"kind": "median"}, "sum", {"kind": "nunique"}, "sum_overflow"])])
    assert len(kc[0][0]) == 0 and [len(c[0]) if not isinstance(c[0], tuple) else len(c[0][0]) for c in rc[0]] == [0, 0, 0, 0]
    assert [c[2] for c in rc[0][:3]] == [10, 4, 3]


def test_hash_kinds_ride_along_with_every_result_shape(G, SG):
    """One sort kind takes the call down the sort path (groupby.cu:64-69); the hash kinds of the same call are answered on the
    group labels and come back aligned to the sorted keys, nulls, structs (SUM_OVERFLOW) and row indices (ARGMAX) included."""
    rng = np.random.default_rng(3)
    n = 30000
    keys = [_random_column(rng, n, "int32", 500, 0.1)]
    v = _random_column(rng, n, "int64", 100000, 0.5)
    # non-negative terms, huge in every second key: a group's SUM_OVERFLOW flag is then the same in every order of the additions
    # (oracle.c follows row order, the device any order)
    big = (keys[0].data.astype(np.int64) & 1) == 1
    v.data[:] = np.where(big, np.abs(v.data >> 2), np.abs(v.data) % 1000)
    w = _random_column(rng, n, "float64", 100000, 0.3)
    # (two calls: one hash-groupby call holds at most 12 distinct accumulators, plan.cpp)
    for requests in ([(v, ["sum", "sum_overflow", "min", "max", "count_valid", "count_all", "mean", "argmax", "argmin", {"kind": "nunique"}])],
                     [(w, ["variance", "std", "m2", "sum_of_squares", "product", {"kind": "median"}]), (v, ["max", {"kind": "nth_element", "n": -1}])]):
        for include in (False, True):
            _compare(G.groupby(keys, requests, include_null_keys=include), SG.groupby(keys, requests, include_null_keys=include), float_rtol=1e-9)


def test_median_at_ten_million_rows(G):
    """Properties the size does not change: sorted distinct keys, and per group the median of numpy on the same rows."""
    from oracle.oracle import HostColumn
    rng = np.random.default_rng(17)
    n = 10_000_000
    k = rng.integers(0, 1000, n).astype(np.int32)
    v = rng.normal(size=n)
    kc, rc = G.groupby([HostColumn(k)], [(HostColumn(v), [{"kind": "median"}, {"kind": "nth_element", "n": 0}, {"kind": "nunique"}, "count_all"])])
    assert G.last_path.name == "SORT"
    assert np.array_equal(kc[0][0], np.arange(1000))
    order = np.argsort(k, kind="stable")
    ks, vs = k[order], v[order]
    bounds = np.searchsorted(ks, np.arange(1001))
    med = np.array([np.median(vs[bounds[g]:bounds[g + 1]]) for g in range(1000)])
    assert kat.equivalent(rc[0][0][0], med, True)
    assert np.array_equal(rc[0][1][0], vs[bounds[:-1]])
    assert np.array_equal(rc[0][2][0], np.diff(bounds))  # normal draws: all distinct
    assert np.array_equal(rc[0][3][0], np.diff(bounds))
