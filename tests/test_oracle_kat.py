"""CPU: pins the oracle (oracle/oracle.c) against the reference's own known-answer tests (tests/golden/*.json,
transcribed from cpp/tests/groupby/*.cpp and cpp/tests/join/join_tests.cpp) and the public murmur3 vectors."""
import numpy as np
import pytest

import kat
from oracle import oracle as O


@pytest.mark.parametrize("name,c,kt,vt", list(kat.groupby_cases()), ids=[x[0] for x in kat.groupby_cases()])
def test_groupby_kat(name, c, kt, vt):
    kat.run_groupby_case(O, c, kt, vt)


@pytest.mark.parametrize("name,c,kt,vt", list(kat.sort_groupby_cases()), ids=[x[0] for x in kat.sort_groupby_cases()])
def test_sort_groupby_kat(name, c, kt, vt):
    """The kinds only the reference's sort-based groupby serves (oracle/sort_groupby.py against nth_element_tests.cpp,
    nunique_tests.cpp, median_tests.cpp, quantile_tests.cpp)."""
    from oracle import sort_groupby as SG
    kat.run_sort_groupby_case(SG, c, kt, vt)


def test_groupby_generated_kats():
    # max_tests.cpp:554-574 — 512 unique keys, keys == values
    k = np.arange(512, dtype=np.int32)
    kc, rc = kat.sort_groups(*O.groupby([k], [(k, ["max"])]))
    assert np.array_equal(kc[0][0], k) and np.array_equal(rc[0][0][0], k)
    # max_tests.cpp:576-597 — 128 keys x 10000, shuffled
    rng = np.random.default_rng(0)
    k = np.tile(np.arange(128, dtype=np.int32), 10000)
    rng.shuffle(k)
    kc, rc = kat.sort_groups(*O.groupby([k], [(k, ["max"])]))
    assert np.array_equal(kc[0][0], np.arange(128)) and np.array_equal(rc[0][0][0], np.arange(128))
    # keys_tests.cpp:355-409 — duplicate aggregations / duplicate columns
    K10 = np.array([1, 2, 3, 1, 2, 2, 1, 3, 3, 2], np.int32)
    V10 = np.arange(10, dtype=np.int32)
    kc, rc = kat.sort_groups(*O.groupby([K10], [(V10, ["sum", "sum"])]))
    assert [list(c[0]) for c in rc[0]] == [[9, 19, 17], [9, 19, 17]]
    kc, rc = kat.sort_groups(*O.groupby([K10], [(V10, ["sum"]), (V10, ["sum"])]))
    assert list(rc[0][0][0]) == [9, 19, 17] and list(rc[1][0][0]) == [9, 19, 17]


def test_groupby_errors():
    K = np.array([1, 2, 3], np.int32)
    with pytest.raises(O.OracleError) as e:  # groupby.cu:225-229
        O.groupby([K], [(np.arange(4, dtype=np.int32), ["sum"])])
    assert e.value.code == 1 and "Size mismatch" in str(e.value)
    with pytest.raises(O.OracleError) as e:  # groupby.cu:186-201: SUM of a timestamp is invalid
        O.groupby([K], [(O.HostColumn(np.arange(3, dtype=np.int64), None, "timestamp_s"), ["sum"])])
    assert e.value.code == 1 and "Invalid type/aggregation" in str(e.value)


@pytest.mark.parametrize("c", kat.load("kat_join.json")["table_cases"], ids=lambda c: c["name"])
def test_join_table_kat(c):
    kat.run_join_table_case(O, c)


@pytest.mark.parametrize("c", kat.load("kat_join.json")["hash_join_cases"], ids=lambda c: c["name"])
def test_hash_join_kat(c):
    right = kat.table_cols(c["right"])
    for p in c["probes"]:
        left = kat.table_cols(p["left"])
        li, ri = O.join(left, right, nulls_equal=(c["nulls"] == "equal"), kind=p["kind"])
        assert O.join_size(left, right, nulls_equal=(c["nulls"] == "equal"), kind=p["kind"]) == p["size"]
        assert kat.sorted_pairs(li, ri) == kat.sorted_pairs(p["gold_left"], p["gold_right"])


@pytest.mark.parametrize("c", kat.load("kat_join.json")["match_context_cases"], ids=lambda c: c["name"])
def test_match_context_kat(c):
    """The reference's match-context vectors (join_tests.cpp:2418-2651) against the oracle's joins: the pairs of the LEFT (for left /
    full contexts) or INNER join imply the per-row counts."""
    left = [kat.table_cols(c["left"])[i] for i in c["on"]]
    right = [kat.table_cols(c["right"])[i] for i in c["on"]]
    kind = "inner" if c["kind"] == "inner" else "left"
    li, _ = O.join(left, right, nulls_equal=(c["nulls"] == "equal"), kind=kind)
    assert kat.match_counts_from_pairs(li, len(c["left"]["cols"][0]), c["kind"]) == c["match_counts"]
    if c["size_equals_sum"]:
        assert O.join_size(left, right, nulls_equal=(c["nulls"] == "equal"), kind=kind) == sum(c["match_counts"])


@pytest.mark.parametrize("c", kat.load("kat_join.json")["sorted_index_cases"], ids=lambda c: c["name"])
def test_sorted_index_kat(c):
    left, right = kat.table_cols(c["left"]), kat.table_cols(c["right"])
    for p in c["probes"]:
        li, ri = O.join(left, right, nulls_equal=(c["nulls"] == "equal"), kind=p["kind"])
        assert O.join_size(left, right, nulls_equal=(c["nulls"] == "equal"), kind=p["kind"]) == p["size"] == len(li)
        assert sorted(int(x) for x in li) == p["sorted_left"] and sorted(int(x) for x in ri) == p["sorted_right"]


@pytest.mark.parametrize("c", kat.load("kat_join.json")["partitioned_cases"], ids=lambda c: c["name"])
def test_partitioned_kat(c):
    """join_tests.cpp:3347-3709 on the oracle: joining the row ranges of the left table one by one (indices shifted back to the whole
    table; a full join's unmatched right rows appended once, as finalize_partitioned_full_join does) gives the whole join."""
    left_all, right_all = kat.partitioned_case_tables(c)
    left = [left_all[i] for i in c["on"]]
    right = [right_all[i] for i in c["on"]]
    ne = c["nulls"] == "equal"
    pairs, matched_right = [], set()
    for a, b in c["ranges"]:
        part = [O.HostColumn(col.data[a:b], None if col.valid is None else col.valid[a:b]) for col in left]  # (int32 columns throughout)
        li, ri = O.join(part, right, nulls_equal=ne, kind="inner" if c["kind"] == "inner" else "left") if b > a else ([], [])
        pairs += [(int(l) + a, int(r)) for l, r in zip(li, ri)]
        matched_right |= {int(r) for r in ri if r != -2**31}
    if "expect_pairs" in c:
        assert len(pairs) == c["expect_pairs"]
    if c["kind"] == "full":
        pairs += [(-2**31, r) for r in range(len(right[0].data)) if r not in matched_right]
    covered = sorted(x for a, b in c["ranges"] for x in range(a, b))
    if covered == list(range(len(left[0].data))):
        el, er = O.join(left, right, nulls_equal=ne, kind=c["kind"])
        assert sorted(pairs) == kat.sorted_pairs(el, er)


def test_join_generated_kats():
    z = np.zeros(65567, np.int32)  # join_tests.cpp:2379-2394
    assert O.join_size([z], [z], nulls_equal=False) == 65567 * 65567
    a = np.array([1197], np.int32)
    with pytest.raises(O.OracleError) as e:  # join_tests.cpp:432-450
        O.join([a, a], [a, a, a])
    assert e.value.code == 2
    with pytest.raises(O.OracleError) as e:
        O.join([], [a, a, a])
    assert e.value.code == 2
    with pytest.raises(O.OracleError) as e:  # hash_join.cu:56-58
        O.join([a], [a.astype(np.int64)])
    assert e.value.code == 3


def test_murmur3_vectors():
    for v in kat.load("murmur3_x86_32.json")["vectors"]:
        assert O.murmur3_32(bytes.fromhex(v["hex"]), v["seed"]) == v["hash"]


def test_row_hash_rules():
    # null -> UINT32_MAX as the first column's hash; -0.0 == +0.0; NaNs canonical; bool any-nonzero == 1
    h = O.row_hash([(np.array([1, 2], np.int32), np.array([True, False]))])
    assert h[1] == 0xFFFFFFFF and h[0] == O.murmur3_32(np.int32(1).tobytes(), 0)
    f = np.array([0.0, -0.0, np.nan, -np.nan], np.float64)
    h = O.row_hash([f])
    assert h[0] == h[1] and h[2] == h[3]
    b = O.HostColumn(np.array([1, 2, 255], np.uint8), None, "bool")
    h = O.row_hash([b])
    assert h[0] == h[1] == h[2]
    # two columns: hash_combine(h0, h1)
    a, c = np.array([7], np.int32), np.array([9], np.int64)
    h0 = O.murmur3_32(a.tobytes(), 0)
    h1 = O.murmur3_32(c.tobytes(), 0)
    assert O.row_hash([a, c])[0] == (h0 ^ ((h1 + 0x9E3779B9 + (h0 << 6) + (h0 >> 2)) & 0xFFFFFFFF)) & 0xFFFFFFFF


def test_oracle_vs_pandas_random():
    """Secondary cross-check used by the reference's Python tests (pandas as oracle,
    python/cudf/cudf/testing/testing.py:911-932): random int64 keys / float64 values with nulls."""
    import pandas as pd
    rng = np.random.default_rng(42)
    n = 20000
    k = rng.integers(0, 500, n, dtype=np.int64)
    v = rng.random(n)
    vv = rng.random(n) > 0.1
    kc, rc = kat.sort_groups(*O.groupby([k], [((v, vv), ["sum", "count_valid", "count_all", "min", "max", "mean"])]))
    df = pd.DataFrame({"k": k, "v": np.where(vv, v, np.nan)})
    g = df.groupby("k")["v"]
    assert np.array_equal(kc[0][0], np.sort(df.k.unique()))
    assert np.allclose(rc[0][0][0], g.sum().to_numpy(), rtol=1e-12)
    assert np.array_equal(rc[0][1][0], g.count().to_numpy())
    assert np.array_equal(rc[0][2][0], g.size().to_numpy())
    assert np.allclose(rc[0][3][0], g.min().to_numpy())
    assert np.allclose(rc[0][4][0], g.max().to_numpy())
    assert np.allclose(rc[0][5][0], g.mean().to_numpy(), rtol=1e-12)
