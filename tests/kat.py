"""Shared KAT / parity harness. Restates the reference's test harness rules:
 - groupby: sort expected and actual by key, keys EQUAL, values EQUIVALENT
   (cpp/tests/groupby/groupby_test_util.cpp:36-48,82-89);
 - join: gather both tables by the returned index vectors, sort rows, compare
   (cpp/tests/join/join_tests.cpp:70-101,318-345,1250-1252);
 - floats: |x-y| <= 4*eps*|x+y|, inf/NaN exact (cpp/tests/utilities/column_utilities.cu:436-440).
A "backend" is anything with groupby(keys, requests, include_null_keys) / join(left, right, nulls_equal, kind)
taking oracle.HostColumn inputs and returning the oracle's tuple shapes.
"""
import json
import os

import numpy as np

from oracle.oracle import KIND, NP_OF_TYPE_ID, TYPE_ID, HostColumn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FP_ULPS = 4


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def _num(x):
    return {"inf": np.inf, "-inf": -np.inf, "nan": np.nan}.get(x, x) if isinstance(x, str) else x


def _sym(v, npt):
    """"max-5" / "min+5": offsets from the limits of the column's storage type (the reference's typed overflow tests
    build their inputs from numeric_limits, sum_overflow_tests.cpp:300-352)."""
    if isinstance(v, str) and v[:3] in ("max", "min"):
        base = np.iinfo(npt).max if v[:3] == "max" else np.iinfo(npt).min
        return int(base) + (int(v[3:]) if len(v) > 3 else 0)
    return _num(v)


def host_col(values, type_name, valid=None):
    npt = NP_OF_TYPE_ID[TYPE_ID[type_name]]
    arr = np.array([_sym(v, npt) for v in values], dtype=np.float64 if "float" in type_name else np.int64).astype(npt)
    return HostColumn(arr, None if valid is None else np.array(valid, dtype=bool), type_name)


def expected_type_id(value_type, agg):
    if isinstance(agg, dict):  # the sort-groupby kinds (detail/aggregation/aggregation.hpp:1015-1049)
        return {"nth_element": TYPE_ID[value_type], "nunique": TYPE_ID["int32"]}.get(agg["kind"], TYPE_ID["float64"])
    integral = value_type in ("int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "bool")
    if agg in ("count_valid", "count_all", "argmin", "argmax"):
        return TYPE_ID["int32"]
    if agg == "mean" and (value_type.startswith("duration") or value_type.startswith("decimal")):
        return TYPE_ID[value_type]  # integer division in the source type (aggregation.hpp:935-941)
    if agg in ("mean", "variance", "std", "m2"):
        return TYPE_ID["float64"]
    if agg in ("sum", "sum_of_squares", "product"):
        return TYPE_ID["int64"] if integral else TYPE_ID[value_type]
    return TYPE_ID[value_type]


def equivalent(a, b, is_float, atol=0.0):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if not is_float:
        return bool(np.array_equal(a, b))
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    with np.errstate(invalid="ignore"):
        return _equivalent_f64(a, b, atol)


def _equivalent_f64(a, b, atol=0.0):
    nan_ok = np.isnan(a) == np.isnan(b)
    inf = np.isinf(a) | np.isinf(b)
    inf_ok = np.where(inf, a == b, True)
    eps = np.finfo(np.float64).eps
    fin = ~(np.isnan(a) | np.isnan(b) | inf)
    close = np.where(fin, np.abs(a - b) <= FP_ULPS * eps * np.abs(a + b) + atol, True)
    return bool(np.all(nan_ok & inf_ok & close))


def equivalent_f32(a, b, atol=0.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    eps = np.finfo(np.float32).eps
    fin = np.isfinite(a) & np.isfinite(b)
    with np.errstate(invalid="ignore"):
        ok = np.where(fin, np.abs(a - b) <= FP_ULPS * eps * np.abs(a + b) + atol, (a == b) | (np.isnan(a) & np.isnan(b)))
    return bool(np.all(ok))


def sort_groups(key_cols, result_cols):
    """Sort rows by the key columns (nulls last), returning permuted copies."""
    n = len(key_cols[0][0]) if key_cols else 0
    if n == 0:
        return key_cols, result_cols
    sort_keys = []
    for data, valid, _ in reversed(key_cols):
        d = np.where(np.isnan(data.astype(np.float64)), np.inf, data) if data.dtype.kind == "f" else data
        if valid is not None:
            d = np.where(valid, d, 0)
            sort_keys.append(d)
            sort_keys.append(~valid)
        else:
            sort_keys.append(d)
    order = np.lexsort(sort_keys)
    take = lambda d: tuple(x[order] for x in d) if isinstance(d, tuple) else d[order]  # (struct column: tuple of children)
    perm = lambda col: (take(col[0]), None if col[1] is None else col[1][order], col[2])
    return [perm(c) for c in key_cols], [[perm(c) for c in req] for req in result_cols]


def sum_atol(max_terms, max_abs):
    """Worst-case error bound of summing `max_terms` float64 values of magnitude <= max_abs in ANY order:
    (m-1) * eps * sum|v_i| <= m^2 * eps * max|v| (Higham, Accuracy and Stability, eq. 4.4). The reference sums with
    unordered atomics (device_atomics.cuh:36-139), so two correct results may differ by this much; the
    reference's 4-ulp rule (column_utilities.cu:436-440) is kept for everything that is not an order-dependent sum."""
    return float(max_terms) ** 2 * np.finfo(np.float64).eps * float(max_abs)


def compare_columns(actual, expected, what="", atol=0.0):
    a_data, a_valid, a_tid = actual
    e_data, e_valid, e_tid = expected
    assert a_tid == e_tid, f"{what}: type id {a_tid} != {e_tid}"
    if isinstance(e_data, tuple):
        # SUM_OVERFLOW struct {sum, overflow}: validity and flags exact; sums exact where the group did not overflow (an
        # overflowed group's sum depends on the arrival order of the reference's atomics; its own tests compare only the flags,
        # sum_overflow_tests.cpp:271-288)
        assert isinstance(a_data, tuple) and len(a_data) == 2, f"{what}: expected a struct of two children"
        n = len(e_data[0])
        av = np.ones(n, bool) if a_valid is None else a_valid
        ev = np.ones(n, bool) if e_valid is None else e_valid
        assert len(a_data[0]) == n and np.array_equal(av, ev), f"{what}: struct validity differs\n{av}\n{ev}"
        af, ef = np.asarray(a_data[1], bool), np.asarray(e_data[1], bool)
        assert np.array_equal(af[ev], ef[ev]), f"{what}: overflow flags differ\n{af}\n{ef}"
        ok = ev & ~ef
        assert np.array_equal(np.asarray(a_data[0])[ok], np.asarray(e_data[0])[ok]), f"{what}: sums differ\n{a_data[0]}\n{e_data[0]}"
        return
    assert len(a_data) == len(e_data), f"{what}: size {len(a_data)} != {len(e_data)}"
    av = np.ones(len(a_data), bool) if a_valid is None else a_valid
    ev = np.ones(len(e_data), bool) if e_valid is None else e_valid
    if not np.array_equal(av, ev):
        bad = np.flatnonzero(av != ev)
        raise AssertionError(f"{what}: validity differs at rows {bad[:8]} (of {len(bad)}): actual valid {av[bad[:8]]} data "
                             f"{a_data[bad[:8]]}, expected valid {ev[bad[:8]]} data {e_data[bad[:8]]}")
    is_float = np.dtype(NP_OF_TYPE_ID[a_tid]).kind == "f"
    a = a_data[av]
    e = e_data[ev]
    if isinstance(atol, np.ndarray):  # one bound per row (per group)
        atol = atol[ev]
    if a_tid == TYPE_ID["float32"]:
        assert equivalent_f32(a, e, atol), f"{what}: values differ\n{a}\n{e}"
    else:
        assert equivalent(a, e, is_float, atol), f"{what}: values differ\n{a}\n{e}"


def groupby_cases():
    doc = load("kat_groupby.json")
    for c in doc["cases"]:
        for kt in c["key_types"]:
            for vt in c["value_types"]:
                yield f'{c["name"]}[{kt}-{vt}]', c, kt, vt


def run_groupby_case(backend, c, kt, vt):
    keys = [host_col(c["keys"], kt, c["keys_valid"])]
    vals = host_col(c["values"], vt, c["values_valid"])
    kc, rc = backend.groupby(keys, [(vals, [c["agg"]])], include_null_keys=(c["null_policy"] == "include"))
    kc, rc = sort_groups(kc, rc)
    if c.get("only_num_groups") is not None:
        assert len(kc[0][0]) == c["only_num_groups"]
        return
    ek = host_col(c["expect_keys"], kt, c["expect_keys_valid"])
    exp_tid = expected_type_id(vt, c["agg"])
    exp_np = NP_OF_TYPE_ID[exp_tid]
    ev_data = np.array([_num(v) for v in c["expect"]], dtype=np.float64 if np.dtype(exp_np).kind == "f" else np.int64).astype(exp_np)
    if c["agg"] == "sum_overflow":
        ev_data = (ev_data, np.array(c["expect_overflow"], dtype=bool))
    ev_valid = None if c["expect_valid"] is None else np.array(c["expect_valid"], dtype=bool)
    ekc, erc = sort_groups([(ek.data, ek.valid, ek.type_id)], [[(ev_data, ev_valid, exp_tid)]])
    compare_columns(kc[0], ekc[0], "keys")
    compare_columns(rc[0][0], erc[0][0], "values")


def sort_groupby_cases():
    doc = load("kat_groupby_sort.json")
    for c in doc["cases"]:
        for kt in c["key_types"]:
            for vt in c["value_types"]:
                yield f'{c["name"]}[{kt}-{vt}]', c, kt, vt


def run_sort_groupby_case(backend, c, kt, vt):
    """The kinds only the sort-based groupby serves: its unique keys ascend, so keys and results are compared in place
    (a QUANTILE result holds groups x quantiles values and could not be permuted with the keys anyway)."""
    keys = [host_col(c["keys"], kt, c["keys_valid"])]
    vals = host_col(c["values"], vt, c["values_valid"])
    kw = {"keys_are_sorted": True} if c.get("keys_are_sorted") else {}
    kc, rc = backend.groupby(keys, [(vals, [c["agg"]])], include_null_keys=(c["null_policy"] == "include"), **kw)
    ek = host_col(c["expect_keys"], kt, c["expect_keys_valid"])
    exp_tid = expected_type_id(vt, c["agg"])
    exp_np = NP_OF_TYPE_ID[exp_tid]
    ev_data = np.array([_num(v) for v in c["expect"]], dtype=np.float64 if np.dtype(exp_np).kind == "f" else np.int64).astype(exp_np)
    ev_valid = None if c["expect_valid"] is None else np.array(c["expect_valid"], dtype=bool)
    ek_data = ek.data.astype(np.bool_) if kt == "bool" else ek.data
    ekc, erc = [(ek_data, ek.valid, ek.type_id)], [[(ev_data, ev_valid, exp_tid)]]
    if isinstance(c["agg"], str):  # a hash kind on pre-sorted keys: the group order is unspecified (the hash path may answer), one value per group
        kc, rc = sort_groups(kc, rc)
        ekc, erc = sort_groups(ekc, erc)
    compare_columns(kc[0], ekc[0], "keys")
    compare_columns(rc[0][0], erc[0][0], "values")


# ---------------------------------------------------------------- joins
def table_cols(t):
    return [host_col(col, ty, v) for col, ty, v in zip(t["cols"], t["types"], t["valid"])]


def gather_rows(cols, idx):
    """Rows of `cols` at idx as tuples with None for NULL (idx == JoinNoMatch -> all NULL)."""
    out = []
    for i in idx:
        if i == -2**31:
            out.append(tuple(None for _ in cols))
        else:
            out.append(tuple(None if (c.valid is not None and not c.valid[i]) else c.data[i].item() for c in cols))
    return out


def _row_key(r):
    return tuple((x is None, 0 if x is None else x) for x in r)


def run_join_table_case(backend, c):
    left, right = table_cols(c["left"]), table_cols(c["right"])
    lk = [left[i] for i in c["left_on"]]
    rk = [right[i] for i in c["right_on"]]
    li, ri = backend.join(lk, rk, nulls_equal=(c["nulls"] == "equal"), kind=c["kind"])
    assert len(li) == len(ri)
    if "expect_num_rows" in c:
        assert len(li) == c["expect_num_rows"]
        return
    got = sorted((l + r for l, r in zip(gather_rows(left, li), gather_rows(right, ri))), key=_row_key)
    g = c["gold"]
    n = len(g["cols"][0])
    gold = []
    for i in range(n):
        gold.append(tuple(None if (g["valid"][j] is not None and not g["valid"][j][i]) else g["cols"][j][i]
                          for j in range(len(g["cols"]))))
    gold = sorted(gold, key=_row_key)
    assert got == gold, f"join rows differ:\n{got}\n{gold}"


def partitioned_case_tables(c):
    """(left host columns, right host columns) of a `partitioned_cases` entry; a generated case builds its columns from the recipe the
    reference test states (join_tests.cpp:3636-3709: left[i] = i % 200 sliced to [1234, 3734), right[i] = i)."""
    if "generator" in c:
        g = c["generator"]
        assert g["left_value"] == "i % 200" and g["right_value"] == "i"
        a, b = g["left_slice"]
        left = [host_col((np.arange(g["left_full_rows"]) % 200)[a:b].tolist(), "int32", None)]
        right = [host_col(np.arange(g["right_rows"]).tolist(), "int32", None)]
        return left, right
    return table_cols(c["left"]), table_cols(c["right"])


def match_counts_from_pairs(li, nrows, kind):
    """Per-left-row match counts a join's pairs imply (what *_join_match_context holds): inner = matches, left / full = matches or 1."""
    li = np.asarray([x for x in li if x != -2**31], dtype=np.int64)
    counts = np.bincount(li, minlength=nrows).astype(np.int64) if nrows else np.zeros(0, np.int64)
    return counts.tolist()


def sorted_pairs(li, ri):
    return sorted(zip([int(x) for x in li], [int(x) for x in ri]))
