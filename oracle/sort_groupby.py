"""TEST INFRASTRUCTURE ONLY — numpy / pure-Python restatement of the reference's sort-based groupby for the kinds its hash
groupby cannot serve (MEDIAN, QUANTILE, NUNIQUE, NTH_ELEMENT). Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (cudf_amd/) never does.

Follows, by file:line of the reference:
  cpp/src/groupby/groupby.cu:64-69           one non-hash kind takes the whole call down the sort path
  cpp/src/groupby/sort/sort_helper.cu:75-117 stable sorted order of the key rows, ascending, nulls after; rows with a null in any
                                             key column behind everything and dropped when null keys are excluded
  sort_helper.cu:121-160                     group offsets (adjacent rows unequal) and labels
  sort_helper.cu:205-225                     values sorted inside each group: stable order of (label, value), nulls after
  cpp/src/groupby/sort/group_nth_element.cu:30-128, group_nunique.cu:35-121, group_quantiles.cu:35-171,
  cpp/src/quantiles/quantiles_util.hpp:20-176 (interpolation arithmetic)
Pinned by tests/golden/kat_groupby_sort.json (transcribed from the reference's gtests of these kinds).

The kinds the hash groupby serves are delegated to oracle.c's groupby keyed on the group label and re-ordered.
"""
import math

import numpy as np

from . import oracle as O

INTERPOLATION = {"linear": 0, "lower": 1, "higher": 2, "midpoint": 3, "nearest": 4, "nearest_half_up": 5}
_FLOAT_IDS = (O.TYPE_ID["float32"], O.TYPE_ID["float64"])
_ARITHMETIC_IDS = tuple(range(O.TYPE_ID["int8"], O.TYPE_ID["bool"] + 1))


def _as_host(c):
    return c if isinstance(c, O.HostColumn) else O.HostColumn(*c) if isinstance(c, tuple) else O.HostColumn(c)


def _logical(col):
    """The column's elements (offset applied) as a numpy array of its storage type, and validity."""
    data = col.data[col.offset:col.offset + col.size]
    if col.type_id == O.TYPE_ID["bool"]:
        data = data != 0
    valid = np.ones(col.size, bool) if col.valid is None else col.valid[col.offset:col.offset + col.size]
    return data, valid


def sortable_words(col):
    """uint64 words whose unsigned order is the reference's ascending order of the values: NaN above every number and all NaNs
    one value, -0 == +0 (row_operator/lexicographic.cuh relational comparator); nulls get word 0 (ordered by the validity)."""
    data, valid = _logical(col)
    if data.dtype.kind == "f":
        d = data.astype(np.float64)  # float32 -> float64 is order- and equality-preserving
        d = np.where(d == 0, 0.0, d)
        bits = d.view(np.uint64).copy()
        bits[np.isnan(d)] = np.uint64(0x7ff8000000000000)
        neg = (bits >> np.uint64(63)) != 0
        words = np.where(neg, ~bits, bits | np.uint64(1 << 63))
    elif data.dtype.kind == "b":
        words = data.astype(np.uint64)
    elif data.dtype.kind == "u":
        words = data.astype(np.uint64)
    else:
        words = data.astype(np.int64).view(np.uint64) ^ np.uint64(1 << 63)
    return np.where(valid, words, np.uint64(0)), valid


class SortHelper:
    """cudf::groupby::detail::sort::sort_groupby_helper restated on the host."""

    def __init__(self, keys, include_null_keys=False, keys_are_sorted=False):
        self.keys = [_as_host(k) for k in keys]
        n = self.keys[0].size
        per_col = [sortable_words(k) for k in self.keys]
        any_null = np.zeros(n, bool)
        for _, valid in per_col:
            any_null |= ~valid
        if keys_are_sorted and not (not include_null_keys and any_null.any()):
            order = np.arange(n)
        else:
            sort_keys = []  # np.lexsort: last key is the primary one; stable
            for words, valid in reversed(per_col):
                sort_keys.append(words)
                sort_keys.append(~valid)
            if not include_null_keys:
                sort_keys.append(any_null)
            order = np.lexsort(sort_keys) if n else np.zeros(0, np.int64)
        self.num_keys = n if include_null_keys else int(n - any_null.sum())
        self.order = order[:self.num_keys]
        if self.num_keys == 0:
            self.offsets = np.zeros(1, np.int64)
            self.labels = np.zeros(0, np.int64)
            self.num_groups = 0
            return
        boundary = np.zeros(self.num_keys, bool)
        boundary[0] = True
        for words, valid in per_col:
            w, v = words[self.order], valid[self.order]
            boundary[1:] |= (w[1:] != w[:-1]) | (v[1:] != v[:-1])
        self.labels = np.cumsum(boundary) - 1
        self.num_groups = int(self.labels[-1]) + 1
        self.offsets = np.append(np.flatnonzero(boundary), self.num_keys)

    def unique_keys(self):
        first = self.order[self.offsets[:-1]]
        out = []
        for k in self.keys:
            data = k.data[k.offset:k.offset + k.size][first]
            valid = None if k.valid is None else k.valid[k.offset:k.offset + k.size][first]
            if k.type_id == O.TYPE_ID["bool"]:
                data = data != 0
            out.append((data.astype(O.NP_OF_TYPE_ID[k.type_id]), valid, k.type_id))
        return out

    def sorted_values_order(self, values):
        """Rows in (label, null, value) order."""
        words, valid = sortable_words(values)
        w, v = words[self.order], valid[self.order]
        inner = np.lexsort([w, ~v, self.labels])
        return self.order[inner]


def _quantile_index(count, quantile):
    quantile = min(max(quantile, 0.0), 1.0)
    val = quantile * (count - 1)
    lower = math.floor(val)
    higher = math.ceil(val)
    nearest = int(np.rint(val))  # half to even, as nearbyint in the default rounding mode
    half_up = int(math.floor(abs(val) + 0.5) * (1 if val >= 0 else -1))  # std::round: half away from zero
    return lower, higher, nearest, half_up, val - lower


def _select_quantile(sorted_vals, is_int64, quantile, interp):
    """quantiles_util.hpp:147-176 on the valid, sorted values of one group; doubles throughout."""
    size = len(sorted_vals)
    lower, higher, nearest, half_up, fraction = _quantile_index(size, quantile)
    at = lambda i: float(sorted_vals[i])
    if interp == 1:
        return at(lower)
    if interp == 2:
        return at(higher)
    if interp == 4:
        return at(nearest)
    if interp == 5:
        return at(half_up)
    if interp == 3:
        if is_int64:  # quantiles_util.hpp:46-53: halves and remainders apart (C++ division truncates toward zero)
            l, h = int(sorted_vals[lower]), int(sorted_vals[higher])
            tdiv = lambda a: a // 2 if a >= 0 else -((-a) // 2)
            trem = lambda a: a - 2 * tdiv(a)
            return float(tdiv(l) + tdiv(h)) + float(trem(l) + trem(h)) * 0.5
        return at(lower) / 2 + at(higher) / 2
    one_minus = 1.0 - fraction
    return one_minus * at(lower) + fraction * at(higher)


def _spec(a):
    if isinstance(a, str):
        return {"kind": a}
    return dict(a)


def _is_sort_kind(kind):
    return kind in ("median", "quantile", "nunique", "nth_element")


def groupby(keys, requests, include_null_keys=False, keys_are_sorted=False):
    """keys: list of columns; requests: list of (values column, [aggregation]) where an aggregation is a kind name of
    oracle.KIND or {"kind": "nth_element", "n": int, "null_policy": "include"|"exclude"} / {"kind": "nunique", "null_policy": ...} /
    {"kind": "median"} / {"kind": "quantile", "quantiles": [...], "interpolation": name}.
    Returns (key_cols, result_cols) in the shapes of oracle.groupby; the keys ascend, nulls last."""
    keys = [_as_host(k) for k in keys]
    requests = [(_as_host(v), [_spec(a) for a in aggs]) for v, aggs in requests]
    n = keys[0].size
    if n == 0:
        res = []
        for v, aggs in requests:
            res.append([(np.zeros(0, O.NP_OF_TYPE_ID[_result_type(v.type_id, a["kind"])]), None, _result_type(v.type_id, a["kind"])) for a in aggs])
        return [(np.zeros(0, O.NP_OF_TYPE_ID[k.type_id]), None, k.type_id) for k in keys], res
    h = SortHelper(keys, include_null_keys, keys_are_sorted)
    G = h.num_groups
    results = [[None] * len(aggs) for _, aggs in requests]

    # the hash kinds: oracle.c keyed on the label (rows of excluded keys in an extra group that is dropped)
    engine = [(r, j) for r, (_, aggs) in enumerate(requests) for j, a in enumerate(aggs) if not _is_sort_kind(a["kind"])]
    if engine and G > 0:
        row_label = np.full(n, G, np.int32)
        row_label[h.order] = h.labels
        ereq = []
        for r, (v, aggs) in enumerate(requests):
            kinds = [a["kind"] for a in aggs if not _is_sort_kind(a["kind"])]
            if kinds:
                ereq.append((v, kinds))
        kc, rc = O.groupby([O.HostColumn(row_label)], ereq, include_null_keys=True)
        where = np.full(G + 1, -1, np.int64)
        where[kc[0][0]] = np.arange(len(kc[0][0]))
        take = where[:G]
        flat = [c for req in rc for c in req]
        for (r, j), (data, valid, tid) in zip(engine, flat):
            if isinstance(data, tuple):
                data = tuple(d[take] for d in data)
            else:
                data = data[take]
            results[r][j] = (data, None if valid is None else valid[take], tid)
    elif engine:
        for r, j in engine:
            v, aggs = requests[r]
            tid = _result_type(v.type_id, aggs[j]["kind"])
            results[r][j] = (np.zeros(0, O.NP_OF_TYPE_ID[tid]), None, tid)

    for r, (v, aggs) in enumerate(requests):
        data, valid = _logical(v)
        for j, a in enumerate(aggs):
            kind = a["kind"]
            if not _is_sort_kind(kind):
                continue
            tid = _result_type(v.type_id, kind)
            npt = O.NP_OF_TYPE_ID[tid]
            if kind == "nth_element":
                nth, include = int(a.get("n", 0)), a.get("null_policy", "include") == "include"
                out, ok = np.zeros(G, npt), np.zeros(G, bool)
                for g in range(G):
                    rows = h.order[h.offsets[g]:h.offsets[g + 1]]  # the group's rows in their original order (stable sort)
                    if not include and not valid.all():
                        rows = rows[valid[rows]]  # group_nth_element.cu:73-116: the n-th VALID row
                    k = len(rows) + nth if nth < 0 else nth
                    if 0 <= k < len(rows):
                        out[g], ok[g] = data[rows[k]], valid[rows[k]]
                results[r][j] = (out, None if ok.all() else ok, tid)
            elif kind == "nunique":
                include = a.get("null_policy", "exclude") == "include"
                words, _ = sortable_words(v)
                out = np.zeros(G, np.int32)
                for g in range(G):
                    rows = h.order[h.offsets[g]:h.offsets[g + 1]]
                    distinct = set(words[rows][valid[rows]].tolist())
                    out[g] = len(distinct) + (1 if include and not valid[rows].all() else 0)
                results[r][j] = (out, None, tid)
            else:
                if v.type_id not in _ARITHMETIC_IDS:
                    raise O.OracleError(1, "Only arithmetic types are supported in quantiles")
                qs = [0.5] if kind == "median" else [float(q) for q in a["quantiles"]]
                interp = 0 if kind == "median" else INTERPOLATION[a.get("interpolation", "linear")]
                vorder = h.sorted_values_order(v)
                out, ok = np.zeros(G * len(qs), np.float64), np.zeros(G * len(qs), bool)
                for g in range(G):
                    rows = vorder[h.offsets[g]:h.offsets[g + 1]]
                    rows = rows[valid[rows]]  # the valid rows lead (nulls after)
                    if len(rows) == 0:
                        continue
                    vals = data[rows]
                    vals = vals.astype(np.float64) if vals.dtype.kind in "fb" else vals.astype(object)
                    for t, q in enumerate(qs):
                        out[g * len(qs) + t] = _select_quantile(vals, v.type_id == O.TYPE_ID["int64"], q, interp)
                        ok[g * len(qs) + t] = True
                # the reference always allocates the mask (group_quantiles.cu:84-88)
                results[r][j] = (out, ok, tid)
    return h.unique_keys(), results


def _result_type(value_tid, kind):
    if kind == "nth_element":
        return value_tid
    if kind == "nunique":
        return O.TYPE_ID["int32"]
    if kind in ("median", "quantile"):
        return O.TYPE_ID["float64"]
    # hash kinds on empty input: detail/aggregation/aggregation.hpp:878-978
    integral = value_tid in tuple(range(O.TYPE_ID["int8"], O.TYPE_ID["uint64"] + 1)) + (O.TYPE_ID["bool"],)
    if kind in ("count_valid", "count_all", "argmin", "argmax"):
        return O.TYPE_ID["int32"]
    if kind in ("mean", "variance", "std", "m2"):
        return O.TYPE_ID["float64"] if value_tid in _ARITHMETIC_IDS else value_tid
    if kind in ("sum", "sum_of_squares", "product"):
        return O.TYPE_ID["int64"] if integral else value_tid
    return value_tid
