"""Adapter that runs the KAT harness (tests/kat.py) against the product on the GPU through the C ABI
(cudf_amd Python mirror -> libcudf_amd.so). Same call shapes as oracle.oracle."""
import numpy as np

import cudf_amd
from cudf_amd import aggregation as agg
from cudf_amd import groupby as gb
from cudf_amd.types import DataType, NullEquality, NullPolicy, TypeId
from oracle.oracle import KIND, HostColumn

_AGG = {"sum": agg.sum, "sum_overflow": agg.sum_with_overflow, "min": agg.min, "max": agg.max, "mean": agg.mean,
        "count_valid": lambda: agg.count(NullPolicy.EXCLUDE), "count_all": lambda: agg.count(NullPolicy.INCLUDE),
        "sum_of_squares": agg.sum_of_squares, "nth_element": lambda: agg.nth_element(0), "median": agg.median,
        "variance": agg.variance, "std": agg.std, "m2": agg.m2, "product": agg.product, "argmin": agg.argmin, "argmax": agg.argmax}


def to_host_column(c):
    if isinstance(c, HostColumn):
        return c
    if isinstance(c, tuple):
        return HostColumn(*c)
    return HostColumn(c)


def to_device(c) -> cudf_amd.Column:
    h = to_host_column(c)
    data = h.data
    dt = DataType(TypeId(h.type_id), getattr(h, "scale", 0))
    if h.type_id == TypeId.BOOL8:
        data = data.astype(np.uint8)
    col = cudf_amd.Column.from_numpy(data, h.valid, dtype=dt, offset=h.offset)
    return col


def from_device(col):
    data, valid = col.to_numpy()
    if col.num_children():  # STRUCT (SUM_OVERFLOW {sum, overflow}): reported under the sum child's type, as the oracle does
        return data, valid, int(col.child(0).type().id())
    return data, valid, int(col.type().id())


last_path = None


def make_aggregation(k):
    """A kind name, or the parameterised form of tests/golden/kat_groupby_sort.json / oracle.sort_groupby: {"kind": ..., ...}."""
    if isinstance(k, str):
        return _AGG[k]()
    policy = lambda dflt: NullPolicy.INCLUDE if k.get("null_policy", dflt) == "include" else NullPolicy.EXCLUDE
    kind = k["kind"]
    if kind == "nth_element":
        return agg.nth_element(int(k.get("n", 0)), policy("include"))
    if kind == "nunique":
        return agg.nunique(policy("exclude"))
    if kind == "quantile":
        return agg.quantile(k["quantiles"], agg.Interpolation[k.get("interpolation", "linear").upper()])
    return _AGG[kind]()


def groupby(keys, requests, include_null_keys=False, keys_are_sorted=False):
    global last_path
    from cudf_amd.types import Sorted
    kt = cudf_amd.Table([to_device(k) for k in keys])
    reqs = [gb.GroupByRequest(to_device(v), [make_aggregation(k) for k in kinds]) for v, kinds in requests]
    g = gb.GroupBy(kt, NullPolicy.INCLUDE if include_null_keys else NullPolicy.EXCLUDE, Sorted.YES if keys_are_sorted else Sorted.NO)
    ukeys, results = g.aggregate(reqs)
    last_path = g.last_path
    return [from_device(c) for c in ukeys.columns()], [[from_device(c) for c in t.columns()] for t in results]


def join(left, right, nulls_equal=True, kind="inner"):
    from cudf_amd import join as J
    lt = cudf_amd.Table([to_device(c) for c in left])
    rt = cudf_amd.Table([to_device(c) for c in right])
    fn = {"inner": J.inner_join, "left": J.left_join, "full": J.full_join}[kind]
    li, ri = fn(lt, rt, NullEquality.EQUAL if nulls_equal else NullEquality.UNEQUAL)
    return li.to_numpy()[0], ri.to_numpy()[0]
