"""BASELINE.json's single-GPU configurations at FULL size, checked through size-independent properties (the CPU oracle
would take minutes per case here; it pins the same code paths at the sizes of tests/test_groupby_gpu.py / test_join_gpu.py):

  C2  1B int64 keys / float64 values, 1M groups, SUM + COUNT: keys = the distinct keys, counts = torch.bincount exactly,
      sums against a float64 scatter-add within the order-of-summation bound; on the dense-key path and on the hash path
  C3  500M x 50M inner join, 5% nulls, UNEQUAL: pair count = membership count, every pair joins equal valid keys, no pair twice;
      on the direct-address table and (sparse keys) on the hash table
  C4  1B rows, keys (int64, int32 with nulls), float64 value with nulls, MEAN + MIN + MAX: the groups are exactly the distinct
      valid key pairs, every group's MIN / MAX / validity bit-exact and its MEAN = sum / count of a torch per-group reference
      (bincount, scatter_add_, scatter_reduce_) within the order-of-summation bound
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _c2(monkeypatch, dense):
    import torch
    import cudf_amd
    from cudf_amd import aggregation as agg, groupby as gb
    from cudf_amd.types import NullPolicy
    if not dense:
        monkeypatch.setenv("CUDF_AMD_GB_DENSE", "0")
    dev = torch.device("cuda", 0)
    n, groups = 1_000_000_000, 1_000_000
    g = torch.Generator(device=dev).manual_seed(42)
    k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
    g.manual_seed(43)
    v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
    grp = gb.GroupBy(cudf_amd.Table([cudf_amd.Column.from_torch(k)]))
    uk, res = grp.aggregate([gb.GroupByRequest(cudf_amd.Column.from_torch(v), [agg.sum(), agg.count(NullPolicy.EXCLUDE)])],
                            stream=torch.cuda.current_stream())
    assert grp.last_path.name == ("DENSE_DIRECT" if dense else "PARTITIONED_LDS")
    keys = uk.columns()[0].to_torch()
    s, c = [x.to_torch() for x in res[0].columns()]
    exp_c = torch.bincount(k, minlength=groups)
    exp_s = torch.zeros(groups, dtype=torch.float64, device=dev).scatter_add_(0, k, v)
    present = exp_c > 0
    assert keys.numel() == int(present.sum())
    assert bool((torch.sort(keys).values == torch.nonzero(present).flatten()).all())  # every key exactly once
    assert bool((c.to(torch.int64) == exp_c[keys]).all())  # counts bit-exact
    # sums: any order of summing m <= max count values in [0, 1) differs by at most m^2 * eps (kat.sum_atol)
    m = float(exp_c.max())
    assert float((s - exp_s[keys]).abs().max()) <= m * m * np.finfo(np.float64).eps
    assert abs(float(s.sum()) - float(v.sum(dtype=torch.float64))) <= 1e-9 * n
    del k, v, keys, s, c, exp_c, exp_s, uk, res
    torch.cuda.empty_cache()


def test_c2_full_size_dense_keys(gpu, monkeypatch):
    _c2(monkeypatch, dense=True)


def test_c2_full_size_hash_tables(gpu, monkeypatch):
    _c2(monkeypatch, dense=False)


@pytest.mark.parametrize("nval", [2, 3])
def test_c2_full_size_several_value_columns(gpu, nval):
    """C2's rows with two / three float64 value columns ({a: SUM + COUNT, b: SUM, c: SUM}) on the dense path: one value stream per
    column through the ring scatter. Per-group torch reference as for C2."""
    import torch
    import cudf_amd
    from cudf_amd import aggregation as agg, groupby as gb
    from cudf_amd.types import NullPolicy
    dev = torch.device("cuda", 0)
    n, groups = 1_000_000_000, 1_000_000
    g = torch.Generator(device=dev).manual_seed(42)
    k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
    vals = []
    for j in range(nval):
        g.manual_seed(43 + j)
        vals.append(torch.rand(n, generator=g, device=dev, dtype=torch.float64))
    reqs = [gb.GroupByRequest(cudf_amd.Column.from_torch(vals[0]), [agg.sum(), agg.count(NullPolicy.EXCLUDE)])]
    reqs += [gb.GroupByRequest(cudf_amd.Column.from_torch(v), [agg.sum()]) for v in vals[1:]]
    grp = gb.GroupBy(cudf_amd.Table([cudf_amd.Column.from_torch(k)]))
    uk, res = grp.aggregate(reqs, stream=torch.cuda.current_stream())
    assert grp.last_path.name == "DENSE_DIRECT"
    keys = uk.columns()[0].to_torch()
    exp_c = torch.bincount(k, minlength=groups)
    present = exp_c > 0
    assert keys.numel() == int(present.sum())
    assert bool((torch.sort(keys).values == torch.nonzero(present).flatten()).all())
    assert bool((res[0].columns()[1].to_torch().to(torch.int64) == exp_c[keys]).all())
    m = float(exp_c.max())
    for j in range(nval):
        s = res[j].columns()[0].to_torch()
        exp_s = torch.zeros(groups, dtype=torch.float64, device=dev).scatter_add_(0, k, vals[j])
        assert float((s - exp_s[keys]).abs().max()) <= m * m * np.finfo(np.float64).eps, f"column {j}"
        del s, exp_s
    del k, vals, keys, uk, res, exp_c
    torch.cuda.empty_cache()


@pytest.mark.parametrize("sparse", [False, True])
def test_c3_full_size(gpu, sparse):
    import torch
    import bench_configs as BC
    run, check, rows, algo, _ = BC.make_c3(1.0, sparse=sparse)
    checks = check(run())
    assert checks["count_ok"] and checks["keys_equal"] and checks["no_null_rows"] and checks["pairs_distinct"], checks
    torch.cuda.empty_cache()


def test_c4_full_size(gpu):
    import torch
    import bench_configs as BC
    run, check, rows, algo_bytes = BC.make_c4(1.0)
    checks = check(run())
    # per-group reference (counts by bincount, sums by scatter_add_, min / max by scatter_reduce_), not global properties
    assert checks["all_ok"] and checks["global_max_ok"] and checks["global_min_ok"] and checks["mean_within_min_max"], checks
    torch.cuda.empty_cache()


def test_sort_groupby_at_two_hundred_million_rows(gpu):
    """The sort-based path (MEDIAN / NTH_ELEMENT / NUNIQUE + a SUM riding along) at 200M rows on 1M groups, against torch on the same rows:
    the unique keys ascend, NTH_ELEMENT(0) = the value of every key's FIRST row (the key sort is stable), NTH_ELEMENT(-1) its last,
    NUNIQUE = the distinct values per key (values drawn from 1000 levels), MEDIAN = the middle of the group's sorted values (stable
    torch sorts by value, then by key), counts exact."""
    import torch
    import cudf_amd
    from cudf_amd import aggregation as agg, groupby as gb
    from cudf_amd.types import NullPolicy
    dev = torch.device("cuda", 0)
    n, groups = 200_000_000, 1_000_000
    g = torch.Generator(device=dev).manual_seed(7)
    k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
    v = torch.randint(0, 1000, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.float64) * 0.25 - 100.0
    grp = gb.GroupBy(cudf_amd.Table([cudf_amd.Column.from_torch(k)]))
    uk, res = grp.aggregate([gb.GroupByRequest(cudf_amd.Column.from_torch(v), [agg.median(), agg.nth_element(0), agg.nth_element(-1), agg.nunique(),
                                                                           agg.sum(), agg.count(NullPolicy.INCLUDE)])],
                            stream=torch.cuda.current_stream())
    assert grp.last_path.name == "SORT"
    keys = uk.columns()[0].to_torch()
    med, first, last, nuniq, s, cnt = [c.to_torch() for c in res[0].columns()]
    exp_c = torch.bincount(k, minlength=groups)
    assert bool((exp_c > 0).all()) and keys.numel() == groups and bool((keys == torch.arange(groups, device=dev)).all())
    assert bool((cnt.to(torch.int64) == exp_c).all())
    by_value = torch.sort(v, stable=True)
    by_key = torch.sort(k[by_value.indices], stable=True)          # rows in (key, value) order
    vs = by_value.values[by_key.indices]
    del by_value, by_key
    off = torch.cumsum(exp_c, 0) - exp_c
    lo, hi = off + (exp_c - 1) // 2, off + exp_c // 2               # the two middle elements (equal for odd sizes)
    assert bool((med == 0.5 * vs[lo] + 0.5 * vs[hi]).all())          # quantiles_util.hpp linear(): (1 - 0.5) * lo + 0.5 * hi
    distinct = torch.ones(n, dtype=torch.int64, device=dev)
    distinct[1:] = (vs[1:] != vs[:-1]).to(torch.int64)
    distinct[off] = 1
    assert bool((nuniq.to(torch.int64) == torch.zeros(groups, dtype=torch.int64, device=dev).scatter_add_(0, torch.repeat_interleave(
        torch.arange(groups, device=dev), exp_c), distinct)).all())
    del vs, distinct
    order = torch.sort(k, stable=True).indices                      # rows in key order, original order inside a key
    assert bool((first == v[order[off]]).all()) and bool((last == v[order[off + exp_c - 1]]).all())
    exp_s = torch.zeros(groups, dtype=torch.float64, device=dev).scatter_add_(0, k, v)
    assert float((s - exp_s).abs().max()) <= float(exp_c.max()) ** 2 * np.finfo(np.float64).eps * 150.0
