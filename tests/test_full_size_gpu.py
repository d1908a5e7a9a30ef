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
