"""Dense-key path (direct-address tables, chunked through the Infinity Cache): parity against torch reductions at sizes
that exercise one and several chunks, then a chunk-size sweep of the C2 call.
  python bench_micro/dense_check.py [rows]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import cudf_amd
from cudf_amd import aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy

dev = torch.device("cuda", 0)


def run(k, v, aggs):
    req = gb.GroupByRequest(cudf_amd.Column.from_torch(v), aggs)
    grp = gb.GroupBy(cudf_amd.Table([cudf_amd.Column.from_torch(k)]))
    keys, res = grp.aggregate([req], stream=torch.cuda.current_stream())
    return grp, keys, res


def check(n, groups, lo=0, dtype=torch.float64, label=""):
    g = torch.Generator(device=dev).manual_seed(7)
    k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64) + lo
    if dtype == torch.float64:
        v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
    else:
        v = torch.randint(-1000, 1000, (n,), generator=g, device=dev, dtype=torch.int64)
    grp, keys, res = run(k, v, [agg.sum(), agg.count(NullPolicy.EXCLUDE), agg.min(), agg.max()])
    kk = keys.columns()[0].to_torch() if hasattr(keys.columns()[0], "to_torch") else torch.from_numpy(keys.columns()[0].to_numpy()[0]).to(dev)
    s, c, mn, mx = [torch.from_numpy(x.to_numpy()[0]).to(dev) for x in res[0].columns()]
    order = torch.argsort(kk)
    kk, s, c, mn, mx = kk[order], s[order], c[order], mn[order], mx[order]
    uk, inv, cnt = torch.unique(k, return_inverse=True, return_counts=True)
    ok_keys = bool(torch.equal(kk, uk))
    ok_cnt = bool(torch.equal(c.to(torch.int64), cnt))
    exp_s = torch.zeros(uk.numel(), dtype=v.dtype, device=dev).index_add_(0, inv, v)
    exp_mn = torch.full((uk.numel(),), float("inf") if dtype == torch.float64 else 2**62, dtype=v.dtype, device=dev).scatter_reduce_(0, inv, v, "amin")
    exp_mx = torch.full((uk.numel(),), float("-inf") if dtype == torch.float64 else -2**62, dtype=v.dtype, device=dev).scatter_reduce_(0, inv, v, "amax")
    if dtype == torch.float64:
        err = float((s - exp_s).abs().max() / exp_s.abs().max())
        ok_sum = err < 1e-12
    else:
        err = 0
        ok_sum = bool(torch.equal(s, exp_s))
    ok_mm = bool(torch.equal(mn, exp_mn)) and bool(torch.equal(mx, exp_mx))
    print(f"{label} n={n} groups={groups} lo={lo} path={grp.last_path.name} keys={ok_keys} counts={ok_cnt} sum={ok_sum} ({err:.1e}) minmax={ok_mm}", flush=True)
    assert ok_keys and ok_cnt and ok_sum and ok_mm


def bench(n, groups, steps=5):
    g = torch.Generator(device=dev).manual_seed(42)
    k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
    v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
    req = gb.GroupByRequest(cudf_amd.Column.from_torch(v), [agg.sum(), agg.count(NullPolicy.EXCLUDE)])
    grp = gb.GroupBy(cudf_amd.Table([cudf_amd.Column.from_torch(k)]))
    st = torch.cuda.current_stream()
    for _ in range(2):
        grp.aggregate([req], stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        grp.aggregate([req], stream=st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, grp.last_path.name


if __name__ == "__main__":
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
    check(5_000_000, 1_000_000, label="one chunk")
    check(30_000_000, 1_000_000, lo=-123456789, label="three chunks, negative keys")
    check(30_000_000, 200_000, lo=10**15, dtype=torch.int64, label="int64 values")
    check(30_000_000, 3_000_000, label="3M keys")
    os.environ["CUDF_AMD_GB_DENSE"] = "0"
    check(30_000_000, 1_000_000, label="dense off")
    os.environ["CUDF_AMD_GB_DENSE"] = "1"
    for q in (2, 4, 6, 8, 12, 16, 24, 1000):
        os.environ["CUDF_AMD_GB_CHUNK_ROWS"] = str(q * 256 * 5120)
        ms, path = bench(rows, 1_000_000)
        print(f"C2 rows={rows} chunk={q}x1.31M rows: {ms:.2f} ms/step path={path}", flush=True)
    os.environ["CUDF_AMD_GB_DENSE"] = "0"
    ms, path = bench(rows, 1_000_000)
    print(f"C2 rows={rows} dense off: {ms:.2f} ms/step path={path}", flush=True)
