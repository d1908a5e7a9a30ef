#!/usr/bin/env python3
"""Call latency of groupby / inner_join at small sizes, where host-side overhead (syncs, launches, allocations) dominates."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cudf_amd
from cudf_amd import aggregation as agg, groupby as gb, join as J
from cudf_amd.types import NullPolicy
dev = torch.device("cuda", 0)
C = cudf_amd.Column.from_torch
g = torch.Generator(device=dev).manual_seed(3)
for n, groups in ((10_000, 100), (1_000_000, 1_000), (1_000_000, 500_000), (10_000_000, 1_000_000)):
    k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
    v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
    kc, vc = C(k), C(v)
    def f():
        grp = gb.GroupBy(cudf_amd.Table([kc]))
        return grp.aggregate([gb.GroupByRequest(vc, [agg.sum(), agg.count(NullPolicy.EXCLUDE)])], stream=torch.cuda.current_stream())
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize()
    print(f"groupby n={n:>9} groups={groups:>8}: {(time.perf_counter() - t0) / 50 * 1e6:8.1f} us per call", flush=True)
for nl, nr in ((10_000, 1_000), (1_000_000, 100_000)):
    lk = torch.randint(0, 2 * nr, (nl,), generator=g, device=dev, dtype=torch.int64)
    rk = torch.randperm(2 * nr, generator=g, device=dev)[:nr].to(torch.int64)
    L, R = cudf_amd.Table([C(lk)]), cudf_amd.Table([C(rk)])
    for _ in range(5): J.inner_join(L, R)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): J.inner_join(L, R)
    torch.cuda.synchronize()
    print(f"inner_join {nl:>9} x {nr:>7}: {(time.perf_counter() - t0) / 50 * 1e6:8.1f} us per call", flush=True)
