#!/usr/bin/env python3
"""Crash / sanity hunt over shapes the benchmarks do not visit, at a size where the big-input paths run (run on the GPU box):
each call must return, its COUNT_ALL must sum to the row count and its group count must match torch.unique.   exotic_shapes.py [rows_millions]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cudf_amd
from cudf_amd import aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy
from bench_configs import bernoulli_mask, timed
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) * 1_000_000 if len(sys.argv) > 1 else 200_000_000
C = cudf_amd.Column.from_torch
g = torch.Generator(device=dev).manual_seed(21)
v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
v32 = torch.randint(-1000, 1000, (n,), generator=g, device=dev, dtype=torch.int32)
vm, vnulls, _ = bernoulli_mask(n, 0.10, 48, dev)
ri = lambda hi, dt=torch.int64: torch.randint(0, hi, (n,), generator=g, device=dev, dtype=dt)
shapes = []
k = ri(3_000_000); shapes.append(("float64 key, 3M groups", [C(k.to(torch.float64) * 0.5)], [(C(v), [agg.sum(), agg.count(NullPolicy.INCLUDE)])], k))
k = ri(2_000_000); shapes.append(("sparse int64 key, 5 aggregations", [C(k * 1_000_003 + 7)], [(C(v), [agg.sum(), agg.min(), agg.max(), agg.mean(), agg.count(NullPolicy.INCLUDE)])], k))
k = ri(500_000); shapes.append(("int64 key, argmin + argmax", [C(k)], [(C(v), [agg.argmin(), agg.argmax(), agg.count(NullPolicy.INCLUDE)])], k))
k = ri(800_000); shapes.append(("two sparse int64 keys", [C(k * 7919), C(k * 31 + 5)], [(C(v), [agg.sum(), agg.count(NullPolicy.INCLUDE)])], k))
k = ri(50_000); shapes.append(("int32 key, int32 value with nulls, variance", [C(k.to(torch.int32))], [(C(v32, vm, vnulls), [agg.variance(), agg.sum(), agg.count(NullPolicy.INCLUDE)])], k))
k = ri(300_000); shapes.append(("three value columns, 300K groups", [C(k)], [(C(v), [agg.sum()]), (C(v32), [agg.max()]), (C(v, vm, vnulls), [agg.mean(), agg.count(NullPolicy.INCLUDE)])], k))
k = ri(20); shapes.append(("20 groups, uint8 key, product", [C(k.to(torch.uint8))], [(C(v), [agg.product(), agg.count(NullPolicy.INCLUDE)])], k))
k = ri(40_000_000); shapes.append(("40M groups (few rows per group)", [C(k)], [(C(v), [agg.sum(), agg.count(NullPolicy.INCLUDE)])], k))
for name, keys, reqs, kref in shapes:
    paths = []
    def f():
        grp = gb.GroupBy(cudf_amd.Table(keys))
        out = grp.aggregate([gb.GroupByRequest(c, a) for c, a in reqs], stream=torch.cuda.current_stream())
        paths.append(grp.last_path.name)
        return out
    (uk, res), dt, prof = timed(f, 2, 1)
    cnt = res[-1].columns()[-1].to_torch()
    ok_count = int(cnt.sum()) == n
    ok_groups = uk.num_rows() == int(torch.unique(kref).numel())
    print(f"{name:45s}: {dt*1e3:8.2f} ms  path {paths[-1]:16s} groups {uk.num_rows():>9} count_sum_ok={ok_count} groups_ok={ok_groups} {prof}", flush=True)
    assert ok_count and ok_groups, name
    del uk, res, cnt
print("exotic shapes OK")
