#!/usr/bin/env python3
"""Where does C4's level-1 partition pass spend its time? The same 1B-row / 10M-group shape with and without masks and
with plain 8-byte columns only (run on the GPU box from the repo root)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cudf_amd
from cudf_amd import aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy
from bench_configs import bernoulli_mask, timed

dev = torch.device("cuda", 0)
n = 1_000_000_000
g = torch.Generator(device=dev).manual_seed(46)
k0 = torch.randint(0, 10_000, (n,), generator=g, device=dev, dtype=torch.int64)
k1 = torch.randint(0, 1_000, (n,), generator=g, device=dev, dtype=torch.int32)
v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
k1m, k1nulls, _ = bernoulli_mask(n, 0.10, 47, dev)
vm, vnulls, _ = bernoulli_mask(n, 0.10, 48, dev)
k1_64 = k1.to(torch.int64)


def run(name, keys, vals, aggs):
    def f():
        grp = gb.GroupBy(keys, NullPolicy.EXCLUDE)
        return grp.aggregate([gb.GroupByRequest(vals, aggs)], stream=torch.cuda.current_stream())
    _, dt, prof = timed(f, 3, 1)
    print(name, round(dt * 1e3, 2), "ms", {k: round(x, 2) for k, x in prof.items()}, flush=True)


C = cudf_amd.Column.from_torch
aggs = [agg.mean(), agg.min(), agg.max()]
run("C4 (int64,int32 nullable) value nullable ", cudf_amd.Table([C(k0), C(k1, k1m, k1nulls)]), C(v, vm, vnulls), aggs)
run("same keys, no masks at all              ", cudf_amd.Table([C(k0), C(k1)]), C(v), aggs)
run("keys (int64,int64), no masks (plain)    ", cudf_amd.Table([C(k0), C(k1_64)]), C(v), aggs)
run("keys (int64,int64), value nullable      ", cudf_amd.Table([C(k0), C(k1_64)]), C(v, vm, vnulls), aggs)
