#!/usr/bin/env python3
"""1B-row groupby SUM+COUNT_VALID over shapes that leave the plain-column fast path (run on the GPU box from the repo root)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cudf_amd
from cudf_amd import aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy
from bench_configs import bernoulli_mask, timed

dev = torch.device("cuda", 0)
n = 1_000_000_000
C = cudf_amd.Column.from_torch
g = torch.Generator(device=dev).manual_seed(1)
v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
vm, vnulls, _ = bernoulli_mask(n, 0.10, 48, dev)
for groups in (1000, 1_000_000):
    k64 = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
    k32 = k64.to(torch.int32)
    for name, keys, vals in (("int64 key, plain value   ", [C(k64)], C(v)), ("int64 key, nullable value", [C(k64)], C(v, vm, vnulls)),
                             ("int32 key, plain value   ", [C(k32)], C(v)), ("int32 key, nullable value", [C(k32)], C(v, vm, vnulls))):
        def f():
            grp = gb.GroupBy(cudf_amd.Table(keys), NullPolicy.EXCLUDE)
            return grp.aggregate([gb.GroupByRequest(vals, [agg.sum(), agg.count(NullPolicy.EXCLUDE)])], stream=torch.cuda.current_stream())
        _, dt, prof = timed(f, 3, 1)
        print(f"groups={groups:>8} {name}: {dt * 1e3:7.2f} ms", {a: round(b, 2) for a, b in prof.items()}, flush=True)
    del k64, k32
