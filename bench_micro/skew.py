#!/usr/bin/env python3
"""The headline shape under skew: a fraction of the rows on 16 hot keys (run on the GPU box from the repo root)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cudf_amd
from cudf_amd import aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy
from bench_configs import timed
dev = torch.device("cuda", 0)
n, groups = 1_000_000_000, 1_000_000
C = cudf_amd.Column.from_torch
g = torch.Generator(device=dev).manual_seed(77)
v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
for frac in (0.0, 0.001, 0.01, 0.2):
    k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
    if frac > 0:
        hot = torch.rand(n, generator=g, device=dev) < frac
        k[hot] = k[hot] % 16
        del hot
    kc, vc = C(k), C(v)
    def f():
        grp = gb.GroupBy(cudf_amd.Table([kc]))
        return grp.aggregate([gb.GroupByRequest(vc, [agg.sum(), agg.count(NullPolicy.EXCLUDE)])], stream=torch.cuda.current_stream())
    _, dt, prof = timed(f, 3, 1)
    print(f"{frac:6.3f} of the rows on 16 hot keys: {dt*1e3:7.2f} ms", {a: round(b, 2) for a, b in prof.items()}, flush=True)
    del k, kc
