#!/usr/bin/env python3
"""The headline shape with SORTED / CLUSTERED keys (run on the GPU box from the repo root): sorted_keys.py [rows_millions]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cudf_amd
from cudf_amd import aggregation as agg, groupby as gb, _lib
from cudf_amd.types import NullPolicy
from bench_configs import timed
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) * 1_000_000 if len(sys.argv) > 1 else 1_000_000_000
groups = n // 1000
C = cudf_amd.Column.from_torch
g = torch.Generator(device=dev).manual_seed(7)
v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
shapes = {
    "sorted (runs of 1000)": lambda: torch.arange(n, device=dev, dtype=torch.int64) // 1000,
    "clustered (runs of 64, keys recur)": lambda: (torch.arange(n, device=dev, dtype=torch.int64) // 64 * 2654435761) % groups,
    "uniform": lambda: torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64),
}
for name, mk in shapes.items():
    k = mk()
    kc, vc = C(k), C(v)
    paths = []
    def f():
        grp = gb.GroupBy(cudf_amd.Table([kc]))
        out = grp.aggregate([gb.GroupByRequest(vc, [agg.sum(), agg.count(NullPolicy.EXCLUDE)])], stream=torch.cuda.current_stream())
        paths.append(grp.last_path.name)
        return out
    _, dt, prof = timed(f, 3, 1)
    print(f"{name:36s}: {dt*1e3:8.2f} ms  path {paths[-1]}", {a: round(b, 2) for a, b in prof.items()}, flush=True)
    del k, kc
