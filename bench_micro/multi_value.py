#!/usr/bin/env python3
"""Several value columns (one request each) over low-cardinality keys: what the single-pass hash kernel does there (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cudf_amd
from cudf_amd import aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy
from bench_configs import timed
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) * 1_000_000 if len(sys.argv) > 1 else 1_000_000_000
C = cudf_amd.Column.from_torch
g = torch.Generator(device=dev).manual_seed(9)
vals = [torch.rand(n, generator=g, device=dev, dtype=torch.float64) for _ in range(3)]
for groups in (1000, 1_000_000):
    k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
    kc = C(k)
    for nv in (1, 2, 3):
        vcs = [C(v) for v in vals[:nv]]
        paths = []
        def f():
            grp = gb.GroupBy(cudf_amd.Table([kc]))
            out = grp.aggregate([gb.GroupByRequest(vc, [agg.sum(), agg.mean()]) for vc in vcs], stream=torch.cuda.current_stream())
            paths.append(grp.last_path.name)
            return out
        _, dt, prof = timed(f, 3, 1)
        gbytes = n * 8 * (1 + nv) / 1e9
        print(f"groups={groups:>8} value columns={nv}: {dt*1e3:7.2f} ms  ({gbytes:.0f} GB of input: {gbytes/dt/1e3:.2f} TB/s)  path {paths[-1]}", {a: round(b, 2) for a, b in prof.items()}, flush=True)
    del k, kc
