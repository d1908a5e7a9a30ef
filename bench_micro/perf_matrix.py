#!/usr/bin/env python3
"""Looks for performance pathologies: 200M-row SUM+COUNT over a matrix of cardinalities and key distributions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cudf_amd
from cudf_amd import aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy
from bench_configs import timed
dev = torch.device("cuda", 0)
n = int(os.environ.get("PERF_ROWS", 200_000_000))
C = cudf_amd.Column.from_torch
g = torch.Generator(device=dev).manual_seed(5)
v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
def keys(kind, groups):
    if kind == "uniform":
        return torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
    if kind == "sorted":
        return (torch.arange(n, device=dev, dtype=torch.int64) * groups) // n
    if kind == "round_robin":
        return torch.arange(n, device=dev, dtype=torch.int64) % groups
    if kind == "zipf":  # P(rank r) ~ 1/r via inverse transform of a log-uniform
        u = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
        return torch.clamp((float(groups) ** u).to(torch.int64) - 1, 0, groups - 1)
    if kind == "huge_values":
        return torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64) * 9_223_372_036_854 - 4_000_000_000_000_000_000
GROUPS = [int(x) for x in os.environ.get("PERF_GROUPS", "1,10,1000,100000,1000000,10000000,100000000").split(",")]
KINDS = os.environ.get("PERF_KINDS", "uniform,sorted,round_robin,zipf,huge_values").split(",")
for groups in GROUPS:
    for kind in KINDS:
        k = keys(kind, groups)
        kc, vc = C(k), C(v)
        def f():
            grp = gb.GroupBy(cudf_amd.Table([kc]))
            return grp.aggregate([gb.GroupByRequest(vc, [agg.sum(), agg.count(NullPolicy.EXCLUDE)])], stream=torch.cuda.current_stream())
        (uk, res), dt, prof = timed(f, 2, 1)
        ok = int(res[0].columns()[1].to_torch().sum()) == n
        flag = "  <== SLOW" if dt * 1e3 > 12 else ""
        print(f"groups={groups:>9} {kind:12s}: {dt*1e3:7.2f} ms  groups_out={uk.num_rows():>9} count_sum_ok={ok} {({a: round(b, 2) for a, b in prof.items()})}{flag}", flush=True)
        del k, kc, uk, res
