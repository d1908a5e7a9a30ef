"""Times the sort-based groupby (MEDIAN / NTH_ELEMENT / NUNIQUE) on the headline shape's keys and prints the library's per-kernel
profile. usage: python bench_micro/sort_groupby_bench.py [rows] [groups] [key dtype int32|int64] [kinds median,nth,nunique,sum]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import torch

import cudf_amd
from cudf_amd import _lib, aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
kdt = {"int32": torch.int32, "int64": torch.int64}[sys.argv[3] if len(sys.argv) > 3 else "int64"]
kinds = (sys.argv[4] if len(sys.argv) > 4 else "median").split(",")
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=kdt)
v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
kc, vc = cudf_amd.Column.from_torch(k), cudf_amd.Column.from_torch(v)
make = {"median": agg.median, "nth": lambda: agg.nth_element(0), "nunique": agg.nunique, "sum": agg.sum,
        "quantile": lambda: agg.quantile([0.25, 0.75])}
lib = _lib.load()


def run():
    grp = gb.GroupBy(cudf_amd.Table([kc]))
    return grp, grp.aggregate([gb.GroupByRequest(vc, [make[x]() for x in kinds])], stream=torch.cuda.current_stream())


run()
torch.cuda.synchronize()
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    grp, r = run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
print(f"rows {n} groups {groups} keys {kdt} kinds {kinds}: {ms:.2f} ms per call, path {grp.last_path.name}, {n / ms / 1e6:.2f} Grows/s")
lib.cudf_amd_profile_enable(1)
lib.cudf_amd_profile_reset()
run()
torch.cuda.synchronize()
buf = C.create_string_buffer(1 << 16)
lib.cudf_amd_profile_report(buf, len(buf))
print(buf.value.decode())
