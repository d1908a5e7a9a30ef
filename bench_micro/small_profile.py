import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cudf_amd
from cudf_amd import _lib, aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy
dev = torch.device("cuda", 0)
C = cudf_amd.Column.from_torch
g = torch.Generator(device=dev).manual_seed(3)
for n, groups in ((10_000, 100), (1_000_000, 1_000), (1_000_000, 500_000), (10_000_000, 1_000_000), (100_000_000, 1_000_000)):
    k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
    v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
    kc, vc = C(k), C(v)
    def f():
        grp = gb.GroupBy(cudf_amd.Table([kc]))
        return grp, grp.aggregate([gb.GroupByRequest(vc, [agg.sum(), agg.count(NullPolicy.EXCLUDE)])], stream=torch.cuda.current_stream())
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): r = f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 50 * 1e6
    _lib.profile_reset(); _lib.profile_enable(True)
    for _ in range(10): r = f()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    prof = {k2: round(v2[1] / 10 * 1e3, 1) for k2, v2 in sorted(_lib.profile_report().items())}
    print(f"n={n} groups={groups}: {us:.1f} us per call, path {r[0].last_path.name}, kernels (us): {prof}, sum {sum(prof.values()):.1f}", flush=True)
