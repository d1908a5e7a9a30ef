"""Where an ARGMIN / ARGMAX call spends its time: the same 1B rows / 500K groups with min + max (one sweep), argmin alone, argmin + argmax."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cudf_amd
from cudf_amd import _lib, aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy
dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
g = torch.Generator(device=dev).manual_seed(21)
k = torch.randint(0, 500_000, (n,), generator=g, device=dev, dtype=torch.int64)
v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
kc, vc = cudf_amd.Column.from_torch(k), cudf_amd.Column.from_torch(v)
for name, aggs in (("min + max + count", [agg.min(), agg.max(), agg.count(NullPolicy.INCLUDE)]), ("argmin + count", [agg.argmin(), agg.count(NullPolicy.INCLUDE)]),
                   ("argmin + argmax + count", [agg.argmin(), agg.argmax(), agg.count(NullPolicy.INCLUDE)])):
    def f():
        grp = gb.GroupBy(cudf_amd.Table([kc]))
        return grp, grp.aggregate([gb.GroupByRequest(vc, aggs)], stream=torch.cuda.current_stream())
    f(); torch.cuda.synchronize()
    _lib.profile_reset(); _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(2):
        grp, r = f()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    _lib.profile_enable(False)
    print(f"{name:26s} {ms:8.2f} ms  path {grp.last_path.name}  {{{', '.join(f'{a}: {b[1] / 2:.2f}' for a, b in sorted(_lib.profile_report().items()))}}}", flush=True)
