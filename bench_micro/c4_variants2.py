"""Where does the composite dense level-1 scatter lose its time? C4 with narrower / wider key types and with / without nulls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cudf_amd
from cudf_amd import _lib, aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy
import bench_configs as B
dev = torch.device("cuda", 0)
n = 1_000_000_000
g = torch.Generator(device=dev).manual_seed(46)
k0 = torch.randint(0, 10_000, (n,), generator=g, device=dev, dtype=torch.int64)
k1_32 = torch.randint(0, 1_000, (n,), generator=g, device=dev, dtype=torch.int32)
v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
k1m, k1nulls, _ = B.bernoulli_mask(n, 0.10, 47, dev)
vm, vnulls, _ = B.bernoulli_mask(n, 0.10, 48, dev)
def run(name, k1, k1mask, vmask):
    keys = cudf_amd.Table([cudf_amd.Column.from_torch(k0), cudf_amd.Column.from_torch(k1, *(k1mask or ()))])
    vals = cudf_amd.Column.from_torch(v, *(vmask or ()))
    def f():
        grp = gb.GroupBy(keys, NullPolicy.EXCLUDE)
        return grp.aggregate([gb.GroupByRequest(vals, [agg.mean(), agg.min(), agg.max()])], stream=torch.cuda.current_stream())
    f(); f(); torch.cuda.synchronize()
    _lib.profile_reset(); _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(3):
        r = None; r = f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3 * 1e3
    _lib.profile_enable(False)
    print(name, round(dt, 2), "ms", {a: round(b[1] / 3, 2) for a, b in _lib.profile_report().items()}, flush=True)
run("int64,int32, no nulls        ", k1_32, None, None)
run("int64,int32, key nulls       ", k1_32, (k1m, k1nulls), None)
run("int64,int32, key+value nulls ", k1_32, (k1m, k1nulls), (vm, vnulls))
k1_64 = k1_32.to(torch.int64)
run("int64,int64, no nulls        ", k1_64, None, None)
