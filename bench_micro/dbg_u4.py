import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import cudf_amd
from cudf_amd import aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy
from bench_configs import bernoulli_mask
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
g = torch.Generator(device=dev).manual_seed(46)
k0 = torch.randint(0, 10_000, (n,), generator=g, device=dev, dtype=torch.int64)
k1 = torch.randint(0, 1_000, (n,), generator=g, device=dev, dtype=torch.int64)
v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
vm, vnulls, _ = bernoulli_mask(n, 0.10, 48, dev)
C = cudf_amd.Column.from_torch
for i in range(3):
    t0 = time.perf_counter()
    grp = gb.GroupBy(cudf_amd.Table([C(k0), C(k1)]), NullPolicy.EXCLUDE)
    uk, res = grp.aggregate([gb.GroupByRequest(C(v, vm, vnulls), [agg.mean(), agg.min(), agg.max()])])
    torch.cuda.synchronize()
    print("call", i, round((time.perf_counter() - t0) * 1e3, 1), "ms groups", uk.num_rows(), flush=True)
