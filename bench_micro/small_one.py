import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cudf_amd
from cudf_amd import _lib, aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy
dev = torch.device("cuda", 0)
C = cudf_amd.Column.from_torch
g = torch.Generator(device=dev).manual_seed(3)
n, groups = int(sys.argv[1]), int(sys.argv[2])
k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
kc, vc = C(k), C(v)
for _ in range(20):
    grp = gb.GroupBy(cudf_amd.Table([kc]))
    r = grp.aggregate([gb.GroupByRequest(vc, [agg.sum(), agg.count(NullPolicy.EXCLUDE)])], stream=torch.cuda.current_stream())
torch.cuda.synchronize()
