import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cudf_amd
from cudf_amd import join as J
from cudf_amd.types import NullEquality
import bench_configs as B
dev = torch.device("cuda", 0)
nl, nr = 500_000_000, 50_000_000
g = torch.Generator(device=dev).manual_seed(12345)
rk = torch.randperm(nr, generator=g, device=dev).to(torch.int64)
sel = torch.rand(nl, generator=g, device=dev) < 0.3
lk = torch.where(sel, torch.randint(0, nr, (nl,), generator=g, device=dev, dtype=torch.int64), torch.randint(nr, 2 * nr, (nl,), generator=g, device=dev, dtype=torch.int64))
del sel
lm, lnulls, lvalid = B.bernoulli_mask(nl, 0.05, 44, dev)
rm, rnulls, rvalid = B.bernoulli_mask(nr, 0.05, 45, dev)
L = cudf_amd.Table([cudf_amd.Column.from_torch(lk, lm, lnulls)])
R = cudf_amd.Table([cudf_amd.Column.from_torch(rk, rm, rnulls)])
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    li, ri = J.inner_join(L, R, NullEquality.UNEQUAL, stream=torch.cuda.current_stream())
    torch.cuda.synchronize(); print("step", i, round((time.perf_counter() - t0) * 1e3, 2), "ms", flush=True)
    del li, ri
