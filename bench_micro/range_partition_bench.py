"""cudf::distributed::range_partition at the C5 shape (1B rows, int64 key + float64 value) for 2/4/8 destinations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cudf_amd
from cudf_amd import _lib, distributed as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
g = torch.Generator(device="cuda").manual_seed(42)
k = torch.randint(0, 1_000_000, (n,), generator=g, device="cuda", dtype=torch.int64)
v = torch.rand(n, generator=g, device="cuda", dtype=torch.float64)
t = cudf_amd.Table([cudf_amd.Column.from_torch(k), cudf_amd.Column.from_torch(v)])
for nd in (2, 4, 8):
    D.range_partition(t, [0], nd)
    torch.cuda.synchronize()
    _lib.profile_reset(); _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(3):
        out, offs = D.range_partition(t, [0], nd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3 * 1e3
    _lib.profile_enable(False)
    print(nd, "destinations:", round(dt, 2), "ms/call", {a: round(b[1] / 3, 2) for a, b in _lib.profile_report().items()}, "sizes", [offs[i + 1] - offs[i] for i in range(nd)][:4], flush=True)
