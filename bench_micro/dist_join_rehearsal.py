#!/usr/bin/env python3
"""distributed_inner_join over RCCL at world_size 1 and C3/2.5 scale (200M x 20M): every pair joins equal keys and the
pair count equals a torch membership count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from cudf_amd import distributed as D
g = torch.Generator(device=dev).manual_seed(9)
nl, nr = 200_000_000, 20_000_000
rk = torch.randperm(2 * nr, generator=g, device=dev)[:nr].to(torch.int64)
lk = torch.randint(0, 4 * nr, (nl,), generator=g, device=dev, dtype=torch.int64)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gl, gr = D.distributed_inner_join(lk, rk)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
present = torch.zeros(4 * nr, dtype=torch.bool, device=dev); present[rk] = True
print("pairs", gl.numel(), "expected", int(present[lk].sum()), "keys equal", bool((lk[gl] == rk[gr]).all()), f"{dt*1e3:.1f} ms", flush=True)
dist.destroy_process_group()
