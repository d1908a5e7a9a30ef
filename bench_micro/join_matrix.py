#!/usr/bin/env python3
"""Looks for performance pathologies of the hash join: 100M probe rows x 10M build rows over a matrix of key
distributions (run on the GPU box from the repo root). Every case checks the pair count against a torch count."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cudf_amd
from cudf_amd import join as J
from cudf_amd.types import NullEquality
dev = torch.device("cuda", 0)
L = int(os.environ.get("JOIN_L", 100_000_000))
R = int(os.environ.get("JOIN_R", 10_000_000))
C = cudf_amd.Column.from_torch
g = torch.Generator(device=dev).manual_seed(3)


def rnd(lo, hi, n):
    return torch.randint(lo, hi, (n,), generator=g, device=dev, dtype=torch.int64)


def cases():
    perm = torch.randperm(R, generator=g, device=dev).to(torch.int64)
    yield "unique build, uniform probe (30% hit)", perm, torch.where(torch.rand(L, generator=g, device=dev) < 0.3, rnd(0, R, L), rnd(R, 2 * R, L))
    yield "unique build, all probes hit", perm, rnd(0, R, L)
    yield "unique build, no probe hits", perm, rnd(R, 2 * R, L)
    yield "unique build, sorted probe", perm, torch.sort(rnd(0, 2 * R, L)).values
    yield "sorted build, sorted probe", torch.arange(R, device=dev, dtype=torch.int64), torch.sort(rnd(0, 2 * R, L)).values
    yield "keys = i << 32 (hash quality)", perm << 32, rnd(0, 2 * R, L) << 32
    yield "keys = i * 2^20 + 7", perm * (1 << 20) + 7, rnd(0, 2 * R, L) * (1 << 20) + 7
    yield "build 8 duplicates per key, probe 10% hit", rnd(0, R // 8, R), torch.where(torch.rand(L, generator=g, device=dev) < 0.1, rnd(0, R // 8, L), rnd(R, 2 * R, L))
    hot = perm.clone()
    hot[:100_000] = 5  # one build key with 100,000 duplicates
    yield "build with one key x 100,000; probes miss it", hot, rnd(R, 2 * R, L)
    yield "build with one key x 100,000; 1000 probes hit it", hot, torch.cat([rnd(R, 2 * R, L - 1000), torch.full((1000,), 5, device=dev, dtype=torch.int64)])
    yield "probe all one key (present once)", perm, torch.full((L,), 12345, device=dev, dtype=torch.int64)
    yield "probe Zipf over the build keys", perm, torch.clamp((float(R) ** torch.rand(L, generator=g, device=dev, dtype=torch.float64)).to(torch.int64) - 1, 0, R - 1)


for name, b, p in cases():
    bt, pt = cudf_amd.Table([C(b)]), cudf_amd.Table([C(p)])
    # expected pair count: sum over probe rows of the multiplicity of their key in the build
    ub, cnt = torch.unique(b, return_counts=True)
    idx = torch.searchsorted(ub, p).clamp(max=ub.numel() - 1)
    expect = int(cnt[idx][ub[idx] == p].sum())
    del ub, cnt, idx
    best = None
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        li, ri = J.inner_join(pt, bt, NullEquality.EQUAL, stream=torch.cuda.current_stream())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    pairs = li.size() if hasattr(li, "size") and callable(li.size) else len(li)
    flag = "  <== SLOW" if best * 1e3 > 25 else ""
    print(f"{name:52s}: {best*1e3:9.2f} ms  pairs={pairs:>11} ok={pairs == expect}{flag}", flush=True)
    del bt, pt, li, ri, b, p
