#!/usr/bin/env python3
"""C3's shape (500M x 50M rows, selectivity 0.3, unique build tuples) with keys the partitioned joins did not take before round 4:
one float64 column, (int64, int64), (int64, int32). Default against CUDF_AMD_JOIN_RADIX=0 (the open-addressing table in HBM) - run twice,
once per setting: the switch is read per join object. Usage: python bench_micro/join_key_shapes.py [scale]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cudf_amd
from cudf_amd import _lib, join as J
from cudf_amd.types import NullEquality

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda", 0)
nl, nr = int(500_000_000 * scale), int(50_000_000 * scale)
g = torch.Generator(device=dev).manual_seed(7)
rk = torch.randperm(nr, generator=g, device=dev).to(torch.int64)
sel = torch.rand(nl, generator=g, device=dev) < 0.3
lk = torch.where(sel, torch.randint(0, nr, (nl,), generator=g, device=dev, dtype=torch.int64),
                 torch.randint(nr, 2 * nr, (nl,), generator=g, device=dev, dtype=torch.int64))
expect = int(sel.sum())
del sel
C = cudf_amd.Column.from_torch


def shapes():
    yield "float64 key", [C(lk.to(torch.float64) * 1.5)], [C(rk.to(torch.float64) * 1.5)]
    yield "(int64, int64) keys", [C(lk * 1_000_003), C(lk ^ 0x5bd1e995)], [C(rk * 1_000_003), C(rk ^ 0x5bd1e995)]
    yield "(int64, int32) keys", [C(lk * 1_000_003), C((lk % 1000).to(torch.int32))], [C(rk * 1_000_003), C((rk % 1000).to(torch.int32))]
    # the same with every probe miss INSIDE the build side's value box (build = even multiples, misses = odd ones): nothing is rejected by range
    le = torch.where(lk < nr, 2 * lk, 2 * (lk - nr) + 1)
    yield "(int64, int32) keys, misses in range", [C(le * 1_000_003), C((le % 1000).to(torch.int32))], [C(2 * rk * 1_000_003), C(((2 * rk) % 1000).to(torch.int32))]


for name, lcols, rcols in shapes():
    L, R = cudf_amd.Table(lcols), cudf_amd.Table(rcols)
    for radix in ("1", "0"):
        os.environ["CUDF_AMD_JOIN_RADIX"] = radix
        J.inner_join(L, R, NullEquality.EQUAL)
        torch.cuda.synchronize()
        _lib.profile_reset(); _lib.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(3):
            out = None
            out = J.inner_join(L, R, NullEquality.EQUAL)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        _lib.profile_enable(False)
        prof = {k: round(v[1] / 3, 3) for k, v in sorted(_lib.profile_report().items())}
        print(json.dumps({"keys": name, "radix": radix == "1", "join_ms": round(ms, 3), "pairs_ok": out[0].size() == expect, "kernels_ms": prof}), flush=True)
        out = None
    del L, R, lcols, rcols
    torch.cuda.empty_cache()
