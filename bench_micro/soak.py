#!/usr/bin/env python3
"""Soak: the headline shape (1B rows, 1M groups) over several seeds and two skews, every result checked exactly against
torch.bincount (counts) and a float64 scatter-add (sums, relative 1e-9).   soak.py [rows] [groups] [seed,seed,...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cudf_amd
from cudf_amd import aggregation as agg, groupby as gb
from cudf_amd.types import NullPolicy
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
seeds = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(8))
C = cudf_amd.Column.from_torch
for seed in seeds:
    g = torch.Generator(device=dev).manual_seed(1000 + seed)
    if seed < 6:
        k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
    else:  # skew: a fifth of the rows on 16 hot keys (forces the exact pipeline when a region overflows)
        k = torch.randint(0, groups, (n,), generator=g, device=dev, dtype=torch.int64)
        hot = torch.rand(n, generator=g, device=dev) < 0.2
        k[hot] = k[hot] % 16
        del hot
    v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
    grp = gb.GroupBy(cudf_amd.Table([C(k)]))
    uk, res = grp.aggregate([gb.GroupByRequest(C(v), [agg.sum(), agg.count(NullPolicy.EXCLUDE)])], stream=torch.cuda.current_stream())
    keys = uk.columns()[0].to_torch()
    s, c = [x.to_torch() for x in res[0].columns()]
    exp_c = torch.bincount(k, minlength=groups)
    exp_s = torch.zeros(groups, dtype=torch.float64, device=dev).scatter_add_(0, k, v)
    present = exp_c > 0
    ok_keys = keys.numel() == int(present.sum()) and bool((torch.sort(keys).values == torch.nonzero(present).flatten()).all())
    ok_c = bool((c.to(torch.int64) == exp_c[keys]).all())
    rel = float(((s - exp_s[keys]).abs() / exp_s[keys].abs().clamp_min(1e-300)).max())
    print(f"seed {seed}: groups {keys.numel()} path {grp.last_path.name} keys_ok {ok_keys} counts_ok {ok_c} max rel sum err {rel:.2e}", flush=True)
    assert ok_keys and ok_c and rel < 1e-9
    del k, v, keys, s, c, exp_c, exp_s, uk, res
print("soak OK")
