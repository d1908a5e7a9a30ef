#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json at full size on one MI355X (not the headline line; `bench.py` is):
  C3  hash inner_join, 500M x 50M int64 keys, 5% nulls on both sides, selectivity 0.3 (UNEQUAL, as the reference's
      benchmark default cpp/benchmarks/join/join_common.hpp:42) + gather of 2+2 float64 payload columns
  C4  groupby keys (int64, int32), value float64, 10M groups, 10% nulls on k1 and on the value, MEAN+MIN+MAX
Prints one JSON line per config with per-kernel times (HIP events on the launch stream) and property checks.
Usage: python bench_configs.py [c3] [c4] [--scale 1.0]
"""
import json
import sys
import time

import torch

import cudf_amd
from cudf_amd import _lib, aggregation as agg, groupby as gb, join as J, partitioning
from cudf_amd.types import NullEquality, NullPolicy


def bernoulli_mask(n, p_null, seed, dev):
    """int32 bitmask words with P(bit = 0) = p_null, plus the null count."""
    g = torch.Generator(device=dev).manual_seed(seed)
    nwords = (n + 31) // 32
    valid = torch.rand(nwords * 32, generator=g, device=dev) >= p_null
    valid[n:] = False
    w = (valid.view(nwords, 32).to(torch.int64) << torch.arange(32, device=dev, dtype=torch.int64)).sum(1)
    words = (w & 0xFFFFFFFF).to(torch.int64)
    words = torch.where(words >= 2**31, words - 2**32, words).to(torch.int32)
    pad = torch.zeros(16, dtype=torch.int32, device=dev)
    return torch.cat([words, pad]), int(n - valid[:n].sum().item()), valid[:n]


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    _lib.profile_reset()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = None  # the previous result goes back to the pool first: a caller keeps one result alive, not two
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    _lib.profile_enable(False)
    prof = {k: v[1] / steps for k, v in sorted(_lib.profile_report().items())}
    return out, dt, prof


def make_c4(scale, dev=None):
    """-> (run callable, check callable(result) -> dict, rows, algorithmic bytes)."""
    dev = dev or torch.device("cuda", 0)
    n = int(1_000_000_000 * scale)
    g = torch.Generator(device=dev).manual_seed(46)
    k0 = torch.randint(0, 10_000, (n,), generator=g, device=dev, dtype=torch.int64)
    k1 = torch.randint(0, 1_000, (n,), generator=g, device=dev, dtype=torch.int32)
    v = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
    k1m, k1nulls, k1valid = bernoulli_mask(n, 0.10, 47, dev)
    vm, vnulls, vvalid = bernoulli_mask(n, 0.10, 48, dev)
    keys = cudf_amd.Table([cudf_amd.Column.from_torch(k0), cudf_amd.Column.from_torch(k1, k1m, k1nulls)])
    vals = cudf_amd.Column.from_torch(v, vm, vnulls)

    def run():
        grp = gb.GroupBy(keys, NullPolicy.EXCLUDE)
        return grp, grp.aggregate([gb.GroupByRequest(vals, [agg.mean(), agg.min(), agg.max()])], stream=torch.cuda.current_stream())

    def check(result):
        """Per-group torch reference (as the full-size C2 test does): the groups are exactly the distinct valid key pairs, every
        group's COUNT-derived validity, MIN and MAX are bit-exact, every MEAN is sum / count within the order-of-summation bound
        (kat.sum_atol: m^2 * eps * max|v| for groups of at most m rows)."""
        import numpy as np
        grp, (uk, res) = result
        G = uk.num_rows()
        NG = 10_000 * 1_000
        combo = k0 * 1000 + k1.to(torch.int64)
        cnt_rows = torch.bincount(combo[k1valid], minlength=NG)          # rows per key pair (NULL k1 dropped: EXCLUDE)
        sel = k1valid & vvalid
        idx, vs = combo[sel], v[sel]
        exp_cnt = torch.bincount(idx, minlength=NG)                      # valid values per key pair
        exp_sum = torch.zeros(NG, dtype=torch.float64, device=dev).scatter_add_(0, idx, vs)
        exp_min = torch.full((NG,), float("inf"), dtype=torch.float64, device=dev).scatter_reduce_(0, idx, vs, "amin", include_self=True)
        exp_max = torch.full((NG,), float("-inf"), dtype=torch.float64, device=dev).scatter_reduce_(0, idx, vs, "amax", include_self=True)
        del idx, vs, sel, combo
        distinct = int((cnt_rows > 0).sum())
        gk0, gk1 = [c.to_torch() for c in uk.columns()]
        gid = gk0 * 1000 + gk1.to(torch.int64)
        keys_in_range = bool(((gid >= 0) & (gid < NG)).all())
        gid = gid.clamp(0, NG - 1)
        keys_once = bool(torch.unique(gid).numel() == G) and bool((cnt_rows[gid] > 0).all())
        cols = res[0].columns()
        mean, mn, mx = [c.to_torch() for c in cols]
        valid = [torch.from_numpy(c.to_numpy()[1]).to(dev) if c.nullable() else torch.ones(G, dtype=torch.bool, device=dev) for c in cols]
        has = exp_cnt[gid] > 0  # a group whose values are all NULL yields NULL results
        masks_ok = all(bool((vd == has).all()) for vd in valid)
        m = float(exp_cnt.max())
        tol = m * m * float(np.finfo(np.float64).eps)
        c_g = exp_cnt[gid].clamp(min=1).to(torch.float64)
        mean_err = float(((mean - exp_sum[gid] / c_g).abs() * c_g)[has].max()) if bool(has.any()) else 0.0
        min_ok = bool((mn[has] == exp_min[gid][has]).all())
        max_ok = bool((mx[has] == exp_max[gid][has]).all())
        ok = has
        out = {"groups": G, "distinct_pairs": distinct, "groups_ok": G == distinct and keys_in_range and keys_once,
               "masks_ok": masks_ok, "per_group_min_ok": min_ok, "per_group_max_ok": max_ok,
               "per_group_mean_ok": mean_err <= tol, "mean_sum_err": mean_err, "mean_sum_tol": tol,
               # the properties of the earlier check, kept so that old and new records can be compared
               "global_max_ok": bool(float(mx[ok].max()) == float(exp_max.max())) if bool(ok.any()) else True,
               "global_min_ok": bool(float(mn[ok].min()) == float(exp_min.min())) if bool(ok.any()) else True,
               "mean_within_min_max": bool(((mean[ok] >= mn[ok]) & (mean[ok] <= mx[ok])).all()), "path": grp.last_path.name}
        out["all_ok"] = all(out[k] for k in ("groups_ok", "masks_ok", "per_group_min_ok", "per_group_max_ok", "per_group_mean_ok"))
        return out

    return run, check, n, n * (8 + 4 + 8) + 2 * n / 8


def c4(scale):
    run, check, n, algo_bytes = make_c4(scale)
    result, dt, prof = timed(run, 3, 2)
    checks = check(result)
    print(json.dumps({"config": "C4", "rows": n, "groups": checks["groups"], "ms": dt * 1e3, "rows_per_s": n / dt,
                      "algorithmic_GBps": algo_bytes / dt / 1e9, "path": checks["path"], "kernels_ms": prof, "checks": checks}), flush=True)


def make_c3(scale, dev=None, sparse=False, inrange=False):
    """-> (run callable, check callable(result) -> dict, L + R rows, algorithmic bytes given the pair count).
    sparse: the same tables with every key multiplied by an odd constant (keys no longer span a small range: the hash table
    instead of the direct-address table).
    inrange: dense keys whose MISSES lie inside the build side's key range - build = the even keys of [0, 2 nr), a probe row
    that matches carries an even key, one that does not an odd key of the same range - so that the direct-address table's range test
    rejects nothing and every probe row costs a table access."""
    dev = dev or torch.device("cuda", 0)
    nl, nr = int(500_000_000 * scale), int(50_000_000 * scale)
    g = torch.Generator(device=dev).manual_seed(12345)
    rk = torch.randperm(nr, generator=g, device=dev).to(torch.int64)  # unique build keys [0, nr)
    sel = torch.rand(nl, generator=g, device=dev) < 0.3
    lk = torch.where(sel, torch.randint(0, nr, (nl,), generator=g, device=dev, dtype=torch.int64),
                     torch.randint(nr, 2 * nr, (nl,), generator=g, device=dev, dtype=torch.int64))
    del sel
    if inrange:
        rk = rk * 2
        lk = torch.where(lk < nr, lk * 2, (lk - nr) * 2 + 1)
    lm, lnulls, lvalid = bernoulli_mask(nl, 0.05, 44, dev)
    rm, rnulls, rvalid = bernoulli_mask(nr, 0.05, 45, dev)
    present = torch.zeros(2 * nr + 2, dtype=torch.bool, device=dev)
    present[rk[rvalid]] = True
    expect = int((present[lk] & lvalid).sum())
    del present
    if sparse:
        lk, rk = lk * 1_000_003, rk * 1_000_003
    L = cudf_amd.Table([cudf_amd.Column.from_torch(lk, lm, lnulls)])
    R = cudf_amd.Table([cudf_amd.Column.from_torch(rk, rm, rnulls)])

    def run():
        return J.inner_join(L, R, NullEquality.UNEQUAL, stream=torch.cuda.current_stream())

    def check(result):
        li, ri = result
        M = li.size()
        li_t, ri_t = li.to_torch().long(), ri.to_torch().long()
        return {"pairs": M, "expected_pairs": expect, "count_ok": M == expect, "keys_equal": bool((lk[li_t] == rk[ri_t]).all()),
                "no_null_rows": bool(lvalid[li_t].all() and rvalid[ri_t].all()),
                "pairs_distinct": bool(torch.unique(li_t * (2 * nr) + ri_t).numel() == M) if M < 200_000_000 else None}

    return run, check, nl + nr, lambda M: 8 * (nl + nr) + (nl + nr) / 8 + 8 * M, (nl, nr, dev)


def c3(scale, sparse=False, inrange=False):
    run, check, rows, algo, (nl, nr, dev) = make_c3(scale, sparse=sparse, inrange=inrange)
    (li, ri), dt, prof = timed(run, 3, 2)
    checks = check((li, ri))
    M = checks["pairs"]
    li_t = li.to_torch().long()
    algo_bytes = algo(M)
    out = {"config": "C3" + (" (sparse keys: hash table)" if sparse else "") + (" (dense keys, misses inside the key range)" if inrange else ""), "left_rows": nl, "right_rows": nr, "pairs": M, "join_ms": dt * 1e3,
           "rows_per_s": (nl + nr) / dt, "algorithmic_GBps": algo_bytes / dt / 1e9, "kernels_ms": prof, "checks": checks}
    # gather 2 + 2 float64 payload columns by the returned indices
    lp = cudf_amd.Table([cudf_amd.Column.from_torch(torch.rand(nl, device=dev, dtype=torch.float64)) for _ in range(2)])
    rp = cudf_amd.Table([cudf_amd.Column.from_torch(torch.rand(nr, device=dev, dtype=torch.float64)) for _ in range(2)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gl = partitioning.gather(lp, li)
    gr = partitioning.gather(rp, ri)
    torch.cuda.synchronize()
    out["gather_ms"] = (time.perf_counter() - t0) * 1e3
    out["checks"]["gather_ok"] = bool((gl.columns()[0].to_torch() == lp.columns()[0].to_torch()[li_t]).all())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    scale = 1.0
    if "--scale" in sys.argv:
        scale = float(sys.argv[sys.argv.index("--scale") + 1])
    which = args or ["c4", "c3"]
    if "c4" in which:
        c4(scale)
        torch.cuda.empty_cache()
    if "c3" in which:
        c3(scale)
        torch.cuda.empty_cache()
    if "c3sparse" in which:
        c3(scale, sparse=True)
        torch.cuda.empty_cache()
    if "c3inrange" in which:
        c3(scale, inrange=True)
