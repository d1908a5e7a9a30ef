#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): rows/s of hash-groupby SUM+COUNT over 1B int64-key / float64-value rows
with 1M groups (configs[1], "C2"), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic rows already resident in HBM:
  N == 1: cudf::groupby::groupby(keys).aggregate({values, [SUM, COUNT_VALID]}) through the C ABI.
  N  > 1: configs[4] ("C5", weak scaling, 1B rows per GPU). `value` is the LITERAL form of the config, inside the library:
          cudf::distributed::shuffle_groupby = hash-range partition of the local ROWS by owner rank -> counts by ncclAllGather,
          rows by ncclSend / ncclRecv over xGMI -> per-GPU groupby. It is bound by the point-to-point xGMI links ((N-1)/N of
          16 GB per GPU over N-1 links). The COMBINER form (cudf::distributed::combine_groupby: per-GPU groupby -> exchange of
          the <= 1M partial groups -> merge groupby on the owner; identical result, xGMI carries MBs) is timed in the same run
          after the primary region and reported beside it as `preaggregated_variant`. BENCH_DIST_MODE=preaggregate swaps
          which of the two is `value`; BENCH_COMBINER=torch runs the combiner through torch.distributed instead of the library.
Rank 0 prints ONE JSON line. `roofline` is measured live with HIP events on the launch stream (the library's
per-kernel profiler); `cpu_baseline` times the CPU oracle (oracle/, test infrastructure) on a bounded sample of the
same workload on this box's host cores, plus pandas/Arrow on configs[0] (10M rows).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)
BYTES_PER_ROW = 16     # algorithmic bytes per row of C2: 8 B key + 8 B value


def _sources_sha256():
    """Same hash as bench_micro/summarize_profiles.py: the sources the library is built from."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "cudf_amd", "csrc")
    for d, _, fs in sorted(os.walk(base)):
        if os.sep + "build" in d:
            continue
        for f in sorted(fs):
            if f.endswith((".hip", ".cpp", ".hpp", ".inl", ".h")):
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=int(os.environ.get("BENCH_ROWS", 1_000_000_000)))
    ap.add_argument("--groups", type=int, default=int(os.environ.get("BENCH_GROUPS", 1_000_000)))
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c3sparse", "c3inrange", "c4"],
                    help="c2 (default, the headline); c3 = configs[2] inner join; c3sparse = the same with keys that do not span a small "
                         "range (hash table instead of the direct-address table); c3inrange = dense keys whose misses lie INSIDE the build side's key range "
                         "(no free range reject); c4 = configs[3] multi-key groupby. One GPU only.")
    ap.add_argument("--scale", type=float, default=1.0, help="c3 / c4: fraction of the BASELINE size (parity / smoke runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="c2 at one GPU: skip the variants (hash-table keys, two value columns) and the C3 / C3-sparse / C4 runs that the "
                         "headline line otherwise carries in `hash_table_variant`, `two_value_columns_variant` and `secondary`")
    ap.add_argument("--variant-steps", type=int, default=5)
    ap.add_argument("--cpu-sample-rows", type=int, default=60_000_000)
    return ap.parse_args()


def cpu_baseline(groups, sample_rows):
    """Times the CPU oracle (kind "port", 1 core) on a bounded sample of the C2 workload, and pandas / Arrow on
    configs[0] (10M rows) for the north_star's "reference CPU Arrow/pandas" comparison."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(42)
    k = rng.integers(0, groups, sample_rows, dtype=np.int64)
    v = rng.random(sample_rows)
    t0 = time.perf_counter()
    O.groupby([k], [(v, ["sum", "count_valid"])])
    dt = time.perf_counter() - t0
    out = {"value": sample_rows / dt, "unit": "rows/s", "cores": 1, "kind": "port",
           "sample": f"{sample_rows} rows of the same int64-key/float64-value workload, {groups} groups, "
                     f"oracle/oracle.c groupby SUM+COUNT_VALID, {dt:.1f} s",
           "host_cores": os.cpu_count()}
    try:
        import pandas as pd
        import pyarrow as pa
        n = 10_000_000
        kk, vv = k[:n], v[:n]
        df = pd.DataFrame({"k": kk, "v": vv})
        best = min(_timeit(lambda: df.groupby("k", sort=False)["v"].sum()) for _ in range(3))
        out["pandas_rows_per_s"] = n / best
        tb = pa.table({"k": kk, "v": vv})
        best = min(_timeit(lambda: tb.group_by("k", use_threads=True).aggregate([("v", "sum")])) for _ in range(3))
        out["arrow_rows_per_s"] = n / best
        out["arrow_threads"] = pa.cpu_count()
        out["pandas_arrow_sample"] = f"configs[0]: {n} rows, {groups} groups, best of 3 (pandas {pd.__version__}, pyarrow {pa.__version__})"
    except Exception as e:  # pandas/pyarrow are optional on the box
        out["pandas_arrow_error"] = repr(e)
    return out


def cpu_baseline_config(config, scale_rows):
    """CPU oracle (kind "port", 1 core) on a bounded sample of the C3 / C4 workload."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(42)
    if config.startswith("c3"):
        nl, nr = 20_000_000, 2_000_000
        rk = rng.permutation(nr).astype(np.int64)
        lk = np.where(rng.random(nl) < 0.3, rng.integers(0, nr, nl), rng.integers(nr, 2 * nr, nl)).astype(np.int64)
        lv, rv = rng.random(nl) > 0.05, rng.random(nr) > 0.05
        t0 = time.perf_counter()
        n_pairs = O.join_size([(lk, lv)], [(rk, rv)], nulls_equal=False, kind="inner")
        dt = time.perf_counter() - t0
        return {"value": (nl + nr) / dt, "unit": "rows/s", "cores": 1, "kind": "port", "host_cores": os.cpu_count(),
                "sample": f"{nl} x {nr} rows of the same int64-key workload (5% nulls, selectivity 0.3, UNEQUAL), oracle/oracle.c inner join "
                          f"size pass ({n_pairs} pairs), {dt:.1f} s"}
    n = 30_000_000
    k0 = rng.integers(0, 10_000, n, dtype=np.int64)
    k1 = rng.integers(0, 1_000, n).astype(np.int32)
    v = rng.random(n)
    k1v, vv = rng.random(n) > 0.1, rng.random(n) > 0.1
    t0 = time.perf_counter()
    O.groupby([k0, (k1, k1v)], [((v, vv), ["mean", "min", "max"])])
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "rows/s", "cores": 1, "kind": "port", "host_cores": os.cpu_count(),
            "sample": f"{n} rows of the same (int64, int32)-key / float64-value workload, 10% nulls on k1 and on the value, ~10M groups, "
                      f"oracle/oracle.c groupby MEAN+MIN+MAX, {dt:.1f} s"}


def run_config(args, json_fd):
    """--config c3 | c3sparse | c4: the same JSON shape as the headline line, for BASELINE configs[2] / configs[3] on one GPU."""
    import torch
    import bench_configs as BC
    from cudf_amd import _lib
    torch.cuda.set_device(0)
    if args.config == "c4":
        run, check, rows, algo_bytes = BC.make_c4(args.scale)
        algo = lambda _res: algo_bytes  # noqa: E731
        metric = "rows/s hash-groupby MEAN+MIN+MAX, 1B rows, keys (int64, int32), 10M groups, 10% nulls"
        workload = "C4: 1xMI355X multi-key groupby (int64+int32) with mean/min/max, 1B rows, 10M groups, 10% nulls on k1 and on the value"
    else:
        run, check, rows, algo_of, _ = BC.make_c3(args.scale, sparse=args.config == "c3sparse", inrange=args.config == "c3inrange")
        algo = lambda res: algo_of(res[0].size())  # noqa: E731
        metric = "rows/s hash inner_join, 500M x 50M int64 keys, 5% nulls"
        workload = ("C3: 1xMI355X hash inner_join, 500M x 50M int64 keys with 5% null mask, selectivity 0.3, null_equality::UNEQUAL"
                    + (" [keys x 1,000,003: sparse, served by the hash table]" if args.config == "c3sparse" else "")
                    + (" [build = the even keys of [0, 100M), misses = odd keys of the same range: no range reject]" if args.config == "c3inrange" else ""))
    for _ in range(max(args.warmup, 1)):
        run()
    torch.cuda.synchronize()
    _lib.profile_reset()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        res = None
        res = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    prof = _lib.profile_report()
    checks = check(res)
    nbytes = algo(res)
    ms = dt / args.steps * 1e3
    name, (launches, total_ms) = max(prof.items(), key=lambda kv: kv[1][1])
    avg_ms = total_ms / max(launches, 1)
    per_launch = nbytes * (args.steps / max(launches, 1))  # a kernel launched several times per step shares the step's bytes
    # counter traffic of this label's kernels at this configuration and scale (bench_micro/collect_profiles.sh), quoted only if it was
    # taken from the source tree this library is built from
    traffic, traffic_note, traffic_all = None, None, None
    try:
        pmc_file = next(f for f in ("r4_pmc_traffic_configs.json", "r3_pmc_traffic_configs.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
        with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
            doc = json.load(f)
        if doc.get("sources_sha256") != _sources_sha256():
            traffic_note = f"profiles/{pmc_file} was collected from a different source tree: not quoted"
        elif args.scale != 1.0:
            traffic_note = "counters were taken at scale 1.0"
        else:
            lab = doc.get("labels", {}).get(args.config, {})
            if name in lab:
                traffic = lab[name]["hbm_bytes_per_step"] / max(launches / args.steps, 1)
            traffic_all = {k: v["hbm_bytes_per_step"] for k, v in lab.items()}
    except Exception as e:  # noqa: BLE001
        traffic_note = f"no PMC file: {e!r}"
    line = {"metric": metric, "value": rows * args.steps / dt, "unit": "rows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64" if args.config.startswith("c3") else "f64", "data": "synthetic",
            "config": {"workload": workload, "scale": args.scale, "rows": rows, "checks": checks},
            "roofline": {"bound": "hbm", "kernel": name, "achieved": per_launch / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": per_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "traffic_bytes_per_step_by_label": traffic_all, "avg_launch_ms": avg_ms,
                         "launches_per_step": launches / args.steps, "algorithmic_bytes_per_step": nbytes,
                         "kernels_ms_per_step": {k: v[1] / args.steps for k, v in sorted(prof.items())},
                         "whole_call_frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_config(args.config, rows)
    os.write(json_fd, (json.dumps(line) + "\n").encode())


def run_secondary_children():
    """configs[2] (dense and sparse keys) and configs[3] at full size, one child process each (python bench.py --config ...)."""
    import subprocess
    out = {}
    budget_s = float(os.environ.get("BENCH_SECONDARY_TIMEOUT_S", "100"))
    # (c3_partitioned_probe: C3 with the probe rows partitioned by key range instead of the ordered direct probe that is the default
    # - less time in the join, pairs partition-major instead of in probe-row order: DESIGN.md section 4)
    # (c3inrange: dense keys whose misses lie inside the build side's key range - the direct-address table without its free range reject)
    for name, cfg, env_extra in (("c3", "c3", {}), ("c3_partitioned_probe", "c3", {"CUDF_AMD_JOIN_DENSE_PROBE": "2"}), ("c3sparse", "c3sparse", {}),
                                 ("c3inrange", "c3inrange", {}), ("c4", "c4", {})):
        t0 = time.perf_counter()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=budget_s, cwd=ROOT, env={**os.environ, **env_extra})
            lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                out[name] = {"error": f"exit code {p.returncode}", "stderr_tail": p.stderr.decode(errors="replace")[-400:]}
                continue
            doc = json.loads(lines[-1])
            checks = doc["config"]["checks"]
            out[name] = {"ms_per_step": doc["ms_per_step"], "value": doc["value"], "unit": doc["unit"], "steps": doc["steps"],
                        "frac": doc["roofline"]["frac"], "whole_call_frac": doc["roofline"]["whole_call_frac"],
                        "dominant_kernel": doc["roofline"]["kernel"], "kernels_ms_per_step": doc["roofline"]["kernels_ms_per_step"],
                        "traffic": doc["roofline"].get("traffic"), "traffic_bytes_per_step_by_label": doc["roofline"].get("traffic_bytes_per_step_by_label"),
                        "traffic_note": doc["roofline"].get("traffic_note"),
                        "checks": checks, "checks_ok": all(v for k, v in checks.items()
                                         if isinstance(v, bool) and (k.endswith("_ok") or k in ("keys_equal", "no_null_rows", "pairs_distinct"))),
                        "wall_s": time.perf_counter() - t0}
        except subprocess.TimeoutExpired:
            out[name] = {"error": f"not finished after {budget_s:.0f} s"}
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": repr(e)}
    return out


def _timeit(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def main():
    args = parse()
    # Libraries print banners on the C-level stdout (RCCL: "RCCL version : ...", "Hostname : ..."): everything but the
    # one JSON line goes to stderr; the line itself is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.config != "c2":
        assert args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1, "--config c3 / c4 run on one GPU"
        return run_config(args, json_fd)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import cudf_amd
    from cudf_amd import _lib, aggregation as agg, groupby as gb
    from cudf_amd.types import NullPolicy

    n, groups = args.rows, args.groups
    gen = torch.Generator(device=dev).manual_seed(42 + rank)
    keys = torch.randint(0, groups, (n,), generator=gen, device=dev, dtype=torch.int64)
    gen.manual_seed(43 + rank)
    vals = torch.rand(n, generator=gen, device=dev, dtype=torch.float64)
    stream = torch.cuda.current_stream()

    force_dist = os.environ.get("BENCH_FORCE_DISTRIBUTED") == "1"  # rehearse the N>1 code path on one GPU
    # N > 1: `value` is the LITERAL config-5 form, inside the library (cudf::distributed::shuffle_groupby: hash-range partition of
    # the rows -> RCCL Send/Recv -> per-GPU groupby); the combiner form is timed in the same run and reported beside it.
    # BENCH_DIST_MODE = shuffle_native (default) | shuffle (the same through torch.distributed) | preaggregate (combiner as `value`)
    dist_mode = os.environ.get("BENCH_DIST_MODE", "shuffle_native")
    other_mode = "shuffle_native" if dist_mode == "preaggregate" else "preaggregate"
    if world == 1 and force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world == 1 and not force_dist:
        kcol, vcol = cudf_amd.Column.from_torch(keys), cudf_amd.Column.from_torch(vals)

        def step():
            g = gb.GroupBy(cudf_amd.Table([kcol]))
            out = g.aggregate([gb.GroupByRequest(vcol, [agg.sum(), agg.count(NullPolicy.EXCLUDE)])], stream=stream)
            return g, out
    else:
        from cudf_amd import distributed as D

        def step():
            return D.distributed_groupby_sum_count(keys, vals, stream=stream, mode=dist_mode)

        # The in-library exchange has only ever run at world size 1 (this pool hands out one GPU). If its first step RAISES on any
        # rank (RCCL not loadable, communicator creation refused), every rank falls back to the same exchange through
        # torch.distributed - still the literal config-5 form - and the line says so. (A collective that hangs cannot be recovered.)
        if dist_mode == "shuffle_native":
            first_error = None
            # (a collective that HANGS on its first use cannot be recovered in-process: a watchdog ends this rank with a non-zero
            # exit code instead of leaving the driver waiting - no re-exec, the GPU has been initialised)
            import threading

            def give_up():
                sys.stderr.write(f"[bench] rank {rank}: the first in-library exchange did not finish within {first_timeout_s} s; exiting\n")
                sys.stderr.flush()
                os._exit(3)
            first_timeout_s = int(os.environ.get("BENCH_FIRST_STEP_TIMEOUT_S", "300"))
            first_watchdog = threading.Timer(first_timeout_s, give_up)
            first_watchdog.daemon = True
            first_watchdog.start()
            try:
                step()
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                first_error = repr(e)
            finally:
                first_watchdog.cancel()
            ok = torch.tensor([0 if first_error else 1], device=dev, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                sys.stderr.write(f"[bench] shuffle_native failed on some rank ({first_error}); falling back to the torch.distributed exchange\n")
                dist_mode = "shuffle"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    _lib.profile_reset()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    barrier()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    prof = _lib.profile_report()

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # ---- N == 1, after the timed region: the LAST timed step's result against a per-group torch reference of the same rows
    # (bincount / scatter_add_; nothing from oracle/). Keys and counts exactly; sums in units of the reference's own rule
    # |x - y| <= 4 eps |x + y| (cpp/tests/utilities/column_utilities.cu:436-440) and against the worst-case bound of ANY summation
    # order of m values below 1, m^2 eps (both sides add with unordered atomics, as the reference does).
    result_check = None
    if world == 1 and not force_dist:
        try:
            rkeys, rres = last[1]
            k_out = rkeys.columns()[0].to_torch()
            s_out = rres[0].columns()[0].to_torch()
            c_out = rres[0].columns()[1].to_torch().to(torch.int64)
            ref_c = torch.bincount(keys, minlength=groups)
            ref_s = torch.zeros(groups, device=dev, dtype=torch.float64).scatter_add_(0, keys, vals)
            present = int((ref_c > 0).sum().item())
            in_range = bool(((k_out >= 0) & (k_out < groups)).all().item())
            distinct = int(torch.unique(k_out).numel())
            gc = ref_c[k_out] if in_range else None
            gs = ref_s[k_out] if in_range else None
            eps = 2.220446049250313e-16
            err = (s_out - gs).abs() if in_range else None
            units = float((err / (4 * eps * (s_out + gs).abs()).clamp_min(1e-300)).max().item()) if in_range else None
            m = float(ref_c.max().item())
            result_check = {
                "groups_out": int(k_out.numel()), "groups_expected": present, "distinct_keys_out": distinct,
                "sum_of_counts": int(c_out.sum().item()), "rows": n,
                "counts_exact": bool(in_range and (c_out == gc).all().item()),
                "max_abs_sum_error": float(err.max().item()) if in_range else None,
                "max_sum_error_in_units_of_4eps_x_plus_y": units,
                "worst_case_reorder_bound_abs": m * m * eps,
                "ok": bool(in_range and k_out.numel() == present and distinct == present and int(c_out.sum().item()) == n
                           and (c_out == gc).all().item() and float(err.max().item()) <= m * m * eps),
                "reference": "torch.bincount / scatter_add_ over the same rows (float64 atomics, unordered)",
            }
            del ref_c, ref_s, gc, gs, err
        except Exception as e:  # noqa: BLE001
            result_check = {"error": repr(e)}

    # ---- N == 1, after the primary timed region: the other side of the cliffs next to the headline, in the driver's own record.
    # hash_table_variant: the same rows with every key multiplied by an odd constant (a bijection: same groups, same counts, but
    # the keys no longer span a small range, so the open-addressing hash tables serve the call instead of the direct-address ones);
    # two_value_columns_variant: the same keys with a second float64 column, {a: SUM, COUNT; b: SUM};
    # secondary: BASELINE configs[2] / configs[3] (and the sparse-key form of the join) at full size, each in a CHILD process (its
    # own HIP context: a failure there cannot cost the headline its line), with the property checks of bench_configs.py.
    extras = {}
    if world == 1 and not force_dist and not args.no_secondary and os.environ.get("BENCH_SECONDARY", "1") != "0":
        primary_groups = last[1][0].num_rows()
        last = (last[0], None)

        def timed_variant(make_step, what):
            try:
                vstep = make_step()
                vstep()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                out = None
                for _ in range(args.variant_steps):
                    out = None
                    out = vstep()
                torch.cuda.synchronize()
                vdt = (time.perf_counter() - t1) / args.variant_steps
                g, (uk, _res) = out
                return {"ms_per_step": vdt * 1e3, "value": n / vdt, "unit": "rows/s", "steps": args.variant_steps, "path": g.last_path.name,
                        "groups": uk.num_rows(), "groups_match_primary": uk.num_rows() == primary_groups,
                        "whole_call_frac": BYTES_PER_ROW * n / vdt / 1e9 / HBM_PEAK_GBS, "what": what}
            except Exception as e:  # noqa: BLE001 - reported in the line
                return {"error": repr(e), "what": what}

        def make_hash_step():
            skeys = keys * 1_000_003
            scol = cudf_amd.Column.from_torch(skeys)

            def vstep():
                g = gb.GroupBy(cudf_amd.Table([scol]))
                return g, g.aggregate([gb.GroupByRequest(vcol, [agg.sum(), agg.count(NullPolicy.EXCLUDE)])], stream=stream)
            vstep.keep = skeys
            return vstep

        def make_two_value_step():
            gen.manual_seed(44 + rank)
            vals2 = torch.rand(n, generator=gen, device=dev, dtype=torch.float64)
            v2col = cudf_amd.Column.from_torch(vals2)

            def vstep():
                g = gb.GroupBy(cudf_amd.Table([kcol]))
                return g, g.aggregate([gb.GroupByRequest(vcol, [agg.sum(), agg.count(NullPolicy.EXCLUDE)]),
                                       gb.GroupByRequest(v2col, [agg.sum()])], stream=stream)
            vstep.keep = vals2
            return vstep

        extras["hash_table_variant"] = timed_variant(
            make_hash_step, "C2 with keys x 1,000,003 (same rows, same groups; sparse keys: open-addressing LDS hash tables)")
        torch.cuda.empty_cache()
        extras["two_value_columns_variant"] = timed_variant(
            make_two_value_step, "C2 with a second float64 value column: {a: SUM + COUNT_VALID, b: SUM} in one call (24 algorithmic bytes per row)")
        if "ms_per_step" in extras["two_value_columns_variant"]:
            tv = extras["two_value_columns_variant"]
            tv["whole_call_frac"] = 24 * n / (tv["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        torch.cuda.empty_cache()
        extras["secondary"] = run_secondary_children()

    def build_line(pre, pre_error):
        ms_per_step = dt / args.steps * 1e3
        total_rows = n * world
        value = total_rows * args.steps / dt
        # dominant kernel by total time in the timed region
        roof = None
        if prof:
            name, (launches, total_ms) = max(prof.items(), key=lambda kv: kv[1][1])
            # one full-size launch per step at N == 1; in the distributed forms the same kernel also runs on the small
            # merge input: charge all of a step's launches to the step's algorithmic bytes (conservative)
            avg_ms = total_ms / max(launches if launches <= args.steps else args.steps, 1)
            achieved = BYTES_PER_ROW * n / (avg_ms * 1e-3) / 1e9
            traffic, traffic_note = None, None
            try:  # PMC traffic of this kernel at this workload, collected with rocprofv3 (see the file's "how"): quoted only
                # if it was taken from the source tree this library is built from - otherwise it is stale and says so
                pmc_file = next(f for f in ("r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
                with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
                    doc = json.load(f)
                if doc.get("sources_sha256") != _sources_sha256():
                    traffic_note = f"profiles/{pmc_file} was collected from a different source tree: not quoted"
                elif name in doc["kernels"] and n == 1_000_000_000 and groups == 1_000_000:
                    traffic = doc["kernels"][name]["hbm_bytes_per_launch"]
            except Exception as e:  # noqa: BLE001
                traffic_note = f"no PMC file: {e!r}"
            roof = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note, "avg_launch_ms": avg_ms,
                    "algorithmic_bytes_per_launch": BYTES_PER_ROW * n,
                    "kernels_ms_per_step": {k: v[1] / args.steps for k, v in sorted(prof.items())},
                    "whole_call_frac": BYTES_PER_ROW * n / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
        line = {
            "metric": "rows/s hash-groupby-sum, 1B int64 rows/1M groups",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("C2: 1xMI355X hash-groupby SUM+COUNT_VALID, single int64 key, float64 value, no nulls"
                                    if world == 1 else
                                    ("combiner variant of C5 (NOT the BASELINE config): per-GPU groupby SUM+COUNT_VALID -> hash partition of the "
                                     "partial groups + RCCL all-to-all -> per-GPU merge groupby" if dist_mode == "preaggregate" else
                                     "C5: 8xMI355X hash-partition of the ROWS by owner rank (hash-range) + RCCL all-to-all over xGMI, then per-GPU "
                                     "groupby SUM+COUNT_VALID" + (" [inside the library: cudf::distributed::shuffle_groupby]" if dist_mode == "shuffle_native"
                                                                  else " [exchange through torch.distributed]"))),
                       "rows_per_gpu": n, "groups": groups, "key": "int64 uniform [0, groups)", "value": "float64 uniform [0,1)",
                       "path": (last[0].last_path.name if (world == 1 and not force_dist) else "PARTITION+ALLTOALL+GROUPBY:" + dist_mode)},
            "roofline": roof,
        }
        if result_check is not None:
            line["result_check"] = result_check
        if pre is not None:
            name = "raw_row_shuffle_variant" if other_mode != "preaggregate" else "preaggregated_variant"
            what = ("C5 literal form: hash-range partition of the raw ROWS by owner rank -> RCCL Send/Recv (16 GB x (N-1)/N per GPU over the "
                    "point-to-point xGMI links) -> per-GPU groupby; same result" if other_mode != "preaggregate" else
                    "local groupby -> hash-partition + RCCL all-to-all of the partial (key, sum, count) rows -> merge; "
                    "same result, xGMI carries MBs")
            line[name] = {"value": total_rows * args.steps / pre, "unit": "rows/s", "ms_per_step": pre / args.steps * 1e3,
                          "what": what, "implementation": ("cudf::distributed::combine_groupby (in the library)" if state.get("pre_mode") == "combine_native"
                                                           else "cudf_amd.distributed (torch.distributed exchange)" if other_mode == "preaggregate" else "cudf::distributed::shuffle_groupby")}
        if pre_error is not None:
            line["raw_row_shuffle_variant" if other_mode != "preaggregate" else "preaggregated_variant"] = {"error": pre_error}
        if world == 1 and not force_dist:
            line.update(extras)
        if world == 1 and not force_dist and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(groups, args.cpu_sample_rows)
        return line


    # N > 1: also time the other form of the exchange (raw-row shuffle by default); reported next to `value`. The
    # secondary form must never cost the primary number its JSON line: an exception is reported in the line, and a
    # watchdog prints the line without it if the secondary form has not finished in time (a collective that never
    # returns cannot be interrupted from Python).
    import threading
    state = {"pre": None, "pre_error": None, "printed": False}
    emit_lock = threading.Lock()

    def emit_and_maybe_exit(timed_out):
        with emit_lock:
            if state["printed"]:
                return
            state["printed"] = True
            if timed_out:
                state["pre_error"] = f"not finished after {watchdog_s} s"
            if rank == 0:
                os.write(json_fd, (json.dumps(build_line(state["pre"], state["pre_error"])) + "\n").encode())
        if timed_out:
            # the primary line is out; a collective that never returned is still a failure of the run: exit non-zero so that
            # drivers looking at the return code see it (no restart / re-exec of a process that has touched the GPU)
            os._exit(3)

    watchdog_s = int(os.environ.get("BENCH_VARIANT_TIMEOUT_S", "300"))
    if world > 1 or force_dist:
        from cudf_amd import distributed as D2

        # the combiner form comes from the library too (cudf::distributed::combine_groupby: local partials -> exchange of the partial
        # groups -> merge); BENCH_COMBINER=torch keeps the round-3 form (the same steps through torch.distributed)
        pre_mode = other_mode
        if other_mode == "preaggregate" and os.environ.get("BENCH_COMBINER", "native") == "native":
            pre_mode = "combine_native"
        state["pre_mode"] = pre_mode

        def pre_step():
            return D2.distributed_groupby_sum_count(keys, vals, stream=stream, mode=pre_mode)

        watchdog = threading.Timer(watchdog_s, emit_and_maybe_exit, args=(True,))
        watchdog.daemon = True
        watchdog.start()
        try:
            pre_step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pre_step()
            barrier()
            tp = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            state["pre"] = float(tp.item())
        except Exception as e:  # noqa: BLE001 - reported in the JSON line
            state["pre_error"] = repr(e)
        watchdog.cancel()

    emit_and_maybe_exit(False)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
