#!/usr/bin/env python3
"""Summarises the rocprofv3 output of collect_profiles.sh: kernel_stats.txt (top kernels by time) and
pmc_traffic.json (HBM bytes per launch of the groupby kernels, gfx950 corrections applied)."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha256():
    """Hash of the kernel / host sources the library is built from: bench.py quotes the PMC traffic only for a matching tree."""
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


def label_map():
    """kernel function name -> profiler labels, read off the sources: every hipLaunchKernelGGL behind a prof::scope of its function."""
    import re
    m = {}
    base = os.path.join(ROOT, "cudf_amd", "csrc")
    for f in glob.glob(os.path.join(base, "**", "*.hip"), recursive=True) + glob.glob(os.path.join(base, "**", "*.cpp"), recursive=True):
        label = None
        for line in open(f, errors="ignore"):
            sc = re.search(r'prof::scope \w+\{([^}]*)\}', line)
            if sc:
                label = re.findall(r'"([a-z0-9_]+)"', sc.group(1))
            for k in re.findall(r'hipLaunchKernelGGL\(\(?\s*(k_[A-Za-z0-9_]+)', line) + re.findall(r'CUDF_AMD_DENSE_VARIANT\((k_[A-Za-z0-9_]+)', line):
                if label:
                    m.setdefault(k, set()).update(label)
            if line.startswith("}"):
                label = None
    return {k: sorted(v) for k, v in m.items()}


def labels_for(full, kernel, targs, labels_of):
    """Profiler labels of one dispatch. Two kernel names serve two labels each, told apart by their first template argument (the level),
    and one name (k_radix_scatter) exists in two namespaces (the joins' ring scatter, the sort-based groupby's radix pass)."""
    first = targs[1:].split(",")[0].strip() if targs else ""
    if kernel == "k_radix_scatter":
        if "join" in full:
            return ["join_partition_level2" if first == "2" else "join_partition"]
        return ["sort_scatter"]
    if kernel == "k_dense_ring_scatter":
        return ["partition_scatter_level2" if first == "2" else "partition_scatter"]
    return labels_of.get(kernel, [])


def config_traffic():
    """PMC traffic per kernel NAME for the C3 / C3-sparse / C4 runs (pmc_<config>_<counter>/): pmc_traffic_configs.json."""
    import re
    doc = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --config <c> "
                  "--steps 2 --warmup 1 --no-cpu-baseline`; per-launch averages over every dispatch of a kernel; read bytes = 2 x FETCH_SIZE "
                  "(gfx950: 128-B requests tallied at 64 B), WRITE_SIZE as is; warm-up launches included in the average", "configs": {}}
    labels_of = label_map()
    doc["sources_sha256"] = sources_sha256()
    doc["how"] += ("; `labels`: HBM bytes per STEP of the kernels that run under each profiler label of the library (the names bench.py's "
                   "`roofline.kernel` carries), = sum over their dispatches / 3 runs (2 steps + 1 warm-up); a kernel that runs under two "
                   "labels (k_radix_join: count and retrieve) is charged to both; the two levels of a ring scatter are told apart by the kernel's first template argument")
    doc["labels"] = {}
    RUNS = 3
    for c in ("c3", "c3sparse", "c3inrange", "c4"):
        per = {}
        lab = defaultdict(float)
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            fs = glob.glob(os.path.join(out, f"pmc_{c}_{counter}", "**", "*counter_collection.csv"), recursive=True)
            if not fs:
                continue
            acc = defaultdict(lambda: [0.0, set()])
            for r in csv.DictReader(open(fs[0])):
                if r.get("Counter_Name") != counter:
                    continue
                full = r.get("Kernel_Name", "")
                m = re.search(r"(k_[a-z0-9_]+)(<[^(]*>)?", full)  # k_name<template arguments>
                name = (m.group(1) + (m.group(2) or "")[:48]) if m else full[:64]
                acc[name][0] += float(r["Counter_Value"])
                acc[name][1].add(r.get("Dispatch_Id"))
                if m:
                    for l in labels_for(full, m.group(1), m.group(2) or "", labels_of):
                        lab[l] += float(r["Counter_Value"]) * 1024 * (2 if counter == "FETCH_SIZE" else 1) / RUNS
            for k, (tot, ids) in acc.items():
                per.setdefault(k, {})[counter + "_KB_per_launch"] = tot / max(1, len(ids))
                per[k]["launches"] = len(ids)
        for k, d in per.items():
            d["read_GB"] = round(2 * d.get("FETCH_SIZE_KB_per_launch", 0) * 1024 / 1e9, 3)
            d["write_GB"] = round(d.get("WRITE_SIZE_KB_per_launch", 0) * 1024 / 1e9, 3)
        if not per:
            continue
        doc["configs"][c] = {k: v for k, v in sorted(per.items(), key=lambda kv: -(kv[1]["read_GB"] + kv[1]["write_GB"]))[:10]}
        doc["labels"][c] = {l: {"hbm_bytes_per_step": b} for l, b in sorted(lab.items())}
    json.dump(doc, open(os.path.join(out, "pmc_traffic_configs.json"), "w"), indent=1)
    print(json.dumps({c: {k: (v["read_GB"], v["write_GB"]) for k, v in d.items()} for c, d in doc["configs"].items()})[:1500])


if len(sys.argv) > 2 and sys.argv[2] == "configs":
    config_traffic()
    sys.exit(0)


def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


def short(name):
    if "k_dense_ring_scatter<2" in name:  # (the level that reads level-1 regions)
        return "partition_scatter_level2"
    for key, label in (("k_dense_ring_scatter", "partition_scatter"), ("k_dense_merge_dump", "aggregate_merge"),
                       ("k_partition_scatter", "partition_scatter"), ("k_partition_hist", "partition_hist"),
                       ("k_aggregate", "aggregate"), ("k_finalize", "finalize"), ("k_estimate", "estimate")):
        if key in name:
            return label
    return None


# ---- kernel stats
f = find("stats", "*kernel_stats.csv")
lines = []
if f:
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("Total_Duration", 0)) or 0))
    lines.append(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary   ({os.path.basename(f)})")
    lines.append(f"{'calls':>6} {'avg_ms':>10} {'total_ms':>10} {'pct':>6}  kernel")
    for r in rows[:14]:
        tot = float(r.get("TotalDurationNs", 0)) / 1e6
        avg = float(r.get("AverageNs", 0)) / 1e6
        lines.append(f"{r.get('Calls', '?'):>6} {avg:10.3f} {tot:10.3f} {float(r.get('Percentage', 0)):6.2f}  {r.get('Name', '')[:150]}")
open(os.path.join(out, "kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:8]))

# ---- PMC traffic
res = {}
for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r.get("Kernel_Name", ""))
        if k is None:
            continue
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1].add(r.get("Dispatch_Id"))
    for k, (tot, ids) in acc.items():
        res.setdefault(k, {})[counter + "_KB"] = tot / max(1, len(ids))
for k, d in res.items():
    rd = 2 * d.get("FETCH_SIZE_KB", 0) * 1024  # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
    wr = d.get("WRITE_SIZE_KB", 0) * 1024
    d.update(read_bytes_corrected=rd, write_bytes=wr, hbm_bytes_per_launch=rd + wr)
doc = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py "
              "--steps 3 --warmup 1 --no-cpu-baseline` (C2, 1B rows), bench_micro/collect_profiles.sh; per-launch averages. "
              "Units: counter KB (x1024 B). gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests "
              "at 64 B, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is taken as is.",
       "sources_sha256": sources_sha256(),
       "kernels": res}
json.dump(doc, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e9, 2) for k, v in res.items()}))
