#!/bin/bash
# L2 hit/miss counters of the join kernels on C3 (run on the GPU box from the repo root)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmcj
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT -o j -- python3 $ROOT/bench_configs.py c3 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    import re
    m = re.search(r"(k_probe_\w+|k_build|k_scan_counts)(<[^>]*>)?", k)
    if not m: continue
    k = m.group(0)
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, d in acc.items():
    c = len(n[k])
    print(k, {a: round(b / c / 1e6, 1) for a, b in d.items()}, "M per launch, hit rate", round(d.get("TCC_HIT_sum", 0) / max(1, d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0)), 3))
PY
