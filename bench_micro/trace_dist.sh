#!/bin/bash
# kernel timeline of one step of the literal config-5 form rehearsed at world 1 (run on the GPU box from the repo root)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/trace_dist
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BENCH_FORCE_DISTRIBUTED=1 BENCH_VARIANT_TIMEOUT_S=1
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob("$OUT/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Size", "")))
rows.sort()
# the last step: from the last k_estimate backwards to the previous range-partition kernel
idx = [i for i, r in enumerate(rows) if "k_dense_ring_scatter" in r[2]]
last = idx[-1]
start = max(0, last - 40)
prev = rows[start][0]
for s, e, n in rows[start:last + 12]:
    print(f"gap {(s - prev) / 1e3:9.1f} us  dur {(e - s) / 1e3:9.1f} us  {n}")
    prev = e
PY
