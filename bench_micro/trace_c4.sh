#!/bin/bash
# per-dispatch kernel durations of one C4 call (run on the GPU box from the repo root)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/trc4
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $ROOT/bench_configs.py ${1:-c4} > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "cudf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-14:]:
    m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
    print(m.group(0)[:60] if m else r["Kernel_Name"][:60], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3), "ms")
PY
