#!/bin/bash
# per-kernel durations of the sorted / clustered shapes (bench_micro/sorted_keys.py) - run on the GPU box from the repo root
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/trace_sorted
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $ROOT/bench_micro/sorted_keys.py ${1:-1000} > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e6:10.3f} {float(r['TotalDurationNs'])/1e6:10.3f}  {r['Name'][:160]}")
PY
