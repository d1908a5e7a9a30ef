#!/bin/bash
# dynamic instruction mix of the groupby kernels on the off-fast-path shapes (run on the GPU box from the repo root)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmci
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT -o i -- python3 $ROOT/bench_micro/shapes.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections, re
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.OrderedDict()
for r in rows:
    m = re.search(r"(k_aggregate|k_partition_scatter\w*)(<[^>]*>)?", r["Kernel_Name"])
    if not m: continue
    key = (r["Dispatch_Id"], m.group(0)[:70])
    by.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
seen = set()
for (d, k), c in by.items():
    if k in seen: continue
    seen.add(k)
    w = c.get("SQ_WAVES", 1)
    print(k, {a: round(b / 1e9 * 64, 2) for a, b in c.items() if a != "SQ_WAVES"}, "(x1e9 lane-instr per 1B rows = instr per row), waves", int(w))
PY
