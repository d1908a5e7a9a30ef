#!/bin/bash
# where do the ring scatter's cycles go? SQ busy / wait / issue counters per kernel of a call (run on the GPU box from the repo root)
#   pmc_ring.sh [c2|c4]
CFG=${1:-c2}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmcr_$CFG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
[ "$CFG" != "c2" ] && ARGS="--config $CFG $ARGS"
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $SET | cut -c1-12 | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT -o $tag -- python3 $ROOT/bench.py $ARGS > $OUT/log_$tag.txt 2>&1
done
python3 - <<PY
import csv, glob, collections, re
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    by = collections.OrderedDict()
    for r in rows:
        m = re.search(r"(k_aggregate_dense|k_dense_ring_scatter<[^>]*>|k_partition_scatter\w*)", r["Kernel_Name"])
        if not m: continue
        by.setdefault((r["Dispatch_Id"], m.group(0)), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    seen = set()
    for (d, k), c in by.items():
        if k in seen: continue
        seen.add(k)
        print(k, {a: f"{b:.3e}" for a, b in c.items()})
PY
