#!/bin/bash
# Collects the rocprofv3 evidence for the headline workload (run on the GPU box from the repo root):
#   1. --kernel-trace --stats          -> per-kernel average durations
#   2. --kernel-trace --pmc FETCH_SIZE -> HBM read traffic   (separate pass, MI355X_MICROARCH.md HBM section)
#   3. --kernel-trace --pmc WRITE_SIZE -> HBM write traffic  (separate pass)
# and writes gpurun_out/prof/{kernel_stats.txt,pmc_traffic.json}; copy them to profiles/ (named per round).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
echo "== stats" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1 &&
echo "== fetch" && timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1 &&
echo "== write" && timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1 &&
python3 $ROOT/bench_micro/summarize_profiles.py $OUT
# C3 / C4 kernel stats of the same build
for c in c3 c4; do timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$c -o stats -- python3 $ROOT/bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline > $OUT/stats_$c.log 2>&1; done
# PMC traffic of the C3 / C3-sparse / C4 kernels: the same two counter passes per configuration
if [ "${PMC_CONFIGS:-1}" = "1" ]; then
  for c in c3 c3sparse c3inrange c4; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${c}_$ctr -o pmc -- python3 $ROOT/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_${c}_$ctr.log 2>&1
    done
  done
  python3 $ROOT/bench_micro/summarize_profiles.py $OUT configs
fi
