#!/bin/bash
# Second half of bench_micro/collect_profiles.sh on its own (a gpurun call holds 20 minutes): the two PMC passes over the
# --config runs -> gpurun_out/prof/pmc_traffic_configs.json, then C3-in-range under the three probe choices.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in c3 c3sparse c3inrange c4; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${c}_$ctr -o pmc -- python3 $ROOT/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_${c}_$ctr.log 2>&1
  done
done
python3 $ROOT/bench_micro/summarize_profiles.py $OUT configs
