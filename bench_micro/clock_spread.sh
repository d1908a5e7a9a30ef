#!/bin/bash
# run-to-run spread of the C2 scatter beside the chip's clocks ("log rocm-smi clocks beside the spread"):
# five separate processes of the headline command, rocm-smi clocks / power before and after each (run on the GPU box from the repo root)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
for i in 1 2 3 4 5; do
  echo "== process $i"
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -6
  python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readlines()[-1])
r = d['roofline']
print('ms_per_step %.3f  scatter %.3f ms  aggregate %.3f ms  frac %.3f' % (d['ms_per_step'], r['kernels_ms_per_step'].get('partition_scatter', 0), r['kernels_ms_per_step'].get('aggregate', 0), r['frac']))"
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -6
done
