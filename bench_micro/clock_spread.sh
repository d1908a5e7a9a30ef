#!/bin/bash
# run-to-run spread of the C2 scatter beside the chip's clocks ("log rocm-smi clocks beside the spread"):
# three separate processes of the headline command; rocm-smi is sampled every 0.2 s WHILE each runs (data generation and the timed
# steps alike) and the highest / median sclk and power seen are printed beside the process's figures (run on the GPU box, repo root)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
for i in 1 2 3; do
  echo "== process $i"
  S=$(mktemp)
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' ' >> $S; echo >> $S; sleep 0.2; done ) &
  SP=$!
  python3 $ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readlines()[-1])
r = d['roofline']
print('ms_per_step %.3f  scatter %.3f ms  aggregate %.3f ms  frac %.3f' % (d['ms_per_step'], r['kernels_ms_per_step'].get('partition_scatter', 0), r['kernels_ms_per_step'].get('aggregate', 0), r['frac']))"
  kill $SP 2>/dev/null; wait $SP 2>/dev/null
  python3 - $S <<'PY'
import re, sys, statistics
sclk, mclk, fclk, pw = [], [], [], []
for line in open(sys.argv[1]):
    m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", line);  sclk += [int(m.group(1))] if m else []
    m = re.search(r"mclk clock level: \d+: \((\d+)Mhz\)", line);  mclk += [int(m.group(1))] if m else []
    m = re.search(r"fclk clock level: \d+: \((\d+)Mhz\)", line);  fclk += [int(m.group(1))] if m else []
    m = re.search(r"Power \(W\): ([\d.]+)", line);               pw += [float(m.group(1))] if m else []
if sclk:
    print("rocm-smi during the process (%d samples): sclk max %d median %d MHz | mclk max %d | fclk max %d | power max %.0f median %.0f W" %
          (len(sclk), max(sclk), statistics.median(sclk), max(mclk or [0]), max(fclk or [0]), max(pw or [0]), statistics.median(pw or [0])))
PY
  rm -f $S
done
