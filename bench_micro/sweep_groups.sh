#!/bin/bash
# C2 shape at other cardinalities (run on the GPU box from the repo root): sweep_groups.sh [groups...]
for G in ${@:-100 1000 100000 1000000 10000000 100000000}; do
  echo -n "groups=$G : "
  timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --groups $G 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print(round(d['ms_per_step'],2), 'ms', round(d['value']/1e9,1), 'G rows/s', {a:round(b,2) for a,b in k.items()}, d['config']['path'])"
done
