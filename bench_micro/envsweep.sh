#!/bin/bash
# bench_micro/envsweep.sh "VAR1=a VAR2=b" "VAR1=c" ...: bench.py C2 under each environment, one summary line each
for E in "$@"; do
  echo -n "$E : "
  env $E timeout -k 10 150 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print(round(d['ms_per_step'],2), d['config'].get('path'), {a:round(b,2) for a,b in k.items()})"
done
