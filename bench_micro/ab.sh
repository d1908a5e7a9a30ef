#!/bin/bash
# A/B of library builds on C2: bench_micro/ab.sh lib1.so lib2.so ... (each run twice, interleaved)
for rep in 1 2; do for L in "$@"; do
  export CUDF_AMD_LIB=$PWD/$L
  echo -n "$L : "
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print(round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items()})"
done; done
