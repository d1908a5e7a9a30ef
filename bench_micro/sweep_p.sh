#!/bin/bash
# sweep of the partition count / slice count on C2 (run on the GPU box from the repo root)
for P in 512 1024; do for SL in 128 256 512; do
  export CUDF_AMD_GB_P=$P CUDF_AMD_GB_SLICES=$SL
  echo -n "P=$P SLICES=$SL : "
  timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print(round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items()}, d['config']['path'])"
done; done
