#!/bin/bash
# parameter sweep of the write-combining scatter on C2 (run on the GPU box from the repo root)
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print(round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items()}, d['config']['path'])"; }
for rep in 1 2; do
run CUDF_AMD_GB_WC=1
run CUDF_AMD_GB_WC_G=4
run CUDF_AMD_GB_WC_G=4 CUDF_AMD_GB_RPT=7
done
