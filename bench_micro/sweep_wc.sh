#!/bin/bash
# parameter sweep of the write-combining scatter on C2 (run on the GPU box from the repo root)
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print(round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items()}, d['config']['path'])"; }
run CUDF_AMD_GB_PLAN_LOAD_PCT=40 CUDF_AMD_GB_LDS_KB=156
run CUDF_AMD_GB_PLAN_LOAD_PCT=40 CUDF_AMD_GB_LDS_KB=158
run CUDF_AMD_GB_PLAN_LOAD_PCT=40 CUDF_AMD_GB_LDS_KB=159
run CUDF_AMD_GB_PLAN_LOAD_PCT=40 CUDF_AMD_GB_LDS_KB=159 CUDF_AMD_GB_AGG_BLOCK=512
run CUDF_AMD_GB_PLAN_LOAD_PCT=40 CUDF_AMD_GB_LDS_KB=159 CUDF_AMD_GB_AGG_BLOCK=768
