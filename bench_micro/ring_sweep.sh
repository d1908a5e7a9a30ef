#!/bin/bash
# C2 through the ring path: partition fan-out (CUDF_AMD_GB_DENSE_LOG2P) x aggregate workgroups per partition (CUDF_AMD_GB_DENSE_NSPLIT)
for l in 7 8; do for ns in 1 2 4; do
  echo "== LOG2P=$l NSPLIT=$ns"
  CUDF_AMD_GB_DENSE_LOG2P=$l CUDF_AMD_GB_DENSE_NSPLIT=$ns timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['roofline']['kernels_ms_per_step'])"
done; done
