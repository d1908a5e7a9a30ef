#!/bin/bash
# A/B of two library builds on one box, alternating: bench_micro/libcudf_amd_base.so (CUDF_AMD_LIB) against the tree's build
for i in 1 2 3; do
for v in new base; do
if [ $v = base ]; then export CUDF_AMD_LIB=$PWD/bench_micro/libcudf_amd_base.so; else unset CUDF_AMD_LIB; fi
python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline ${AB_ARGS} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$v', round(d['ms_per_step'],3), {a:round(b,3) for a,b in k.items()})"
done; done
