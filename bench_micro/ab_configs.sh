#!/bin/bash
# A/B of two library builds over the secondary configurations, alternating: bench_micro/libcudf_amd_base.so (CUDF_AMD_LIB) against the tree's build
for c in ${AB_CONFIGS:-c4 c3sparse c3}; do
for i in 1 2; do
for v in new base; do
if [ $v = base ]; then export CUDF_AMD_LIB=$PWD/bench_micro/libcudf_amd_base.so; else unset CUDF_AMD_LIB; fi
python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('roofline',{}).get('kernels_ms_per_step',{}); print('$c $v', round(d['ms_per_step'],3), {a:round(b,3) for a,b in k.items()})"
done; done; done
for v in new base; do
if [ $v = base ]; then export CUDF_AMD_LIB=$PWD/bench_micro/libcudf_amd_base.so; else unset CUDF_AMD_LIB; fi
echo "multi_value $v"; python bench_micro/multi_value.py 2>/dev/null | grep "groups= 1000000"
done
