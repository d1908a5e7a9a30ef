for i in 1 2 3; do
for v in 1 0; do
CUDF_AMD_GB_DENSE_LINES=$v python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('lines=$v', round(d['ms_per_step'],3), {a:round(b,3) for a,b in k.items()})"
done; done
