for P in 512 1024; do for RPT in 4 8; do for LDS in 72 150; do for SL in 512 1024; do
  export CUDF_AMD_GB_P=$P CUDF_AMD_GB_RPT=$RPT CUDF_AMD_GB_LDS_KB=$LDS CUDF_AMD_GB_SLICES=$SL
  echo -n "P=$P RPT=$RPT LDS=$LDS SLICES=$SL : "
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print(round(d['ms_per_step'],2), 'hist',round(k['partition_hist'],2),'scat',round(k['partition_scatter'],2),'agg',round(k['aggregate'],2))"
done; done; done; done
