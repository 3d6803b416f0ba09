for v in 1 0 1 0; do
  GSDF_MLP_LEAN_BWD_BWD=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean=$v', round(j['value'],1), 'it/s', round(j['ms_per_step'],3),'ms', j['step_ms_hip_events']['p50'])"
done
