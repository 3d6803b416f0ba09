for v in 2 0 1 2 0; do
  GSDF_RASTER_ROW_LISTS=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --ray-batch pool 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows=$v', round(j['value'],1), 'it/s', round(j['ms_per_step'],3),'ms', j['step_ms_hip_events']['p50'])"
done
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>gpurun_out/bench_sampled.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sampled ray batch', round(j['value'],1), 'it/s', j['config']['ray_batch'])"
tail -3 gpurun_out/bench_sampled.err
