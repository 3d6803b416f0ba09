# A/B of one environment switch on the headline step, interleaved: tools/ab_env.sh NAME "v1 v2 v1 v2" [extra bench flags]
name=$1; vals=$2; shift 2
for v in $vals; do
  env $name=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name=$v', round(j['value'],1), 'it/s', round(j['ms_per_step'],3),'ms', j['step_ms_hip_events']['p50'])"
done
