run() { name=$1; shift; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err; python - <<EOF
import json
try:
    j=json.loads(open("gpurun_out/$name.json").read().strip().splitlines()[-1])
    print("$name", round(j["value"],1), round(j["ms_per_step"],2), j["roofline"]["ms_per_step_by_kernel"])
except Exception as e: print("$name", "ERR", e, open("gpurun_out/$name.err").read()[-500:])
EOF
}
run unmasked
run unmasked_rayside --ray-leg-on-scatter-xcds 0
run unmasked_noaux --ray-weights-aux 0
run xcd2 --scatter-xcds 2
run xcd4 --scatter-xcds 4
GSDF_BENCH_HOST_TIMES=1 run unmasked_host
grep "host ms" gpurun_out/unmasked_host.err
