# The joint iteration at the other BASELINE shapes (parity-test shapes, not bench lines): tools/bench_other_workloads.sh rNN  ->  gpurun_out/rNN_bench_other_workloads.json
TAG=${1:-r04}
python - <<EOF
import json, subprocess, sys
out = []
for args in (["--workload", "cfg1_replica_300k"], ["--workload", "cfg1_replica_300k", "--no-sdf"], ["--workload", "cfg4_3M_640x512_K16"], ["--workload", "cfg0_10k_256"]):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-secondary"] + args, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        out.append({k: j[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "step_ms_hip_events", "config") if k in j})
        print(args, round(j["value"], 1), flush=True)
    except Exception as e:
        print(args, "ERR", e, r.stderr[-400:], flush=True)
json.dump(out, open("gpurun_out/${TAG}_bench_other_workloads.json", "w"), indent=1)
EOF
