mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.out 2> gpurun_out/bench_default.err
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --step-trace gpurun_out/step_trace_final.txt > gpurun_out/bench_trace.out 2>&1
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/bench_again.out 2>&1
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --deterministic > gpurun_out/bench_det.out 2>&1
tail -1 gpurun_out/bench_default.out | cut -c1-300
tail -1 gpurun_out/bench_again.out | cut -c1-300
tail -1 gpurun_out/bench_det.out | cut -c1-300
