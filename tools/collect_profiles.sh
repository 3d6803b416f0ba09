#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the round's rocprofv3 evidence into gpurun_out/prof_<tag>/.
# usage: tools/collect_profiles.sh r01
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1) kernel stats of the DEFAULT bench command (two-leg overlapped step)
rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_cfg3_under_rocprof.json 2> $OUT/stats.log
python $REPO/tools/summarize_rocprof.py /tmp/p1 $OUT/${TAG}_bench_cfg3_kernel_stats.csv > /dev/null
python $REPO/tools/trace_timeline.py /tmp/p1 $OUT/${TAG}_bench_cfg3_step_timeline.txt > /dev/null
# 2) the same work on ONE stream (kernel durations without neighbours)
rm -rf /tmp/p2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-overlap > $OUT/${TAG}_bench_cfg3_serial_under_rocprof.json 2>> $OUT/stats.log
python $REPO/tools/summarize_rocprof.py /tmp/p2 $OUT/${TAG}_bench_cfg3_serial_kernel_stats.csv > /dev/null
# 3) HBM traffic counters, one counter per run, single stream
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p3 && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p3 -- python $REPO/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-overlap > /dev/null 2>> $OUT/stats.log
  python $REPO/tools/summarize_rocprof.py /tmp/p3 $OUT/${TAG}_bench_cfg3_pmc_$c.csv > /dev/null
done
ls -la $OUT
