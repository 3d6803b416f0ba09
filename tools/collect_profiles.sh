#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the round's rocprofv3 evidence into gpurun_out/prof_<tag>/.
# usage: tools/collect_profiles.sh r02 [quick]
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-secondary"
# 1) kernel stats + timeline of the DEFAULT bench command (two-leg overlapped step)
rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- $B --steps 20 --warmup 5 > $OUT/${TAG}_bench_cfg3_under_rocprof.json 2> $OUT/stats.log
python $REPO/tools/summarize_rocprof.py /tmp/p1 $OUT/${TAG}_bench_cfg3_kernel_stats.csv > /dev/null
python $REPO/tools/trace_timeline.py /tmp/p1 $OUT/${TAG}_bench_cfg3_step_timeline.txt > /dev/null
[ "${2:-}" = quick ] && { ls -la $OUT; exit 0; }   # quick: the overlapped stats + timeline only
# 2) the same work on ONE stream (kernel durations without neighbours)
rm -rf /tmp/p2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- $B --steps 20 --warmup 5 --no-overlap > $OUT/${TAG}_bench_cfg3_serial_under_rocprof.json 2>> $OUT/stats.log
python $REPO/tools/summarize_rocprof.py /tmp/p2 $OUT/${TAG}_bench_cfg3_serial_kernel_stats.csv > /dev/null
# 3) counters, one group per run, single stream (--pmc only with --kernel-trace)
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  name=$(echo $c | tr ' ' '+')
  rm -rf /tmp/p3 && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p3 -- $B --steps 5 --warmup 1 --no-overlap > /dev/null 2>> $OUT/stats.log
  python $REPO/tools/summarize_rocprof.py /tmp/p3 $OUT/${TAG}_bench_cfg3_pmc_$name.csv > /dev/null
done
ls -la $OUT
