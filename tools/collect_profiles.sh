#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the round's rocprofv3 evidence into gpurun_out/prof_<tag>/.
# ONE workload per run: the headline command only (--no-secondary --no-cpu-baseline), so that one CSV row = one kernel of one workload; the
# detail record written next to each CSV (…detail.json: all_steps.mean_sdf_points, mean_M, mean_I over EVERY step the process ran, warm-up
# included) holds the unit counts the row's average belongs to.
# usage: tools/collect_profiles.sh r06 [quick]
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-secondary"
run() {   # run <name> <rocprof args...> -- <bench args...>: the bench's detail record lands beside the CSV
  local name=$1; shift
  export GSDF_BENCH_DETAIL=prof_$TAG/${TAG}_bench_cfg3_$name.detail.json
  rm -rf /tmp/p && rocprofv3 "$@" > $OUT/${TAG}_bench_cfg3_$name.line.json 2>> $OUT/stats.log
}
# 1) kernel stats + timeline of the DEFAULT bench command (two-leg overlapped step)
run overlapped --kernel-trace --stats --output-format csv -d /tmp/p -- $B --steps 20 --warmup 5
python $REPO/tools/summarize_rocprof.py /tmp/p $OUT/${TAG}_bench_cfg3_kernel_stats.csv > /dev/null
python $REPO/tools/trace_timeline.py /tmp/p $OUT/${TAG}_bench_cfg3_step_timeline.txt > /dev/null
[ "${2:-}" = quick ] && { ls -la $OUT; exit 0; }
# 2) the same work on ONE stream (kernel durations without neighbours): the table the roofline fractions are reproducible from
run serial --kernel-trace --stats --output-format csv -d /tmp/p -- $B --steps 20 --warmup 5 --no-overlap
python $REPO/tools/summarize_rocprof.py /tmp/p $OUT/${TAG}_bench_cfg3_serial_kernel_stats.csv > /dev/null
# 3) counters, one group per run, single stream (--pmc only with --kernel-trace).  TCP / TCC / TA groups: the hash-grid forward's L1 / L2 side
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  name=$(echo $c | tr ' ' '+')
  run pmc_$name --kernel-trace --pmc $c --output-format csv -d /tmp/p -- $B --steps 5 --warmup 1 --no-overlap
  python $REPO/tools/summarize_rocprof.py /tmp/p $OUT/${TAG}_bench_cfg3_pmc_$name.csv > /dev/null
done
# 4) round 6: the L1 / L2 / TA side of the kernels the TWO-STREAM step launches (the RESIDENT hash-grid forward, 3 workgroups per CU): the headline
#    command without --no-overlap (counter collection serialises the kernels: the resident kernel's own figures)
for c in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  name=$(echo $c | tr ' ' '+')
  run pmc2s_$name --kernel-trace --pmc $c --output-format csv -d /tmp/p -- $B --steps 5 --warmup 1
  python $REPO/tools/summarize_rocprof.py /tmp/p $OUT/${TAG}_bench_cfg3_pmc2s_$name.csv > /dev/null
done
ls -la $OUT
