#!/bin/bash
# Run ON THE GPU BOX: kernel stats + counter passes of the compositing kernels alone (tools/exp_raster_quads.py child, cfg3 shape).
# usage: tools/prof_raster.sh <tag>      -> gpurun_out/prof_raster_<tag>/
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_raster_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/exp_raster_quads.py cfg3_1M_1080p child /tmp/x.pt"
rm -rf /tmp/p && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -- $CMD > $OUT/line.json 2>> $OUT/log.txt
python $REPO/tools/summarize_rocprof.py /tmp/p $OUT/${TAG}_raster_kernel_stats.csv > /dev/null
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  name=$(echo $c | tr ' ' '+')
  rm -rf /tmp/p && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p -- $CMD > /dev/null 2>> $OUT/log.txt
  python $REPO/tools/summarize_rocprof.py /tmp/p $OUT/${TAG}_raster_pmc_$name.csv > /dev/null
done
grep -h "raster_\|unpack" $OUT/*.csv | grep -v "true" | cut -c1-60,300- | head -80
