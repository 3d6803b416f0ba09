#!/bin/bash
# Run ON THE GPU BOX: counter passes on tools/exp_scatter.py (binned scatter kernels alone).  usage: [SCRIPT=exp_scatter2.py] tools/prof_scatter.sh [B]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_scatter; mkdir -p $OUT; rm -f $OUT/summary.txt
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $c | tr ' ' '+')
  rm -rf /tmp/pm && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pm -- python $REPO/tools/${SCRIPT:-exp_scatter.py} ${1:-3290000} > /dev/null 2> $OUT/err_$name.log
  python - "$name" <<'PY' >> $OUT/summary.txt
import csv, glob, sys, collections
f = glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in f:
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].split('(')[0][:40]
        if 'bin_' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(sys.argv[1][:30], '|', k, '|', {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
cat $OUT/summary.txt
