#!/bin/bash
# Run ON THE GPU BOX: counter passes on tools/exp_mlp.py (decoder kernels alone).  usage: tools/prof_mlp.sh [B]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_mlp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export EXP_TORCH_TOPOLOGY=${EXP_TORCH_TOPOLOGY:-0}
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  name=$(echo $c | tr ' ' '+')
  rm -rf /tmp/pm && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pm -- python $REPO/tools/exp_mlp.py ${1:-3290000} > /dev/null 2> $OUT/err_$name.log
  python - "$name" <<'PY' >> $OUT/summary.txt
import csv, glob, sys, collections
f = glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in f:
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].split('(')[0][:60]
        if 'mlp' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(sys.argv[1], '|', k, '|', {c: round(sum(v) / len(v)) for c, v in d.items()}, 'launches', len(next(iter(d.values()))))
PY
done
cat $OUT/summary.txt
