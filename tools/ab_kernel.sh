#!/bin/bash
# Run ON THE GPU BOX: one experiment script under rocprofv3 --kernel-trace --stats with the product library and with compile-time variants of it
# (tools/build_variants.sh); prints the script's own output and the kernels matching a pattern.
# usage: tools/ab_kernel.sh <kernel-name pattern> "<python script + args>" <variant> [...]     ("base" = the product library)
PAT=$1; CMD=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
L=$REPO/gs-sdf_amd/lib
cp $L/libgsdf_hip.so /tmp/base.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/base.so $L/libgsdf_hip.so; else cp $L/variants/$v/libgsdf_hip.so $L/libgsdf_hip.so; fi
  rm -rf /tmp/abk
  echo "== $v"
  (cd $REPO && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -- python $CMD 2>/dev/null | grep -v "^\s*$" | tail -5)
  python - "$PAT" <<'PY'
import csv, glob, sys, re
for p in glob.glob('/tmp/abk/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if re.search(sys.argv[1], r['Name']):
            print(f"   {r['Name'].split('(')[0][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
PY
done
cp /tmp/base.so $L/libgsdf_hip.so
