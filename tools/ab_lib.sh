#!/bin/bash
# Run ON THE GPU BOX: bench.py (headline command, short) with the product library and with compile-time variants of it (tools/build_variants.sh),
# interleaved.  usage: tools/ab_lib.sh <reps> <variant>[,ENV=VALUE|--bench-flag=value ...] [...]    ("base" = the product library; ENV=VALUE pairs are
# exported for that run, items starting with -- are passed to bench.py, e.g. base,--hashgrid-resident=2)
REPS=$1; shift
L=gs-sdf_amd/lib
cp $L/libgsdf_hip.so /tmp/base.so
for r in $(seq $REPS); do
  for spec in "$@"; do
    v=${spec%%,*}
    envs=$(echo "$spec" | tr ',' '\n' | tail -n +2 | grep -v '^--' | tr '\n' ' ')
    flags=$(echo "$spec" | tr ',' '\n' | tail -n +2 | grep '^--' | tr '=\n' '  ')
    if [ $v = base ]; then cp /tmp/base.so $L/libgsdf_hip.so; else cp $L/variants/$v/libgsdf_hip.so $L/libgsdf_hip.so; fi
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $flags 2>/dev/null | tail -1 > /tmp/line.json
    python - "$spec" <<'PY'
import json, sys
try:
    j = json.loads(open("/tmp/line.json").read())
    k = j["roofline"]["ms_per_step_by_kernel"]
    print(sys.argv[1], round(j["value"], 1), round(j["ms_per_step"], 3), "hg_in_step", j["roofline"]["in_step"]["avg_launch_ms"], "hg_alone", j["roofline"]["avg_launch_ms"],
          "raster_bwd", k.get("rasterize_2dgs_bwd"), "raster_fwd", k.get("rasterize_2dgs_fwd"), flush=True)
except Exception as e:
    print(sys.argv[1], "ERR", e, flush=True)
PY
  done
done
cp /tmp/base.so $L/libgsdf_hip.so
