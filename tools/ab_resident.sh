#!/bin/bash
# A/B of GSDF_HASHGRID_RESIDENT (resident grid of w workgroups per CU for the stencil hash-grid forward) on the headline step
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["ms_per_step_by_kernel"]; print(round(d["value"],1), round(d["ms_per_step"],3), "hg_fwd", k.get("hashgrid_fwd"), "alone", d["roofline"].get("avg_launch_ms"))'
for r in ${@:-0 2 0 2 0 2}; do echo "resident=$r overlapped"; GSDF_HASHGRID_RESIDENT=$r python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P"; done
