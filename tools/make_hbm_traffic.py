"""profiles/rNN_bench_<workload>_pmc_{FETCH,WRITE}_SIZE.csv (tools/summarize_rocprof.py output) -> profiles/hbm_traffic.json:
HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, KiB counters x 1024, launch-weighted mean over the template variants
of a kernel) for the kernels bench.py's roofline leg can name.  Usage: make_hbm_traffic.py <round tag, e.g. r01> [workload]"""
import csv, json, os, sys

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "cfg3_1M_1080p"
short = workload.split("_")[0]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
OPS = {"hashgrid_fwd": "hashgrid_fwd_kernel", "hashgrid_bwd": "hashgrid_bwd_kernel<true", "hashgrid_bwd_input": "hashgrid_bwd_kernel<false",
       "rasterize_2dgs_fwd": "raster_fwd_kernel", "rasterize_2dgs_bwd": "raster_bwd_kernel", "mlp_fwd": "mlp_fwd_kernel",
       "mlp_bwd_data": "mlp_bwd_data_kernel", "mlp_bwd_weights": "mlp_bwd_weights_kernel", "l1_dssim_fwd": "l1_dssim_fwd_kernel",
       "l1_dssim_bwd": "l1_dssim_bwd_kernel", "adam": "adam_kernel"}
tot = {k: [0.0, 0] for k in OPS}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    seen = {k: 0 for k in OPS}
    for r in csv.DictReader(open(os.path.join(root, f"{tag}_bench_{short}_pmc_{counter}.csv"))):
        for op, pat in OPS.items():
            if pat in r["Kernel_Name"] and r["Counter_Name"] == counter:
                tot[op][0] += float(r["sum"]) * 1024.0
                seen[op] += int(r["launches"])
    for op in OPS:
        tot[op][1] = max(tot[op][1], seen[op])
out = {op: int(v / n) for op, (v, n) in tot.items() if n}
out["_note"] = (f"{tag}: FETCH_SIZE + WRITE_SIZE per launch in bytes (KiB counters x1024, mean over all launches of the run; FETCH "
                "uncorrected for the gfx950 x2 under-report of wide reads: the gather kernels read 4-36 B pieces). Collected with "
                "bench.py --no-overlap so that one kernel runs at a time.")
path = os.path.join(root, "hbm_traffic.json")
allw = json.load(open(path)) if os.path.exists(path) else {}
allw[workload] = out
json.dump(allw, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
