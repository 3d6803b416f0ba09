"""profiles/rNN_bench_<workload>_pmc_*.csv (tools/summarize_rocprof.py output) -> profiles/hbm_traffic.json and profiles/valu_insts.json:
  * HBM bytes per operator launch (FETCH_SIZE + WRITE_SIZE, KiB counters x 1024, mean over the launches of the run; an operator
    made of several kernels — the binned scatter — is the sum of its kernels' per-launch means),
  * wave64 VALU instructions per launch of the compositing kernels (SQ_INSTS_VALU) for bench.py's VALU-issue roofline.
Usage: make_hbm_traffic.py <round tag, e.g. r02> [workload]"""
import csv, json, os, sys

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "cfg3_1M_1080p"
short = workload.split("_")[0]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
OPS = {"hashgrid_fwd": ["hashgrid_fwd_kernel", "hashgrid_fwd_xcd_kernel", "hashgrid_fwd_stencil_kernel"],
       "hashgrid_bwd": ["bin_vmax_kernel", "bin_count_kernel", "bin_plan_kernel", "bin_emit_kernel", "bin_apply_kernel", "bin_reduce_kernel"],
       "hashgrid_bwd_atomic": ["hashgrid_bwd_kernel<true"], "hashgrid_bwd_input": ["hashgrid_bwd_kernel<false", "hashgrid_bwd_jac_kernel"],
       "rasterize_2dgs_fwd": ["raster_fwd_quads_kernel", "raster_pack_kernel", "raster_mask_kernel"], "rasterize_2dgs_bwd": ["raster_bwd_quads_kernel", "unpack_records_kernel"], "mlp_fwd": ["mlp_fwd_kernel", "mlp_fwd_split_kernel"],
       "mlp_bwd": ["mlp_bwd_split_kernel"], "mlp_bwd_data": ["mlp_bwd_data_kernel"], "mlp_bwd_weights": ["mlp_bwd_weights_kernel"], "l1_dssim_fwd": ["l1_dssim_fwd_kernel"],
       "l1_dssim_bwd": ["l1_dssim_bwd_kernel"], "adam": ["adam_kernel"]}


def per_launch(counter_file, counter):
    """kernel-name pattern -> (sum over the run, launches) for `counter`"""
    rows = [r for r in csv.DictReader(open(counter_file)) if r.get("Counter_Name") == counter]
    out = {}
    for op, pats in OPS.items():
        tot = 0.0
        n_launch = []
        for pat in pats:
            v = sum(float(r["sum"]) for r in rows if pat in r["Kernel_Name"])
            n = sum(int(r["launches"]) for r in rows if pat in r["Kernel_Name"])
            if n:
                tot += v
                n_launch.append(n)
        if n_launch:
            out[op] = (tot, max(n_launch))         # the kernels of one operator launch equally often; variants are summed
    return out


hbm = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    f = os.path.join(root, f"{tag}_bench_{short}_pmc_{counter}.csv")
    for op, (tot, n) in per_launch(f, counter).items():
        hbm[op] = hbm.get(op, 0.0) + tot * 1024.0 / n
# the unit counts the per-launch means belong to: the detail record bench.py wrote beside the FETCH_SIZE run (every step of that process)
import glob
det = {}
for f in glob.glob(os.path.join(root, f"{tag}_bench_{short}_pmc_FETCH_SIZE.detail.json")):
    det = json.load(open(f)).get("all_steps", {})
out = {op: {"bytes": int(v), "sdf_points": det.get("mean_sdf_points"), "M": det.get("mean_M"), "I": det.get("mean_I"),
            "source": f"profiles/{tag}_bench_{short}_pmc_FETCH_SIZE.csv + _WRITE_SIZE.csv"} for op, v in hbm.items()}
out["_note"] = (f"{tag}: FETCH_SIZE + WRITE_SIZE per operator launch in bytes (KiB counters x1024, mean over the launches of the run; FETCH "
                "uncorrected for the gfx950 x2 under-report of wide reads). Collected with bench.py --no-overlap so that one kernel "
                "runs at a time; hashgrid_bwd = vmax + count + plan + emit + apply of the binned scatter.")
path = os.path.join(root, "hbm_traffic.json")
allw = json.load(open(path)) if os.path.exists(path) else {}
allw[workload] = out
json.dump(allw, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))

sq = os.path.join(root, f"{tag}_bench_{short}_pmc_SQ_INSTS_VALU+SQ_ACTIVE_INST_VALU.csv")
if os.path.exists(sq):
    v = {op: int(t / n) for op, (t, n) in per_launch(sq, "SQ_INSTS_VALU").items() if op.startswith("rasterize")}
    a = {op: int(t / n) for op, (t, n) in per_launch(sq, "SQ_ACTIVE_INST_VALU").items() if op.startswith("rasterize")}
    vpath = os.path.join(root, "valu_insts.json")
    allv = json.load(open(vpath)) if os.path.exists(vpath) else {}
    allv[workload] = v
    allv[workload + "_active_cycles_div4"] = a
    allv["_note"] = (f"{tag}: wave64 VALU instructions per launch (rocprofv3 --pmc SQ_INSTS_VALU, mean over the launches, bench.py --no-overlap); "
                     "issue peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction = 614 G/s")
    json.dump(allv, open(vpath, "w"), indent=1)
    print(json.dumps(v, indent=1))
