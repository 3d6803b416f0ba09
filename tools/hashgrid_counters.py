"""profiles/rNN_bench_cfg3_pmc_TCP*/TCC*/TA*.csv (tools/collect_profiles.sh + summarize_rocprof.py) -> the L1 / L2 side of the dominant kernel
(hashgrid_fwd_stencil_kernel) as derived figures: L1->L2 read requests per launch and per query point, their mean latency, the number in flight
per CU (Little: requests x latency / kernel cycles / CUs), the L2 hit rate, fabric reads per launch, TA busy.
Two launch shapes: the one-stream run's full grid (<JAC = true, RESIDENT = false, int>) and — round 6 — the RESIDENT grid the two-stream step
launches (<true, true, int>, 3 workgroups per CU), whose counters come from the pmc2s_* passes of tools/collect_profiles.sh (the headline command with
both streams; counter collection runs kernels one at a time, so these are the resident kernel's own figures, not its figures beside the other leg).
Usage: hashgrid_counters.py r06  ->  profiles/r06_hashgrid_fwd_l1_l2_counters.json"""
import csv, glob, json, os, sys

tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
def derive(KERNEL, pmc, stats_csv):
    vals = {}
    for f in glob.glob(os.path.join(root, f"{tag}_bench_cfg3_{pmc}_*.csv")):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") and KERNEL in r["Kernel_Name"]:
                vals[r["Counter_Name"]] = float(r["mean_per_launch"])
    det = {}
    for f in glob.glob(os.path.join(root, f"{tag}_bench_cfg3_{pmc}_TCP*.detail.json")):
        det = json.load(open(f)).get("all_steps", {})
    stats = {}
    if os.path.exists(os.path.join(root, stats_csv)):
        for r in csv.DictReader(open(os.path.join(root, stats_csv))):
            if KERNEL in r["Name"]:
                stats = r
    g = lambda k: vals.get(k)
    out = {"kernel": KERNEL, "counters_per_launch": vals, "query_points_per_launch": det.get("mean_sdf_points"),
           "average_launch_ns_in_" + stats_csv: float(stats["AverageNs"]) if stats else None}
    if g("TCP_TCC_READ_REQ_sum") and g("TCP_TCC_READ_REQ_LATENCY_sum") and g("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs of the chip (34 M for a 1.7 ms launch at ~2.4 GHz): cycles of ONE clock domain = / 8
        req, lat, cyc = g("TCP_TCC_READ_REQ_sum"), g("TCP_TCC_READ_REQ_LATENCY_sum"), g("GRBM_GUI_ACTIVE") / 8.0
        out["kernel_cycles"] = cyc
        out["l1_to_l2_read_requests"] = req
        out["mean_read_latency_cycles"] = lat / req
        out["reads_in_flight_per_cu"] = lat / cyc / 256.0
        out["requests_per_query_point"] = req / det["mean_sdf_points"] if det.get("mean_sdf_points") else None
        out["requests_per_cycle_per_cu"] = req / cyc / 256.0
        out["pending_stall_fraction_of_cu_cycles"] = g("TCP_PENDING_STALL_CYCLES_sum") / (cyc * 256.0) if g("TCP_PENDING_STALL_CYCLES_sum") else None
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum"):
        out["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        out["fabric_read_requests"] = g("TCC_EA0_RDREQ_sum")
    if g("TA_TA_BUSY_sum") and g("GRBM_GUI_ACTIVE"):
        out["ta_busy_fraction"] = g("TA_TA_BUSY_sum") / (g("GRBM_GUI_ACTIVE") / 8.0 * 256.0)
    return out


out = {"one_stream_full_grid": derive("hashgrid_fwd_stencil_kernel<true, false, int, true>", "pmc", f"{tag}_bench_cfg3_serial_kernel_stats.csv"),
       "two_stream_resident_grid": derive("hashgrid_fwd_stencil_kernel<true, true, int, true>", "pmc2s", f"{tag}_bench_cfg3_kernel_stats.csv"),
       "reading": ("the L1s hold ~reads_in_flight_per_cu line requests in flight per CU for the whole launch (the vector L1's miss capacity is 64) at "
                   "mean_read_latency_cycles each: the kernel's rate is requests = in-flight x CUs / latency (Little), not a function of occupancy; half "
                   "of the requests miss the 4 MiB L2 of their XCD (the 14 hashed levels are 4 MiB each) and are served by the Infinity Cache")}
path = os.path.join(root, f"{tag}_hashgrid_fwd_l1_l2_counters.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
