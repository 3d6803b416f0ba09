"""Roofline arithmetic and the compact driver line of bench.py."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# C-ABI entry points -> operator names of the roofline / kernel_ms tables (gsdf_timing_begin / _end time whole entry points)
CABI_OPS = {"gsdf_hashgrid_fwd": "hashgrid_fwd", "gsdf_hashgrid_fwd_stencil": "hashgrid_fwd", "gsdf_hashgrid_fwd_jac_rows": "hashgrid_fwd",
            "gsdf_hashgrid_fwd_jac": "hashgrid_fwd", "gsdf_hashgrid_bwd_binned2": "hashgrid_bwd", "gsdf_hashgrid_bwd_binned_stencil": "hashgrid_bwd",
            "gsdf_hashgrid_bwd": "hashgrid_bwd", "gsdf_hashgrid_bwd_jac": "hashgrid_bwd_input", "gsdf_hashgrid_bwd_bwd": "hashgrid_bwd_bwd",
            "gsdf_mlp_fwd": "mlp_fwd", "gsdf_mlp_bwd": "mlp_bwd", "gsdf_mlp_bwd_data": "mlp_bwd_data", "gsdf_mlp_bwd_weights": "mlp_bwd_weights",
            "gsdf_mlp_bwd_bwd": "mlp_bwd_bwd", "gsdf_rasterize_2dgs_fwd": "rasterize_2dgs_fwd", "gsdf_rasterize_2dgs_bwd": "rasterize_2dgs_bwd"}


def cabi_timing_to_ops(rep):
    """gs_sdf_amd.capi.timing_end() report -> (median, mean, calls) per operator name (entry points of one operator merged)."""
    med, mean, calls, tot = {}, {}, {}, {}
    for name, r in rep.items():
        op = CABI_OPS.get(name, name[5:] if name.startswith("gsdf_") else name)
        calls[op] = calls.get(op, 0) + r["calls"]
        tot[op] = tot.get(op, 0.0) + r["total_ms"]
        med[op] = max(med.get(op, 0.0), r["median_ms"])        # merged entry points: the larger launch's median
    for op in calls:
        mean[op] = tot[op] / max(1, calls[op])
    return med, mean, calls



PROFILE_ROUND = "r06"
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
BF16_MFMA_PEAK = 2500.0          # dense bf16 MFMA, TFLOP/s
F32_MFMA_PEAK = 157.3
VALU_ISSUE_PEAK = 614.4e9        # 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction

ROOF_OPS = {"hashgrid_bwd", "hashgrid_fwd", "rasterize_2dgs_fwd", "rasterize_2dgs_bwd", "mlp_fwd", "mlp_bwd", "mlp_bwd_data", "mlp_bwd_weights",
            "mlp_bwd_bwd"}
# kernel-name fragments of the committed rocprofv3 stats -> operator
KERNEL_TO_OP = (("hashgrid_fwd", "hashgrid_fwd"), ("raster_bwd_", "rasterize_2dgs_bwd"), ("raster_fwd_", "rasterize_2dgs_fwd"),
                ("mlp_fwd_split_kernel<32, 512, false>", "mlp_fwd"), ("bin_apply", "hashgrid_bwd"), ("bin_emit", "hashgrid_bwd"))


def algorithmic(avg, N, W, H, deg, analytic, no_sdf, dec_dims):
    """SURVEY.md section 8d bytes / flops per STEP of every operator of the roofline table (fp32), from the step's measured sizes."""
    M, I, n_gs = avg["M"], avg["I"], avg.get("n_gs_sdf", 0.0)
    P, T, Kb = W * H, ((W + 15) // 16) * ((H + 15) // 16), (deg + 1) ** 2
    n_ray = avg.get("n_ray_pts", 32768.0)
    base_pts = 0 if no_sdf else n_ray + n_gs                  # points that carry gradients (ray batch + splat samples)
    sdf_pts = 7 * base_pts                                    # + their 6 central-difference points (forward-only when analytic)
    alg = {"rasterize_2dgs_bwd": 80 * I + 48 * P + 88 * M, "rasterize_2dgs_fwd": 80 * I + 48 * P}
    flops = {}
    if not no_sdf:
        macs = sum(a * b for a, b in zip(dec_dims[:-1], dec_dims[1:]))
        # S1 per query point: fwd 12 + 1024 (16 levels x 8 corners x 8 B) + 128 (+ 384 B of Jacobian per gradient-carrying point);
        # bwd 8 + 128 + 1024 scatter (+ 128 + 12 for the second-order operands of the analytic configuration)
        alg["hashgrid_fwd"] = 1164 * sdf_pts + 384 * base_pts
        alg["hashgrid_bwd"] = (1300 * base_pts) if analytic else (1160 * sdf_pts)
        bwd_pts = base_pts if analytic else sdf_pts
        flops = {"mlp_fwd": 2 * macs * sdf_pts, "mlp_bwd": 2 * 2 * macs * bwd_pts, "mlp_bwd_data": 2 * macs * bwd_pts,
                 "mlp_bwd_weights": 2 * macs * bwd_pts, "mlp_bwd_bwd": 2 * 2 * macs * base_pts}
    b_splat = (80 + 12 * Kb) * N + (364 + 12 * Kb) * M + 204 * I + 96 * P + 4 * T
    return {"alg": alg, "flops": flops, "b_splat": b_splat, "base_pts": base_pts, "sdf_pts": sdf_pts, "n_ray": n_ray, "n_gs": n_gs, "M": M, "I": I,
            "T": T, "P": P}


def committed_ranking(workload, analytic, no_sdf):
    """operator ranked first (by GPU time) in THIS round's committed one-stream rocprofv3 stats of the headline command, or None"""
    import csv
    spath = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_bench_cfg3_serial_kernel_stats.csv")
    if workload != "cfg3_1M_1080p" or not analytic or no_sdf or not os.path.exists(spath):
        return None
    share = {}
    for row in list(csv.reader(open(spath)))[1:]:
        for sub, op in KERNEL_TO_OP:
            if sub in row[0]:
                share[op] = share.get(op, 0.0) + float(row[4])
                break
    if not share:
        return None
    top = max(share, key=lambda k: share[k])
    return {"operator": top, "percent_of_gpu_time": share[top], "file": os.path.relpath(spath, ROOT)}


def roofline(a, calls, kern_mean, kern_med, steps, workload, analytic, no_sdf, split_mlp, elapsed_per_step):
    """-> the detailed roofline object: dominant kernel first, every other roofline operator under `others`."""
    launches = lambda k: max(1.0, calls.get(k, 0) / steps)
    alg = {k: v / launches(k) for k, v in a["alg"].items() if calls.get(k)}
    flops = {k: v / launches(k) for k, v in a["flops"].items() if calls.get(k)}
    # time per step of an operator = MEAN launch x launches per step (launches of an SDF operator differ in size; `alg` is the per-launch mean)
    per_step = {k: kern_mean.get(k, 0.0) * calls.get(k, 0) / steps for k in list(alg) + list(flops)}
    # The dominant kernel is ranked LIVE (largest time per step by the in-bench HIP-event timers of this run); the committed one-stream rocprofv3
    # stats of this round are the cross-check, and a disagreement is said so in the line (a committed CSV must not pick the kernel)
    live_first = max(per_step, key=lambda k: per_step[k])
    rank = committed_ranking(workload, analytic, no_sdf)
    dom, rule = live_first, "largest time per step by the in-bench HIP-event timers of this run"
    if rank is not None:
        rule += (f"; {rank['file']} (rocprofv3 --kernel-trace --stats of this command on one stream) ranks {rank['operator']} first "
                 f"({rank['percent_of_gpu_time']:.1f} % of GPU time): " + ("agrees" if rank["operator"] == dom else "DISAGREES"))

    def roof(k):
        if k in alg:
            ach = alg[k] / (kern_mean[k] * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "algorithmic_bytes": int(alg[k])}
        ach = flops[k] / (kern_mean[k] * 1e-3) / 1e12
        if split_mlp and k in ("mlp_fwd", "mlp_bwd"):
            # csrc/mlp_split.hip: fp32 operands as three exact bf16 terms, six partial products per multiply-add -> the pipe executes 6x the
            # algorithmic flops; priced against its dense bf16 peak
            return {"bound": "mfma", "achieved": 6 * ach, "peak": BF16_MFMA_PEAK, "unit": "TFLOP/s", "frac": 6 * ach / BF16_MFMA_PEAK,
                    "pipe": "bf16 MFMA, fp32-accurate 3-term operand split (6 products per multiply-add)", "algorithmic_flops": int(flops[k]),
                    "fp32_equivalent_tflops": ach}
        return {"bound": "mfma", "achieved": ach, "peak": F32_MFMA_PEAK, "unit": "TFLOP/s", "frac": ach / F32_MFMA_PEAK, "algorithmic_flops": int(flops[k])}

    # HBM-side traffic of the dominant kernel from the committed PMC run of the same command (FETCH_SIZE + WRITE_SIZE, separate passes).  The
    # entry carries the point count of THAT run: bytes scale with the points, so the per-point figure is applied to this run's points
    traffic = traffic_note = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        e = json.load(open(tpath)).get(workload, {}).get(dom)
        if isinstance(e, dict) and e.get("sdf_points") and dom.startswith(("hashgrid", "mlp")):
            traffic = e["bytes"] * a["sdf_pts"] / e["sdf_points"]
            traffic_note = (f"{e['bytes']} B per launch at {e['sdf_points']:.0f} query points in the PMC run ({e.get('source', 'profiles/')}), scaled to this "
                            f"run's {a['sdf_pts']:.0f} points; FETCH_SIZE as reported (64 B per request; gather widths are uncalibrated on gfx950, "
                            "MI355X_MICROARCH.md), Infinity-Cache hits included")
        elif isinstance(e, dict):
            traffic, traffic_note = e["bytes"], e.get("source")
        elif e:
            traffic, traffic_note = e, "per launch in the committed PMC run (point count of that run not recorded)"
    vpath = os.path.join(ROOT, "profiles", "valu_insts.json")
    valu = json.load(open(vpath)).get(workload) if os.path.exists(vpath) else None
    dur = kern_mean.get(dom, float("nan"))
    out = dict(roof(dom), kernel=dom, kernel_selection=rule, avg_launch_ms=dur, median_launch_ms=kern_med.get(dom),
               launches_per_step=calls.get(dom, 0) / steps, traffic=traffic, traffic_note=traffic_note,
               traffic_over_algorithmic=(None if not traffic else traffic / alg[dom] if dom in alg else None),
               timing=("HIP events on the launch stream over the timed steps (gsdf_timing_begin/_end inside the C ABI: one pair around everything an entry "
                       "point launches); mean launch; the two legs share the chip, so a launch's duration includes its neighbours' slowdown — profiles/ holds "
                       "the one-stream rocprofv3 stats"),
               ms_per_step_by_kernel={k: round(v, 4) for k, v in per_step.items()},
               others={k: dict(roof(k), avg_launch_ms=kern_mean[k], median_launch_ms=kern_med.get(k)) for k in per_step if k != dom and kern_med.get(k)},
               # compositing kernels are bound by VALU issue, not HBM: wave64 VALU instructions per launch (profiles/valu_insts.json, rocprofv3 --pmc
               # SQ_INSTS_VALU on one stream) over the launch time measured here, against 1024 SIMDs x 2.4 GHz / 4 cycles
               valu=(None if not valu else {k: {"insts_per_launch": v, "frac_of_issue_peak": v / (kern_mean[k] * 1e-3) / VALU_ISSUE_PEAK}
                                            for k, v in valu.items() if kern_mean.get(k)}),
               step_B_splat_bytes=int(a["b_splat"]), step_hbm_frac=a["b_splat"] / elapsed_per_step / 8e12)
    return out


def _finite(o):
    """strict JSON: NaN / Infinity become null"""
    if isinstance(o, float):
        return o if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def compact_line(d):
    """The ONE line the driver parses: <= 4 KB, strict JSON.  Everything else is in gpurun_out/bench_detail.json (and the earlier stdout lines)."""
    r = d["roofline"]
    c = d["config"]
    line = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": c["workload"], "step_impl": c["step_impl_short"], "sdf_config": c["sdf_config"], "sample_mode": c["sample_mode_short"],
                      "ray_batch": c["ray_batch_short"], "parallelism": c["parallelism"], "sdf_points_per_step": c["sdf_points_per_step"],
                      "M": c["M"], "I": c["I"]}
    line["internal_warmup_steps"] = d.get("internal_warmup_steps")      # untimed steady-state steps bench.py runs on top of --warmup
    if d.get("all_steps") and c.get("sdf_points_per_step"):
        line["sdf_points_timed_vs_run_mean"] = [c["sdf_points_per_step"], round(d["all_steps"].get("mean_sdf_points", 0.0))]
    line["step_ms_hip_events"] = {k: round(v, 3) for k, v in d["step_ms_hip_events"].items() if k != "what"}
    line["roofline"] = {"kernel": r["kernel"], "bound": r["bound"], "achieved": round(r["achieved"], 1), "peak": r["peak"], "unit": r["unit"],
                        "frac": round(r["frac"], 4), "algorithmic_bytes": r.get("algorithmic_bytes"), "avg_launch_ms": round(r["avg_launch_ms"], 4),
                        "launches_per_step": r["launches_per_step"], "traffic": None if r["traffic"] is None else int(r["traffic"]),
                        "traffic_over_algorithmic": None if not r.get("traffic_over_algorithmic") else round(r["traffic_over_algorithmic"], 2),
                        "valu_issue_frac": None if not r.get("valu") else {k.replace("rasterize_2dgs_", ""): round(v["frac_of_issue_peak"], 3) for k, v in r["valu"].items()},
                        "measured": "alone (same step on one stream, HIP events)" if r.get("in_step") else "timed region",
                        "sdf_points_per_step": r.get("sdf_points_per_step"),
                        "in_step": None if not r.get("in_step") else {"avg_launch_ms": round(r["in_step"]["avg_launch_ms"], 4), "frac": round(r["in_step"]["frac"], 4),
                                                                      "sdf_points_per_step": r["in_step"]["sdf_points_per_step"],
                                                                      "what": "timed region, two streams: beside the other leg's kernels"},
                        "step_hbm_frac": round(r["step_hbm_frac"], 4),
                        "ms_per_step_by_kernel": (r["in_step"] if r.get("in_step") else r)["ms_per_step_by_kernel"]}
    cb = d.get("cpu_baseline")
    if cb:
        p = cb.get("parity", {})
        worst, above, n_el = 0.0, 0, 0
        for k, v in p.items():
            if isinstance(v, dict) and "worst_over_tolerance" in v:
                worst = max(worst, v["worst_over_tolerance"]); above += v["above_1e-4"]; n_el += v["elements"]
        line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample_short"],
                                "parity": {"integer_outputs_bit_exact": p.get("integer_outputs_bit_exact"), "worst_over_tolerance": round(worst, 4),
                                           "elements_above_1e-4": above, "elements": n_el, "pinned": "unpinned (no reference kernel source or vector exists); "
                                           "oracle cross-checked by an independent fp32 build, profiles/parity_r06.json"}}
    if d.get("collectives"):
        c_ = d["collectives"]
        line["collectives"] = {"backend": c_["backend"], "world_size": c_["world_size"],
                               "splat_allreduce_ms": None if not c_["splat"]["mean_ms"] else round(c_["splat"]["mean_ms"], 3), "splat_bytes": c_["splat"]["bytes"],
                               "sdf_allreduce_ms": None if not c_["sdf"]["mean_ms"] else round(c_["sdf"]["mean_ms"], 3), "sdf_bytes": c_["sdf"]["bytes"]}
    if d.get("replica_checksums"):
        line["replicas_identical"] = d["replica_checksums"]["all_equal"]
    if d.get("secondary"):
        line["secondary"] = {k: (round(v["value"], 2) if isinstance(v, dict) and "value" in v else "error") for k, v in d["secondary"].items()}
        line["secondary_unit"] = "iters/s (each its own line above and in the detail file)"
    line["detail"] = "gpurun_out/bench_detail.json"
    line = _finite(line)
    s = json.dumps(line)
    if len(s) > 4000:          # never let an unforeseen field push the line past the driver's limit
        line["roofline"].pop("ms_per_step_by_kernel", None)
        line.pop("secondary", None); line.pop("secondary_unit", None)
        s = json.dumps(line)
    assert len(s.encode()) <= 4096, len(s)
    return s
