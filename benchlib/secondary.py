"""Secondary lines of bench.py (never the headline): each is printed as `[secondary] <name> {json}` BEFORE the final compact line and kept in full in
gpurun_out/bench_detail.json.  A failing extra line must never take the headline down: every one is wrapped."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _sub(args, extra, steps, workload=None):
    """this same script in a subprocess (nothing of the caller's allocator / stream state leaks into it) -> its detail record"""
    name = f"bench_detail_secondary_{os.getpid()}.json"
    cmd = [sys.executable, BENCH, "--gpus", "1", "--steps", str(steps), "--warmup", str(args.warmup), "--workload", workload or args.workload,
           "--step-terms", args.step_terms, "--splat-order", args.splat_order, "--no-secondary", "--no-cpu-baseline"] + (["--no-overlap"] if args.no_overlap else []) + extra
    env = dict(os.environ, GSDF_BENCH_DETAIL=name)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    path = os.path.join(ROOT, "gpurun_out", name)
    try:
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-300:])
        return json.load(open(path))
    finally:
        if os.path.exists(path):
            os.remove(path)


def _short(j, **more):
    out = {"value": j["value"], "unit": "iters/s", "ms_per_step": j["ms_per_step"], "steps": j["steps"]}
    if "step_ms_hip_events" in j:
        out["step_ms_hip_events"] = {k: v for k, v in j["step_ms_hip_events"].items() if k != "what"}
    out.update(more)
    return out


def run_all(args, impl, analytic, sc, views, K, ug6, target, N, W, H, deg, dev):
    from benchlib.steps import reference_loop, refine_amortised
    out = {}
    steps = min(args.steps, 40)
    other_cfg = "tcnn" if analytic else "default"
    other_mode = "stochastic" if args.sample_mode == "center" else "center"
    other_impl = "python" if impl == "cpp" else "cpp"
    jobs = (
        # the OTHER SDF configuration (decoder_implementation 1 + numerical gradient), same step otherwise
        ("other_sdf_config", lambda: (lambda j: _short(j, sdf_config=other_cfg, roofline={k: j["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms")}))(
            _sub(args, ["--sdf-config", other_cfg, "--sample-mode", args.sample_mode, "--step-impl", impl], steps))),
        # the joint iteration as neural_mapping_node runs it linked against the drop-in with ZERO source edits
        ("reference_loop_zero_edits", lambda: reference_loop(args, sc, views, K, ug6, target, N, W, H, deg, dev)),
        # the other host implementation of the same step
        ("python_mirror_step" if other_impl == "python" else "cpp_joint_iteration", lambda: (lambda j: _short(j, step_impl=j["config"]["step_impl_short"]))(
            _sub(args, ["--sdf-config", args.sdf_config, "--sample-mode", args.sample_mode, "--step-impl", other_impl], steps))),
        # the other SDF-sample mode (the reference's default draws one stochastic point per visible splat)
        ("other_sample_mode", lambda: (lambda j: _short(j, sample_mode=other_mode))(
            _sub(args, ["--sdf-config", args.sdf_config, "--sample-mode", other_mode, "--step-impl", impl], steps))),
        # refinement inside a measured run (grow / split / prune / Adam-state surgery every refine_every steps)
        ("refine_amortised", lambda: refine_amortised(args, sc, views, K, target, N, W, H, deg, dev)),
    )
    # the other BASELINE.json configurations and one stress workload on THIS tree (parity-test shapes; never the headline): each line carries its
    # own dominant kernel's roofline fraction and the list sizes
    def other(workload, extra):
        j = _sub(args, extra + ["--sdf-config", args.sdf_config, "--sample-mode", args.sample_mode], min(args.steps, 20), workload=workload)
        r, c = j["roofline"], j["config"]
        return _short(j, workload=c["workload"], step_impl=c["step_impl_short"], M=c["M"], I=c["I"], L=c["L"], sdf_points_per_step=c["sdf_points_per_step"],
                      roofline={"kernel": r["kernel"], "bound": r["bound"], "frac": round(r["frac"], 4), "avg_launch_ms": round(r["avg_launch_ms"], 4)},
                      ms_per_step_by_kernel=(r.get("in_step") or r)["ms_per_step_by_kernel"])
    if args.workload == "cfg3_1M_1080p":
        jobs = jobs + (
            ("configs1_replica_300k_splat_only", lambda: other("cfg1_replica_300k", ["--no-sdf"])),       # BASELINE.json configs[1]
            ("configs2_replica_300k_joint", lambda: other("cfg1_replica_300k", ["--step-impl", impl])),    # configs[2]
            ("configs4_shape_3M_640x512_K16", lambda: other("cfg4_3M_640x512_K16", ["--step-impl", impl])),  # configs[4]'s shape on one GPU
            ("stress_1M_1080p_sigma2_12", lambda: other("stress_1M_1080p_sigma2_12", ["--step-impl", impl])),
        )
    for name, fn in jobs:
        try:
            out[name] = fn()
        except Exception as e:   # noqa: BLE001
            out[name] = {"error": repr(e)[:300]}
        short = {k: v for k, v in out[name].items() if k not in ("events", "config", "metric")} if isinstance(out[name], dict) else out[name]
        print(f"[secondary] {name} " + json.dumps(short), flush=True)      # the full record is in the detail file
    return out
