"""The driver parses the LAST stdout line of bench.py: it must be one strict-JSON object of at most 4 KB (BENCH_r04.json: a 20 KB line came back
`parsed: null`).  benchlib/report.py: compact_line builds it from the detail record; checked here on a record of the full shape."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _detail():
    op = lambda k: {"bound": "hbm", "achieved": 2265.2649097897684, "peak": 8000.0, "unit": "GB/s", "frac": 0.283158113723721,
                    "algorithmic_bytes": 2604326023, "avg_launch_ms": 1.1496783499999998, "median_launch_ms": 1.12613}
    ops = ["rasterize_2dgs_bwd", "rasterize_2dgs_fwd", "hashgrid_fwd", "hashgrid_bwd", "mlp_fwd", "mlp_bwd", "mlp_bwd_data", "mlp_bwd_bwd"]
    par = {k: {"elements": 6220800, "above_1e-4": 3, "worst_over_1e-4_bar": 12.123456789, "worst_over_tolerance": 0.1512345678, "rel_l2": 1.23456789e-7}
           for k in ("render_colors", "render_depths", "render_alphas", "render_normals", "render_median", "visibilities", "v_densify",
                     "v_means (compositing + projection + SH backward)", "v_quats (compositing + projection backward)", "v_scales", "v_opacities", "v_sh")}
    par["integer_outputs_bit_exact"] = True
    par["note"] = "x" * 1500
    return {"metric": "train iters/sec (splat raster + SDF fwd+bwd), 1M Gaussians @1080p", "value": 199.12345678901234, "unit": "iters/s", "n_gpus": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 5.0212345678, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "step_ms_hip_events": {"p10": 4.9123456, "p50": 5.0123456, "p90": 5.2123456, "max": 5.9123456, "what": "y" * 200},
            "config": {"workload": "cfg3_1M_1080p: 1000000 random Gaussians, 1920x1080, sh_degree 0, 1 view/GPU/step", "M": 999000, "I": 4100000, "L": 502,
                       "sdf_points_per_step": 2137000, "sdf_config": "default", "ray_batch_short": "sampled in the step (a16)",
                       "step_impl_short": "C++ gsdf_extras::JointIteration, 2 streams, direct splat leg", "sample_mode_short": "center",
                       "parallelism": "view-parallel x8: backend nccl (RCCL), world_size 8, one process per GPU, per-family gradient all-reduce on the owning leg's stream"},
            "roofline": dict(op("hashgrid_fwd"), kernel="hashgrid_fwd", kernel_selection="z" * 400, launches_per_step=1.0, traffic=5368205880.7,
                             traffic_note="n" * 300, traffic_over_algorithmic=2.0612345, timing="t" * 300,
                             ms_per_step_by_kernel={k: 1.3702 for k in ops}, others={k: op(k) for k in ops},
                             valu={"rasterize_2dgs_fwd": {"insts_per_launch": 186795449, "frac_of_issue_peak": 0.7412345},
                                   "rasterize_2dgs_bwd": {"insts_per_launch": 461007655, "frac_of_issue_peak": 0.5512345}},
                             step_B_splat_bytes=1030000000, step_hbm_frac=0.02571234),
            "kernel_ms": {f"op{i}": 0.123456 for i in range(40)},
            "secondary": {k: {"value": 44.21234, "unit": "iters/s", "ms_per_step": 22.6, "config": {"what": "w" * 300}} for k in
                          ("other_sdf_config", "reference_loop_zero_edits", "python_mirror_step", "other_sample_mode", "refine_amortised")},
            "cpu_baseline": {"value": 0.026, "unit": "iters/s", "cores": 256, "kind": "port", "sample": "s" * 400, "sample_short": "s" * 150, "parity": par}}


def test_compact_line_is_strict_json_of_at_most_4_kb_and_keeps_the_contract_keys():
    from benchlib.report import compact_line
    s = compact_line(_detail())
    assert "\n" not in s and len(s.encode()) <= 4096
    j = json.loads(s, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))     # NaN / Infinity are not strict JSON
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert "workload" in j["config"] and "model" not in j["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_bytes"):
        assert k in j["roofline"], k
    assert abs(j["roofline"]["frac"] - j["roofline"]["achieved"] / j["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample", "parity"):
        assert k in j["cpu_baseline"], k
    assert j["cpu_baseline"]["parity"]["worst_over_tolerance"] <= 1.0 and j["cpu_baseline"]["parity"]["elements_above_1e-4"] == 36
    assert j["secondary"]["reference_loop_zero_edits"] == 44.21
