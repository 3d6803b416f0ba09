"""Round 6: the compositing kernels (lane-quad lists) at a bench workload's shape, each variant in its own process.  A variant is the product
library ("quads") or a compile-time variant of it ("quads@<name>", built by tools/build_variants.sh).  profiles/r06_raster_quad_vs_row_lists_cfg3.json
is this tool's record of the quad kernels against round 5's row-list kernels (forward bit-identical), taken before those were removed:
  * kernel time of gsdf_rasterize_2dgs_fwd / _bwd over REPS launches (HIP events; the quad forward's time includes its pack + mask passes),
  * pair counters of the instrumented instantiations (wave iterations, evaluated / blending lanes),
  * the outputs themselves: the parent compares the forward images bit for bit and the gradients element-wise.
python tools/exp_raster_quads.py [workload]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import torch
    sys.path.insert(0, ROOT)
    import gs_sdf_amd.capi as capi
    if os.environ.get("GSDF_EXP_LIB"):      # compile-time variant of the library (tools/build_variants.sh), this tool only
        capi.LIB_PATH = os.path.join(ROOT, "gs-sdf_amd", "lib", "variants", os.environ["GSDF_EXP_LIB"], "libgsdf_hip.so")
    import gs_sdf_amd.ops as ops, gs_sdf_amd.synth as synth
    from bench import WORKLOADS, WORKLOAD_SIGMA_PX
    dev = torch.device("cuda:0")
    N, W, H, deg, replica = WORKLOADS[sys.argv[1]]
    sc = synth.make_scene(N, W, H, sh_degree=deg, seed=0, replica=replica, sigma_px=WORKLOAD_SIGMA_PX.get(sys.argv[1], (0.5, 4.0)))
    vm = synth.make_views(2, seed=1)[1:2].to(dev)
    d = lambda t: t.to(dev)
    with torch.no_grad():
        cam, gid, radii, m2d, dep, rt, nrm, smp, sw = ops.fully_fused_projection_2dgs(d(sc["means"]), d(sc["quats"]), d(sc["log_scales"].exp()), vm, d(sc["K"]), W, H, 0.05, 300.0, 0.0)
        col = ops.get_view_colors(vm, d(sc["means"]), radii, d(sc["sh"]), cam, gid, deg)
        opa = torch.sigmoid(d(sc["logit_opacities"]))[gid].contiguous()
        tpg, flat, offs = ops.tile_encode(W, H, 16, m2d, radii, dep, True, 1, cam, gid)
        ug = {k: v.to(dev) for k, v in synth.upstream_grads(H, W, seed=2).items()}
        for _ in range(3):
            fwd = ops.rasterize_fwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat)
            g = ops.rasterize_bwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat, fwd, ug, None)
        torch.cuda.synchronize()
        REPS = 20
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * REPS)]
        for i in range(REPS):
            ev[3 * i].record(); fwd = ops.rasterize_fwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat)
            ev[3 * i + 1].record(); g = ops.rasterize_bwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat, fwd, ug, None)
            ev[3 * i + 2].record()
        torch.cuda.synchronize()
        f = sorted(ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(REPS)); b = sorted(ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(REPS))
        cnt = torch.zeros(16, dtype=torch.int64, device=dev)
        fc = ops.rasterize_fwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat, counters=cnt)
        ops.rasterize_bwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat, fc, ug, cnt)
        torch.cuda.synchronize()
        c = cnt.cpu().tolist()
        out = {k: v.cpu() for k, v in fwd.items() if v is not None and k not in ("fwd_ws", "trace_bits")}
        out.update({k: v.cpu() for k, v in g.items() if v is not None})
        torch.save(out, sys.argv[3])
    print(json.dumps(dict(fwd_ms_median=f[REPS // 2], bwd_ms_median=b[REPS // 2], fwd_ms_min=f[0], bwd_ms_min=b[0], M=int(gid.shape[0]), I=int(flat.shape[0]),
                          fwd_wave_iterations=c[0], fwd_lanes_live=c[1], fwd_lanes_ok=c[2], fwd_lanes_blend=c[3], fwd_empty_visits=c[7],
                          fwd_useful_of_evaluated=c[3] / max(64 * c[0], 1),
                          bwd_wave_iterations=c[4], bwd_lanes_live=c[5], bwd_lanes_blend=c[6], bwd_useful_of_evaluated=c[6] / max(64 * c[4], 1))))
    sys.exit(0)
import torch
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3_1M_1080p"
out = {}
VARIANTS = (("quads", {}), ("quads_again", {}))
if os.environ.get("GSDF_EXP_VARIANTS"):
    # "quads", "rows", or "quads@<library variant>"
    VARIANTS = tuple((v, ({"GSDF_EXP_LIB": v.split("@")[1]} if "@" in v else {})) for v in os.environ["GSDF_EXP_VARIANTS"].split(";"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for name, env in VARIANTS:
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, __file__, wl, "child", f"/tmp/raster_{name.replace('@', '_')}.pt"], env=e, capture_output=True, text=True)
    out[name] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-3000:]}
    print(name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in out[name].items() if k.endswith("_ms_median") or k == "error"}, flush=True)
names = [n for n, _ in VARIANTS if "error" not in out[n]]
if len(names) >= 2:      # the first two variants' outputs against each other
    a, b = torch.load(f"/tmp/raster_{names[0].replace('@', '_')}.pt"), torch.load(f"/tmp/raster_{names[1].replace('@', '_')}.pt")
    cmp = {}
    for k in a:
        x, y = a[k], b[k]
        if k.startswith("v_"):
            den = y.abs().max().item() + 1e-30
            cmp[k] = dict(max_abs_diff_over_max=float((x - y).abs().max().item() / den), frac_rel_gt_1e4=float(((x - y).abs() > 1e-4 * y.abs() + 1e-7 * den).float().mean().item()))
        else:
            cmp[k] = dict(bit_identical=bool(torch.equal(x, y)), n_diff=int((x != y).sum().item()))
    out[f"{names[0]}_vs_{names[1]}"] = cmp
    print(json.dumps(cmp, indent=1))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"raster_quads_{wl}.json"), "w"), indent=1)
