"""A/B of the compositing kernels at a bench workload's shape: mean kernel time of gsdf_rasterize_2dgs_fwd / _bwd over REPS launches
(HIP events), for the row-list kernels (default) and the quadrant-list kernels (GSDF_RASTER_ROW_LISTS=0), each in its own process.
python tools/exp_raster_ab.py [workload]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import torch
    sys.path.insert(0, ROOT)
    import gs_sdf_amd.capi as capi, gs_sdf_amd.ops as ops, gs_sdf_amd.synth as synth
    from bench import WORKLOADS
    dev = torch.device("cuda:0")
    N, W, H, deg, replica = WORKLOADS[sys.argv[1]]
    sc = synth.make_scene(N, W, H, sh_degree=deg, seed=0, replica=replica)
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
            ops.rasterize_bwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat, fwd, ug, None)
        torch.cuda.synchronize()
        REPS = 20
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * REPS)]
        for i in range(REPS):
            ev[3 * i].record(); fwd = ops.rasterize_fwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat)
            ev[3 * i + 1].record(); ops.rasterize_bwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat, fwd, ug, None)
            ev[3 * i + 2].record()
        torch.cuda.synchronize()
        f = sorted(ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(REPS)); b = sorted(ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(REPS))
    print(json.dumps(dict(fwd_ms_median=f[REPS // 2], bwd_ms_median=b[REPS // 2], fwd_ms_min=f[0], bwd_ms_min=b[0], M=int(gid.shape[0]), I=int(flat.shape[0]))))
    sys.exit(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3_1M_1080p"
out = {}
VARIANTS = (("row_lists", {}), ("quadrant_lists", {"GSDF_RASTER_ROW_LISTS": "0"}), ("quadrant_forward_row_backward", {"GSDF_RASTER_ROW_LISTS": "2"}), ("row_lists_again", {}))
if os.environ.get("GSDF_EXP_VARIANTS"):
    VARIANTS = tuple((v, dict(kv.split("=") for kv in v.split(",") if kv)) for v in os.environ["GSDF_EXP_VARIANTS"].split(";"))
for name, env in VARIANTS:
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, __file__, wl, "child"], env=e, capture_output=True, text=True)
    out[name] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-2000:]}
    print(name, out[name], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"raster_ab_{wl}.json"), "w"), indent=1)
