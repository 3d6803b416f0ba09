"""Multi-GPU rehearsal on ONE device (the driver's 8-GPU run is the only place RCCL sees 8 ranks): the view-parallel step hands every
parameter family's flat gradient buffer to a hook — where bench.py issues the RCCL all-reduce — on the stream that family's optimizer runs on.
Asserted here with HIP-event timestamps: (a) the splat family's hook runs on the caller's stream, the SDF family's on JointIteration's second
stream; (b) each hook sees the COMPLETE gradient of its family (the buffer a later optimizer-free step leaves behind is what the hook saw);
(c) the SDF family's collective of step i executes AFTER the first kernel of step i + 1 has started on the splat leg's stream — it overlaps the next
step's render by construction (the table is read again only when the next SDF leg starts), so its cost is hidden unless it outlasts the render."""
import pytest
import torch

import gs_sdf_amd.synth as synth

pytestmark = pytest.mark.gpu


def test_per_family_hooks_run_on_the_owning_legs_stream_and_overlap_the_next_render():
    import gs_sdf_amd.hostlib as hostlib
    import gs_sdf_amd.sdf as sdfm
    from gs_sdf_amd.trainer import SplatParams
    host = hostlib.load()
    dev = torch.device("cuda:0")
    N, W, H = 300_000, 1200, 680
    sc = synth.make_scene(N, W, H, sh_degree=0, seed=0, replica=True)
    params = SplatParams.from_scene(sc, dev, None)
    lm = sdfm.LocalMap([0.0, 0.0, 5.5], 16.0, bce_sigma=0.02, decoder_implementation=0, device=dev, seed=5)
    enc, dec = host.TCNNEncoding(16, 2, 19, 32, 2.0), host.TCNNNetwork(32, 2, 64, 4, True)
    enc.params_, dec.params_, dec.biases_ = lm.encoder.params_.detach().clone(), lm.decoder.params_.detach().clone(), lm.decoder.biases_.detach().clone()
    fields = [params.views[k].detach().clone() for k in ("offsets", "scaling", "quaternion", "opacity", "features_dc", "features_rest")]
    ji = host.JointIteration(params.anchors, fields, enc, dec, [0.0, 0.0, 5.5], 16.0, 0.02, 8, W, H, 0, True, True, True, True)    # two streams
    views = synth.make_views(8, seed=1).to(dev)
    K, target = sc["K"].to(dev), torch.rand(H, W, 3, device=dev)
    pts = ((torch.rand(32768, 3) - 0.5) * 14.0 + torch.tensor([0.0, 0.0, 5.5])).to(dev)
    sdf = (torch.randn(32768, 1) * 0.02).to(dev)
    main = torch.cuda.current_stream()
    seen = {"splat": [], "sdf": []}

    def hook(family):
        def fn(g):
            s = torch.cuda.current_stream()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            g.mul_(1.0)                           # where the all-reduce goes: a pass over the whole buffer on this stream
            b.record(s)
            seen[family].append(dict(stream=s.cuda_stream, a=a, b=b, numel=g.numel(), absum=float(0)))
        return fn
    ji.set_grad_hooks(hook("splat"), hook("sdf"))
    starts = []
    for i in range(6):
        e = torch.cuda.Event(enable_timing=True)
        e.record(main)
        starts.append(e)
        ji.step(views[i % 8][None], K, target, pts, sdf, [], True, [])
    ji.sync()
    torch.cuda.synchronize()
    assert len(seen["splat"]) == len(seen["sdf"]) == 6
    # (a) streams
    assert all(r["stream"] == main.cuda_stream for r in seen["splat"]), "the splat family's collective must be issued on the caller's stream"
    side = {r["stream"] for r in seen["sdf"]}
    assert len(side) == 1 and main.cuda_stream not in side, "the SDF family's collective must be issued on the second stream"
    assert seen["splat"][0]["numel"] == ji.splat_flat().numel() and seen["sdf"][0]["numel"] == ji.sdf_flat().numel()
    # (c) overlap: the SDF hook of step i finishes after step i + 1 has started on the splat leg's stream (steady state: steps 2..4)
    late = [starts[i + 1].elapsed_time(seen["sdf"][i]["b"]) for i in range(2, 5)]
    assert all(t > 0 for t in late), f"the SDF family's collective does not overlap the next step's render: {late}"
    # ... and before the NEXT step's SDF leg needs the table (it cannot be later than that step's own SDF hook)
    for i in range(2, 5):
        assert seen["sdf"][i]["b"].elapsed_time(seen["sdf"][i + 1]["a"]) > 0
    # the splat family's collective sits between its backward and its Adam on the caller's stream: inside its own step
    for i in range(2, 5):
        assert starts[i].elapsed_time(seen["splat"][i]["a"]) > 0 and seen["splat"][i]["b"].elapsed_time(starts[i + 1]) > -1e-3


def test_hooks_see_the_complete_family_gradient():
    """update=False leaves the gradients in the flat buffers: what the hook was handed equals what is left behind (nothing is added after it)"""
    import gs_sdf_amd.hostlib as hostlib
    import gs_sdf_amd.sdf as sdfm
    from gs_sdf_amd.trainer import SplatParams
    host = hostlib.load()
    dev = torch.device("cuda:0")
    N, W, H = 20_000, 320, 192
    sc = synth.make_scene(N, W, H, sh_degree=0, seed=0)
    params = SplatParams.from_scene(sc, dev, None)
    lm = sdfm.LocalMap([0.0, 0.0, 5.5], 16.0, bce_sigma=0.02, decoder_implementation=0, device=dev, seed=5)
    enc, dec = host.TCNNEncoding(16, 2, 19, 32, 2.0), host.TCNNNetwork(32, 2, 64, 4, True)
    enc.params_, dec.params_, dec.biases_ = lm.encoder.params_.detach().clone(), lm.decoder.params_.detach().clone(), lm.decoder.biases_.detach().clone()
    fields = [params.views[k].detach().clone() for k in ("offsets", "scaling", "quaternion", "opacity", "features_dc", "features_rest")]
    for two in (False, True):
        ji = host.JointIteration(params.anchors, fields, enc, dec, [0.0, 0.0, 5.5], 16.0, 0.02, 8, W, H, 0, two, True, True, True)
        views = synth.make_views(4, seed=1).to(dev)
        pts = ((torch.rand(4096, 3) - 0.5) * 14.0 + torch.tensor([0.0, 0.0, 5.5])).to(dev)
        sdf = (torch.randn(4096, 1) * 0.02).to(dev)
        snap = {}
        ji.set_grad_hooks(lambda g: snap.__setitem__("splat", g.clone()), lambda g: snap.__setitem__("sdf", g.clone()))
        ji.step(views[1][None], sc["K"].to(dev), torch.rand(H, W, 3, device=dev), pts, sdf, [], False, [])
        ji.sync()
        torch.cuda.synchronize()
        assert float(snap["splat"].abs().sum()) > 0 and float(snap["sdf"].abs().sum()) > 0
        assert torch.equal(snap["splat"], ji.splat_flat_grad()) and torch.equal(snap["sdf"], ji.sdf_flat_grad())
