"""Host-visible count words (include/gsdf_hip.h: gsdf_host_words_alloc): a count output of the C ABI may point at pinned, device-mapped host
memory that the host polls instead of reading a device scalar back.  Same counts either way, on every operator that hands the host a size."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    import gs_sdf_amd.synth as synth
    dev = torch.device("cuda:0")
    sc = synth.make_scene(20000, 320, 240, sh_degree=0, seed=3)
    vm = synth.make_views(2, seed=4)[1:].to(dev)
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}, vm


def _cull(L, capi, sc, vm, count_ptr):
    N = sc["means"].shape[0]
    dev = sc["means"].device
    radii = torch.empty(N, dtype=torch.int32, device=dev)
    ws = torch.empty(L.gsdf_projection_2dgs_ws_bytes(N, 1), dtype=torch.uint8, device=dev)
    scales = sc["log_scales"].exp().contiguous()
    capi.check(L.gsdf_projection_2dgs_cull(N, 1, capi.f32(sc["means"]), capi.f32(sc["quats"]), capi.f32(scales), capi.f32(vm), capi.f32(sc["K"]), 320, 240,
                                           0.05, 300.0, 0.0, capi.ptr(radii), capi.ptr(ws), count_ptr, capi.stream()), "cull")
    return radii


def test_count_lands_in_the_host_word_without_a_synchronisation(scene):
    import gs_sdf_amd.capi as capi
    L = capi.lib()
    sc, vm = scene
    n_dev = torch.empty(1, dtype=torch.int64, device="cuda:0")
    radii = _cull(L, capi, sc, vm, capi.ptr(n_dev))
    want = int(n_dev.item())
    assert want == int((radii > 0).sum()) and want > 0
    words = capi.HostWords(3)
    for i in (2, 0, 1, 2):                     # any word, re-armed and reused
        words.arm(i)
        assert words._host[i] == capi.HostWords.ARMED
        _cull(L, capi, sc, vm, words.dev(i))
        assert words.wait(i) == want           # (polls; falls back to a stream synchronisation after ~2 ms)


def test_operator_layer_reads_the_same_sizes_either_way(scene, monkeypatch):
    import gs_sdf_amd.ops as ops
    sc, vm = scene
    scales = sc["log_scales"].exp().contiguous()

    def run():
        o = ops.fully_fused_projection_2dgs(sc["means"], sc["quats"], scales, vm, sc["K"], 320, 240, 0.05, 300.0, 0.0)
        t = ops.tile_encode(320, 240, 16, o[3], o[2], o[4], True, 1, o[0], o[1])
        return o, t

    monkeypatch.setenv("GSDF_HOST_COUNTS", "0")
    o0, t0 = run()
    monkeypatch.setenv("GSDF_HOST_COUNTS", "1")
    o1, t1 = run()
    assert o0[0].numel() == o1[0].numel() > 0 and t0[1].numel() == t1[1].numel() > 0
    for a, b in zip(list(o0) + list(t0), list(o1) + list(t1)):
        assert torch.equal(a, b)
