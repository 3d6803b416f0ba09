"""local_map_checkpoint.pt (gs_sdf_amd/checkpoint.py) against libtorch's own torch::save / torch::load (CPU test): a C++
program that registers its parameters the way the reference's LocalMap does writes an archive that load_local_map_checkpoint
reads, and reads the archive save_local_map_checkpoint writes — both decoder implementations."""
import os
import subprocess
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TABLE = 4096


@pytest.fixture(scope="module")
def helper(tmp_path_factory):
    import torch.utils.cpp_extension as ce
    out = tmp_path_factory.mktemp("pt") / "pt_roundtrip"
    lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O1", "-std=c++17", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           *[f"-I{p}" for p in ce.include_paths()], os.path.join(ROOT, "tests", "cpp", "pt_roundtrip.cpp"), "-o", str(out),
           f"-L{lib}", f"-Wl,-rpath,{lib}", "-ltorch", "-ltorch_cpu", "-lc10"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out)


def _run(helper, *args):
    r = subprocess.run([helper, *[str(a) for a in args]], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    return {l.split()[0]: (int(l.split()[1]), float(l.split()[2]), float(l.split()[3])) for l in r.stdout.strip().splitlines()}


def _fake_local_map(impl, seed):
    """A CPU stand-in with the attributes checkpoint.py touches (the real LocalMap needs the HIP library and a GPU)."""
    g = torch.Generator().manual_seed(seed)
    enc = types.SimpleNamespace(params_=torch.randn(N_TABLE, generator=g))
    if impl == 0:
        torch.manual_seed(seed)
        mods = [torch.nn.Linear(32, 64), torch.nn.ReLU(True)]
        for _ in range(3):
            mods += [torch.nn.Linear(64, 64), torch.nn.ReLU(True)]
        dec = torch.nn.Sequential(*mods, torch.nn.Linear(64, 2))
    else:
        dims = [32, 64, 64, 64, 2]
        dec = types.SimpleNamespace(dims=dims, params_=torch.randn(sum(i * o for i, o in zip(dims[:-1], dims[1:])), generator=g), biases_=None)
    return types.SimpleNamespace(encoder=enc, decoder=dec, decoder_implementation=impl)


def _sums(lm):
    from gs_sdf_amd.checkpoint import _layers
    d = {"encoder_local_map": lm.encoder.params_.double()}
    if lm.decoder_implementation == 1:
        d["decoder"] = torch.cat([w.reshape(-1) for w, _ in _layers(lm)]).double()
    else:
        for k, (w, b) in enumerate(_layers(lm)):
            d[f"decoder.{2 * k}.weight"], d[f"decoder.{2 * k}.bias"] = w.double(), b.double()
    return {k: (v.numel(), float(v.sum()), float((v * v).sum())) for k, v in d.items()}


@pytest.mark.parametrize("impl", [0, 1])
def test_python_archive_loads_in_libtorch_and_back(helper, tmp_path, impl):
    from gs_sdf_amd.checkpoint import load_local_map_checkpoint, save_local_map_checkpoint
    # Python -> torch::load
    lm = _fake_local_map(impl, 3)
    p1 = tmp_path / "from_python.pt"
    save_local_map_checkpoint(lm, p1)
    got = _run(helper, "load", impl, N_TABLE, p1)
    want = _sums(lm)
    assert set(got) == set(want)
    for k in want:
        assert got[k][0] == want[k][0] and abs(got[k][1] - want[k][1]) < 1e-6 * max(1.0, want[k][2]) and abs(got[k][2] - want[k][2]) < 1e-6 * want[k][2]
    # torch::save -> Python
    p2 = tmp_path / "from_libtorch.pt"
    want2 = _run(helper, "save", impl, N_TABLE, p2)
    lm2 = load_local_map_checkpoint(_fake_local_map(impl, 4), p2)
    got2 = _sums(lm2)
    assert set(got2) == set(want2)
    for k in want2:
        assert got2[k][0] == want2[k][0] and abs(got2[k][1] - want2[k][1]) < 1e-6 * max(1.0, want2[k][2]) and abs(got2[k][2] - want2[k][2]) < 1e-6 * want2[k][2]
    # shape mismatches are refused
    bad = _fake_local_map(impl, 5)
    bad.encoder.params_ = torch.zeros(N_TABLE + 8)
    with pytest.raises(RuntimeError):
        load_local_map_checkpoint(bad, p2)


def test_tcnn_padded_decoder_layout_is_accepted_and_written(tmp_path):
    """Upstream tiny-cuda-nn's FullyFusedMLP pads the output width to 16: the flat "decoder" of such a checkpoint is 16 x 64 in its last
    layer.  load takes the real rows; pad_tcnn_output=True writes that layout."""
    from gs_sdf_amd.checkpoint import _layers, load_local_map_checkpoint, save_local_map_checkpoint
    lm = _fake_local_map(1, 7)
    n_plain = lm.decoder.params_.numel()
    p = tmp_path / "padded.pt"
    save_local_map_checkpoint(lm, p, pad_tcnn_output=True)
    flat = dict(torch.jit.load(str(p)).named_parameters())["decoder"]
    assert flat.numel() == n_plain - 2 * 64 + 16 * 64
    tail = flat[-16 * 64:].view(16, 64)
    assert torch.equal(tail[:2], _layers(lm)[-1][0]) and float(tail[2:].abs().max()) == 0.0
    back = load_local_map_checkpoint(_fake_local_map(1, 8), p)
    assert torch.equal(back.decoder.params_, lm.decoder.params_)
    # a size that is neither layout is refused with both sizes named
    bad = _fake_local_map(1, 9)
    bad.decoder = types.SimpleNamespace(dims=[32, 64, 64, 2], params_=torch.zeros(32 * 64 + 64 * 64 + 128), biases_=None)
    with pytest.raises(RuntimeError, match="padded layout"):
        load_local_map_checkpoint(bad, p)


def test_refused_checkpoint_leaves_the_map_untouched(tmp_path):
    """the decoder is validated BEFORE the encoder is copied — a checkpoint whose decoder does not fit changes nothing."""
    from gs_sdf_amd.checkpoint import load_local_map_checkpoint, save_local_map_checkpoint
    p = tmp_path / "impl1.pt"
    save_local_map_checkpoint(_fake_local_map(1, 3), p)
    lm = _fake_local_map(0, 4)                                # a biased Sequential decoder cannot take the flat tcnn parameter
    dims = [32, 64, 64, 64, 64, 2]
    g = torch.Generator().manual_seed(1)
    lm.decoder = types.SimpleNamespace(dims=dims, params_=torch.randn(sum(i * o for i, o in zip(dims[:-1], dims[1:])), generator=g),
                                       biases_=torch.randn(sum(dims[1:]), generator=g))
    before = (lm.encoder.params_.clone(), lm.decoder.params_.clone(), lm.decoder.biases_.clone())
    with pytest.raises(RuntimeError, match="does not fit"):
        load_local_map_checkpoint(lm, p)
    assert torch.equal(lm.encoder.params_, before[0]) and torch.equal(lm.decoder.params_, before[1]) and torch.equal(lm.decoder.biases_, before[2])
