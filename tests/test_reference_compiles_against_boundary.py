"""The drop-in boundary compiled against the REFERENCE'S OWN host sources (CPU test; needs /root/reference, so it is skipped
on the GPU box): `g++ -fsyntax-only` of

    include/neural_net/encoding_map.cpp    (TCNNEncoding construction / forward, :15-26,59)
    include/neural_net/sub_map.cpp         (OctreeAS build / query through kaolin_wisp_cpp, :22-35,76-80)
    include/neural_net/local_map.cpp       (LocalMap ctor, get_sdf, get_gradient, sample, meshing, :16-173,449-516)
    include/neural_gaussian/neural_gaussian.cpp   (rasterization_2dgs_sdf :129-271, NeuralGS, distCUDA2 :314)
    include/neural_mapping/neural_mapping.cpp     (NeuralSLAM: sample :73-104, sdf / gs batch iterations :138-300, gs_train :400-486, checkpoints)

with this repository's headers standing where the un-vendored submodules' headers would be
(gs-sdf_amd/host/{gsplat_cpp,tcnn_binding,kaolin_wisp_cpp,kaolin,spatial.h} + compat/nlohmann) and inert stand-ins
(tests/ref_compile_stubs/; the same set oracle/ref_link/build.py compiles and LINKS the reference's sources with) for what is neither on the path nor in this image: OpenCV, PCL, Eigen, the CUDA runtime header,
and the reference's other un-vendored submodules llog and ply_utils/tinyply.  Every call the reference makes into the
replaced submodules therefore type-checks against the replacement's declarations: argument order, types, return tuples,
member names (params_, get_out_dim, ...).  Nothing from /root/reference is copied; the sources are compiled where they lie."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/include"
FILES = ["neural_net/encoding_map.cpp", "neural_net/sub_map.cpp", "neural_net/local_map.cpp", "neural_gaussian/neural_gaussian.cpp",
         # the trainer itself (NeuralSLAM::train / gs_train / sample, the loop north_star names): built without ENABLE_ROS it needs nothing but
         # libtorch, the drop-in headers and the inert stand-ins; it is type-checked here (linking it needs the data loader: OpenCV / PCL I/O)
         "neural_mapping/neural_mapping.cpp"]


def _flags():
    import torch
    import torch.utils.cpp_extension as ce
    inc = [os.path.join(ROOT, "tests", "ref_compile_stubs"), REF, os.path.join(ROOT, "gs-sdf_amd", "host"),
           os.path.join(ROOT, "gs-sdf_amd", "host", "compat"), os.path.join(ROOT, "include")] + ce.include_paths()
    return ["-std=c++17", "-fsyntax-only", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"] + [f"-I{p}" for p in inc]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_reference_host_sources_typecheck_against_the_dropin_headers():
    flags = _flags()

    def run(f):
        r = subprocess.run(["g++", *flags, os.path.join(REF, f)], capture_output=True, text=True, timeout=900)
        return f, r.returncode, "\n".join(l for l in r.stderr.splitlines() if "error" in l or "fatal" in l)[:3000]
    with ThreadPoolExecutor(4) as ex:
        res = list(ex.map(run, FILES))
    bad = [f"{f}:\n{err}" for f, rc, err in res if rc != 0]
    assert not bad, "reference sources do not compile against the drop-in headers:\n" + "\n".join(bad)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_the_typecheck_really_sees_the_dropin_declarations(tmp_path):
    """Guard against a vacuous pass: the same compile with a deliberately wrong call (one argument too few for
    fully_fused_projection_2dgs) must fail, and the drop-in headers must be the ones included."""
    src = tmp_path / "probe.cpp"
    src.write_text('#include "gsplat_cpp/fully_fused_projection.h"\n#include "tcnn_binding/tcnn_binding.h"\n'
                   'void f(torch::Tensor t) { auto r = fully_fused_projection_2dgs(t, t, t, t, t, 1, 1, 0.1f, 1.f, 0.f, true); }\n')
    r = subprocess.run(["g++", *[x for x in _flags()], "-H", str(src)], capture_output=True, text=True, timeout=900)
    assert r.returncode != 0
    assert os.path.join("gs-sdf_amd", "host", "gsplat_cpp", "fully_fused_projection.h") in r.stderr
