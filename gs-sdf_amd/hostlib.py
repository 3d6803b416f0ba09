"""Loader of the pybind11 test harness around the C++/libtorch operator layer (gs-sdf_amd/host/).
The harness exposes the reference-named C++ operators (`fully_fused_projection_2dgs`, `gsplat_cpp::tile_encode`,
`rasterize_to_pixels_2dgs`, `TCNNEncoding`, `TCNNNetwork`, `distCUDA2`) so the parity tests can drive the exact
functions the reference's host code would link against."""
import importlib.util
import os
import sysconfig

import torch  # noqa: F401  (loads libtorch / libamdhip64 before the extension)

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")


def load():
    path = os.path.join(_LIB, "_gsdf_host" + sysconfig.get_config_var("EXT_SUFFIX"))
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with __graft_entry__.build() (make -C gs-sdf_amd/host)")
    spec = importlib.util.spec_from_file_location("_gsdf_host", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # the library's default is the fork's behaviour (stochastic SDF samples from the projection); the Python tests compare operator outputs
    # with deterministic mirrors, so the harness starts in centre mode and tests of the stochastic path switch it on for their scope
    mod.set_sample_mode(False)
    return mod
