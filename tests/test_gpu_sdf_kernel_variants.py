"""Environment-selected variants of the SDF path's kernels (read once per process, hence the subprocesses).

GSDF_MLP_BWD_RANGES=0 runs the decoder's one-pass backward as ONE launch over all layers (rounds 3-4; 124-229 spilled registers on the 5-layer
net) instead of two launches over the layer ranges {top, top-1} and {the rest} with the chain's state handed over as a register image
(round 5, the default: no scratch).  Same tiles, same order of accumulation: the whole decoder parity suite must hold either way.

GSDF_HASHGRID_RESIDENT=w launches the stencil hash-grid forward as a resident grid (w workgroups per CU that walk the chunks) instead of
one workgroup per chunk; read once per process, off by default (measured, the step does not move).  The walk changes which
workgroup computes a chunk, never a chunk's arithmetic: features and Jacobians stay bit-identical to the row-major kernels."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("resident", ["1", "2"])
def test_resident_grid_is_bit_identical(resident):
    env = dict(os.environ, GSDF_HASHGRID_RESIDENT=resident)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_sdf_parity.py"), "-m", "gpu", "-q", "-x", "-k",
                        "stencil_forward_is_bit_identical or sdf_leg_at_the_joint_iteration_size"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_decoder_backward_in_one_launch_still_passes():
    env = dict(os.environ, GSDF_MLP_BWD_RANGES="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_sdf_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_sdf_default_config.py"), "-m", "gpu", "-q", "-x", "-k",
                        "mlp or eikonal or joint_iteration_size or default or double"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
