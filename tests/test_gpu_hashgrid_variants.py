"""GSDF_HASHGRID_RESIDENT=w launches the stencil hash-grid forward as a resident grid (w workgroups per CU that walk the chunks) instead of
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
