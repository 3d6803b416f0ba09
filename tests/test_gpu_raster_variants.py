"""The compositing kernels exist in two list shapes: quadrant lists (a wave's 64 pixels follow one list) and row lists (every
16-lane row its own).  The default (round 5) is row lists in both passes; GSDF_RASTER_ROW_LISTS = 0 selects quadrant lists in both, 2 the
round-4 default (quadrant forward, row-list backward); read once per process.  Every combination takes its skip decisions from the same
4x4 reach mask, so forward and backward drop a pair in the same pixels.  Each combination must pass the whole decision-matched parity contract — which
includes that the instrumented forward is bit-identical to the product forward and that last_ids / median_ids equal the oracle's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", ["0", "2"])
def test_compositing_parity_holds_for_the_other_list_shapes(variant):
    env = dict(os.environ, GSDF_RASTER_ROW_LISTS=variant)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_splat_parity.py"), "-m", "gpu", "-q", "-x", "-k",
                        "rasterize_fwd_bwd or pathological or long_tile_lists"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
