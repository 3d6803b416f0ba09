"""Write tests/golden/reference_intree.npz: outputs of the REFERENCE'S OWN compiled in-tree functions (losses, SSIM window, Adam state
surgery, geometry helpers, ray sampling) on seeded inputs.  Needs the module oracle/ref_link/build.py builds from /root/reference:

    python oracle/ref_link/build.py && python tools/gen_reference_intree_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_link import build, cases  # noqa: E402

m = build.load()
assert m is not None, "oracle/_ref/_gsdf_reference*.so is missing: python oracle/ref_link/build.py"
inp = cases.inputs()
out = cases.evaluate(m, inp)
path = os.path.join(ROOT, "tests", "golden", "reference_intree.npz")
np.savez_compressed(path, **{"in_" + k: v.numpy() for k, v in inp.items()}, **{"out_" + k: v for k, v in out.items()})
print(f"{path}: {len(inp)} inputs, {len(out)} reference outputs, {os.path.getsize(path) / 1024:.0f} KiB")
