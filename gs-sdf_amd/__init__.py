"""gs-sdf_amd: MI355X-native (gfx950) replacement for the GS-SDF hot path.

Scope: the 2D-Gaussian-splat rasteriser (project -> tile-bin ->
alpha-composite, forward and backward) and the hash-grid SDF network, as hand-written HIP
kernels behind a C ABI (include/gsdf_hip.h), plus the host-side mirror of the reference's
operator interface (`gsplat_cpp`, `tcnn_binding`, `simple-knn` headers of
/root/reference/include/neural_gaussian/neural_gaussian.cpp:5-14).
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
