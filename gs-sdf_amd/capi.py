"""ctypes binding of the C ABI in include/gsdf_hip.h (libgsdf_hip.so).

Torch is plumbing here: it owns device memory and the stream; every compute call goes through the
C ABI with raw device pointers.  There is NO fallback: if the HIP library is missing or a call
fails, a RuntimeError is raised.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgsdf_hip.so")
_lib = None

_i64, _i32, _f32, _u64, _vp, _sz = C.c_int64, C.c_int, C.c_float, C.c_uint64, C.c_void_p, C.c_size_t

ABI_VERSION = 10     # include/gsdf_hip.h: GSDF_ABI_VERSION this binding was written against


class RasterInstr(C.Structure):
    """include/gsdf_hip.h: gsdf_raster_instr (instrumented compositing launches: counters or the decision record)."""
    _fields_ = [("counters", C.c_void_p), ("trace_rows", C.c_void_p), ("trace_stride", C.c_int32), ("trace_bits", C.c_void_p)]


_SIGS = {
    "gsdf_abi_version": (C.c_int, []),
    "gsdf_last_error": (C.c_char_p, []),
    "gsdf_deterministic": (C.c_int, [C.c_int]),
    "gsdf_timing_begin": (C.c_int, [C.c_char_p]),
    "gsdf_timing_end": (_sz, [C.c_char_p, _sz]),
    "gsdf_timing_trace": (_sz, [C.c_char_p, _sz]),
    "gsdf_host_words_alloc": (C.c_int, [_i32, _vp, _vp]),
    "gsdf_host_words_free": (C.c_int, [_vp]),
    "gsdf_projection_2dgs_ws_bytes": (_sz, [_i64, _i64]),
    "gsdf_projection_2dgs_cull": (C.c_int, [_i64, _i64] + [_vp] * 5 + [_i32, _i32, _f32, _f32, _f32] + [_vp] * 4),
    "gsdf_projection_2dgs_fill": (C.c_int, [_i64, _i64] + [_vp] * 5 + [_i32, _i32, _u64, _vp, _vp, _i64] + [_vp] * 10),
    "gsdf_projection_2dgs_bwd": (C.c_int, [_i64, _i64, _i64] + [_vp] * 5 + [_i32, _i32, _u64] + [_vp] * 11),
    "gsdf_view_colors_fwd": (C.c_int, [_i64, _i64, _i32] + [_vp] * 7),
    "gsdf_view_colors_bwd": (C.c_int, [_i64, _i64, _i32] + [_vp] * 8 + [_i32, _vp]),
    "gsdf_tile_count_ws_bytes": (_sz, [_i64]),
    "gsdf_tile_count": (C.c_int, [_i64, _i32, _i32, _i32] + [_vp] * 7),
    "gsdf_tile_encode_ws_bytes": (_sz, [_i64, _i64]),
    "gsdf_tile_encode": (C.c_int, [_i64, _i64, _i64, _i32, _i32, _i32] + [_vp] * 10),
    "gsdf_rasterize_2dgs_fwd_ws_bytes": (_sz, [_i64, _i64]),
    "gsdf_rasterize_2dgs_fwd": (C.c_int, [_i64, _i64, _i64, _i32, _i32, _i32] + [_vp] * 20),
    "gsdf_rasterize_2dgs_bwd_ws_bytes": (_sz, [_i64, _i64]),
    "gsdf_rasterize_2dgs_bwd": (C.c_int, [_i64, _i64, _i64, _i32, _i32, _i32] + [_vp] * 28),
    "gsdf_rasterize_2dgs_fwd_instr": (C.c_int, [_i64, _i64, _i64, _i32, _i32, _i32] + [_vp] * 21),
    "gsdf_rasterize_2dgs_bwd_instr": (C.c_int, [_i64, _i64, _i64, _i32, _i32, _i32] + [_vp] * 29),
    "gsdf_nan_rows_accumulate": (C.c_int, [_i64] + [_vp] * 5),
    "gsdf_visible_gather": (C.c_int, [_i64] + [_vp] * 7),
    "gsdf_rows_scatter_add": (C.c_int, [_i64, _i32, _vp, _i32, _vp, _vp, _vp]),
    "gsdf_render_post_fwd": (C.c_int, [_i64, _i32] + [_vp] * 10),
    "gsdf_render_post_bwd": (C.c_int, [_i64, _i32] + [_vp] * 12),
    "gsdf_hashgrid_offsets": (_i64, [_i32, _i32, _i32, _i32, _f32, _vp]),
    "gsdf_hashgrid_fwd": (C.c_int, [_i64, _i32, _i32, _i32, _i32, _f32] + [_vp] * 4),
    "gsdf_hashgrid_fwd_jac": (C.c_int, [_i64, _i32, _i32, _i32, _i32, _f32] + [_vp] * 5),
    "gsdf_hashgrid_fwd_jac_rows": (C.c_int, [_i64, _i64, _i32, _i32, _i32, _i32, _f32] + [_vp] * 5),
    "gsdf_hashgrid_fwd_stencil_points": (C.c_int, [_i64, _vp, _i64, _vp, _vp, _f32, _vp, _f32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "gsdf_hashgrid_fwd_stencil": (C.c_int, [_i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32] + [_vp] * 5),
    "gsdf_hashgrid_fwd_stencil_resident": (C.c_int, [_i32]),
    "gsdf_hashgrid_bwd_jac": (C.c_int, [_i64, _i32, _i32] + [_vp] * 4),
    "gsdf_hashgrid_bwd_jac_scatter": (C.c_int, [_i64, _i32, _i32, _vp, _vp, _f32, _vp, _vp, _vp]),
    "gsdf_hashgrid_bwd": (C.c_int, [_i64, _i32, _i32, _i32, _i32, _f32] + [_vp] * 6),
    "gsdf_hashgrid_bwd_binned_ws_bytes": (_sz, [_i64, _i32, _i32, _i32, _i32, _f32]),
    "gsdf_hashgrid_bwd_binned": (C.c_int, [_i64, _i32, _i32, _i32, _i32, _f32] + [_vp] * 4 + [_sz, _vp]),
    "gsdf_hashgrid_bwd_binned_stencil": (C.c_int, [_i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32] + [_vp] * 4 + [_sz, _vp]),
    "gsdf_hashgrid_bwd_bwd": (C.c_int, [_i64, _i32, _i32, _i32, _i32, _f32] + [_vp] * 8),
    "gsdf_hashgrid_bwd_bwd_bwd": (C.c_int, [_i64, _i32, _i32, _i32, _i32, _f32] + [_vp] * 11),
    "gsdf_mlp_fwd": (C.c_int, [_i64, _i32] + [_vp] * 7),
    "gsdf_mlp_acts_floats": (_sz, [_i64, _i32]),
    "gsdf_mlp_bwd_ws_bytes": (_sz, [_i64, _i32]),
    "gsdf_mlp_bwd_ws_bytes_for": (_sz, [_i64, _i32, _vp, _i32]),
    "gsdf_mlp_bwd_is_one_pass": (_i32, [_i32, _vp]),
    "gsdf_mlp_bwd": (C.c_int, [_i64, _i32] + [_vp] * 11),
    "gsdf_mlp_bwd_weights": (C.c_int, [_i64, _i32, _vp, _i32] + [_vp] * 7),
    "gsdf_mlp_bwd_bwd_ws_bytes": (_sz, [_i64, _i32]),
    "gsdf_mlp_bwd_bwd": (C.c_int, [_i64, _i32] + [_vp] * 10),
    "gsdf_hashgrid_bwd_binned2": (C.c_int, [_i64, _i32, _i32, _i32, _i32, _f32] + [_vp] * 6 + [_sz, _vp]),
    "gsdf_sdf_data_term_grad": (C.c_int, [_i64, _i64, _vp, _i32, _vp, _vp, _vp] + [_f32] * 3 + [_vp] * 2),
    "gsdf_sdf_analytic_loss": (C.c_int, [_i64, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp] + [_f32] * 7 + [_vp] * 5),
    "gsdf_l1_dssim_fwd": (C.c_int, [_i32, _i32] + [_vp] * 6),
    "gsdf_l1_dssim_bwd": (C.c_int, [_i32, _i32] + [_vp] * 5 + [_f32, _f32, _vp, _vp]),
    "gsdf_normal_consistency_fwd": (C.c_int, [_i32, _i32] + [_vp] * 7),
    "gsdf_normal_consistency_bwd": (C.c_int, [_i32, _i32] + [_vp] * 9),
    "gsdf_normal_consistency_fwd_bwd": (C.c_int, [_i32, _i32] + [_vp] * 10),
    "gsdf_sdf_query_points": (C.c_int, [_i64, _i32, _vp, _f32, _vp, _f32, _vp, _vp]),
    "gsdf_sdf_query_points2": (C.c_int, [_i64, _vp, _i64, _vp, _vp, _i32, _f32, _vp, _f32, _vp, _vp]),
    "gsdf_gs_sdf_loss": (C.c_int, [_i64, _vp, _i32, _vp, _vp, _f32, _vp, _vp, _vp]),
    "gsdf_gs_sdf_eik_loss": (C.c_int, [_i64, _i32, _vp, _i32, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp]),
    "gsdf_sdf_ray_loss": (C.c_int, [_i64, _i32, _vp, _i32, _vp, _f32, _f32, _f32, _vp, _vp, _vp]),
    "gsdf_occ_bytes": (_sz, [_i32]),
    "gsdf_occ_build": (C.c_int, [_i32, _i64, _vp, _i32, _vp, _vp]),
    "gsdf_occ_query": (C.c_int, [_i32, _i32, _i64, _vp, _vp, _vp, _vp]),
    "gsdf_occ_query_world": (C.c_int, [_i32, _i32, _i64, _vp, _vp, _f32, _vp, _vp, _vp]),
    "gsdf_visible_set_ws_bytes": (_sz, [_i64]),
    "gsdf_visible_set": (C.c_int, [_i32, _i32, _i64, _vp, _vp, _f32, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp]),
    "gsdf_occ_voxel_counts": (C.c_int, [_i32, _vp, _vp, _vp]),
    "gsdf_occ_voxel_list": (C.c_int, [_i32, _vp, _vp, _vp, _vp]),
    "gsdf_occ_raymarch_count": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gsdf_occ_raymarch_fill": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "gsdf_refine_ws_bytes": (C.c_size_t, [_i64]),
    "gsdf_refine_plan": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "gsdf_refine_apply": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsdf_ray_sampler_count": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),                      # (const gsdf_ray_sampler_args *, counts, offsets_incl, total, stream)
    "gsdf_ray_sampler_fill": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsdf_mc_count": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _f32, _vp, _vp, _vp]),
    "gsdf_mc_emit": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsdf_splat_activations_fwd": (C.c_int, [_i64] + [_vp] * 8),
    "gsdf_splat_activations_bwd": (C.c_int, [_i64] + [_vp] * 9),
    "gsdf_isotropic_loss_fwd": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "gsdf_isotropic_loss_bwd": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp]),
    "gsdf_isotropic_loss_fwd_bwd": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsdf_nan_rows": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsdf_densify_stats": (C.c_int, [_i64, _i64, _i32, _i32, _i32] + [_vp] * 9),
    "gsdf_flat_rows_gather": (C.c_int, [_i32, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "gsdf_stream_set_xcds": (C.c_int, [_vp, _i32]),
    "gsdf_adam_step": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _i64, _vp]),
    "gsdf_adam_step_zero_grad": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _i64, _vp]),
    "gsdf_knn_ws_bytes": (_sz, [_i64]),
    "gsdf_knn_mean_dist2": (C.c_int, [_i64] + [_vp] * 4),
}


def lib():
    """Loads libgsdf_hip.so; fails loudly (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  gs-sdf_amd has no CPU or eager fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(l, name)       # AttributeError if the library does not export the symbol
            f.restype, f.argtypes = res, args
        got = l.gsdf_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI version {got}, this binding was written against {ABI_VERSION} (include/gsdf_hip.h): "
                               "stale library, rebuild it with __graft_entry__.build()")
        _lib = l
    return _lib


class HostWords:
    """Counts the host waits for between two launches without a copy kernel and a stream synchronisation (include/gsdf_hip.h:
    gsdf_host_words_alloc; the C++ operator layer's host/src/util.h: HostWords): arm(i), pass dev(i) as the operator's count output, wait(i)."""
    ARMED = -(1 << 63)

    def __init__(self, n=1):
        h, d = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)()
        check(lib().gsdf_host_words_alloc(n, C.byref(h), C.byref(d)), "host_words_alloc")
        self._host, self._dev, self.n = h, C.cast(d, C.c_void_p).value, n

    def dev(self, i=0):
        return C.c_void_p(self._dev + 8 * i)

    def arm(self, i=0):
        self._host[i] = self.ARMED

    def wait(self, i=0):
        h = self._host
        for _ in range(20000):          # ~2 ms of polling, then the queue is drained the ordinary way
            v = h[i]
            if v != self.ARMED:
                return v
        torch.cuda.current_stream().synchronize()
        v = h[i]
        if v == self.ARMED:
            raise RuntimeError("HostWords: the count was never written")
        return v

    def __del__(self):
        try:
            lib().gsdf_host_words_free(self._host)
        except Exception:
            pass


class RaySamplerArgs(C.Structure):
    """include/gsdf_hip.h: gsdf_ray_sampler_args"""
    _fields_ = [("level", C.c_int), ("n_rays", C.c_int64), ("origin", C.c_void_p), ("direction", C.c_void_p), ("depth", C.c_void_p),
                ("end_xyz", C.c_void_p), ("grid", C.c_void_p), ("rand_free", C.c_void_p), ("randn_surf", C.c_void_p),
                ("free_sample_num", C.c_int), ("surface_sample_num", C.c_int), ("map_origin", C.c_float * 3), ("map_size_inv", C.c_float),
                ("map_half", C.c_float), ("range_lo", C.c_float * 3), ("range_hi", C.c_float * 3), ("sample_std", C.c_float),
                ("truncated_dis", C.c_float)]


_words = {}          # (thread id, device index) -> HostWords: two threads never arm the same word, one word per device


def count_via_host_word(launch, device, upper=None):
    """Runs launch(count_pointer) and returns the count it wrote: through a host-visible word (GSDF_HOST_COUNTS=0: a device scalar + item()).
    `upper`: the largest plausible count (it is about to be used as an allocation size); anything outside [0, upper] raises."""
    if os.environ.get("GSDF_HOST_COUNTS", "1") == "0":
        n = torch.empty(1, dtype=torch.int64, device=device)
        launch(C.c_void_p(n.data_ptr()))
        v = int(n.item())
    else:
        import threading
        dev = torch.device(device)
        key = (threading.get_ident(), dev.index if dev.index is not None else torch.cuda.current_device())
        w = _words.get(key)
        if w is None:
            w = _words[key] = HostWords(1)
        w.arm(0)
        launch(w.dev(0))
        v = int(w.wait(0))
    if v < 0 or (upper is not None and v > upper):
        raise RuntimeError(f"count_via_host_word: implausible count {v} (expected 0..{upper})")
    return v


class deterministic:
    """`with capi.deterministic():` — order-independent accumulation in the kernels this thread launches (include/gsdf_hip.h: gsdf_deterministic)"""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.before = lib().gsdf_deterministic(1 if self.on else 0)
        return self

    def __exit__(self, *exc):
        lib().gsdf_deterministic(self.before)
        return False


def timing_begin(only=None):
    """Per-entry-point device timing of the C ABI (include/gsdf_hip.h: gsdf_timing_begin), whoever calls it (Python mirror or the C++
    operator layer): `only` = iterable of entry-point names or None for all."""
    check(lib().gsdf_timing_begin(None if only is None else ",".join(only).encode()), "timing_begin")


def timing_end():
    """-> {entry point: dict(calls, total_ms, mean_ms, min_ms, max_ms, median_ms)}; stops the timing."""
    buf = C.create_string_buffer(1 << 16)      # one line per entry point: ~100 lines of < 100 bytes at most
    lib().gsdf_timing_end(buf, len(buf))
    return _parse_timing(buf.value.decode())


def timing_trace():
    """-> [(entry point, begin_ms, end_ms)] in call order, relative to the first call's begin event; stops the timing."""
    buf = C.create_string_buffer(1 << 22)      # (the call hands the collection over once: one buffer large enough for ~50 k calls)
    lib().gsdf_timing_trace(buf, len(buf))
    out = []
    for ln in buf.value.decode().splitlines():
        name, a, b = ln.split()
        out.append((name, float(a), float(b)))
    return out


def _parse_timing(txt):
    out = {}
    for line in txt.splitlines():
        name, calls, total, mn, mx, med = line.split()
        out[name] = dict(calls=int(calls), total_ms=float(total), mean_ms=float(total) / max(1, int(calls)), min_ms=float(mn), max_ms=float(mx),
                         median_ms=float(med))
    return out


def exported_symbols():
    return sorted(_SIGS)


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed ({code}): {lib().gsdf_last_error().decode()}")


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None, name="tensor"):
    """Raw device pointer of a contiguous CUDA(HIP) tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def f32(t, name="tensor"):
    return ptr(t, torch.float32, name)
