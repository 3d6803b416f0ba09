"""HIP stream plumbing for the two-leg training step (trainer.join_grad, bench.py): XCD-partitioned streams.

The hash-grid backward (scatter) kernel is bound by the fp32 atomic units (~21 G cache-line requests/s chip-wide,
profiles/README.md).  While it runs, every memory request that shares an XCD with it queues behind its atomics in that
XCD's L2 / fabric port: a kernel on another stream does not merely slow down, it finishes when the scatter finishes
(measured beside a 1.7 ms scatter: 256 MB copy x4 0.40 -> 1.68 ms, l1_dssim fwd+bwd 0.37 -> 1.71 ms, radix sort 0.31 ->
1.79 ms; a strided CU mask that leaves both streams on every XCD changes nothing).  Workgroups are dealt round-robin over
the XCDs of a queue's CU mask, so the cure is to give the scatter kernel whole XCDs of its own and keep the other streams
off them (hipExtStreamCreateWithCUMask):

    scatter on XCD 0-1, the rest on XCD 2-7:  scatter 1.7 -> 2.3 ms, copy 0.42 -> 0.55, l1_dssim 0.38 -> 0.43, sort 0.35 -> 0.36
    (scatter alone: 1 XCD 3.0 ms, 2 XCDs 1.8 ms, 4 XCDs 1.6 ms, 8 XCDs 1.6 ms)

tools/exp_cumask.py is the experiment.
"""
import ctypes as C

import torch

_KEEP = []          # ExternalStream does not own the HIP stream; keep the handles alive for the process lifetime
N_XCD = 8           # MI355X: 8 XCDs x 32 CUs; CU-mask bit = 32 * xcd + cu


def _masked_stream(bits, total, device):
    n_words = (total + 31) // 32
    words = (C.c_uint32 * n_words)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(n_words)])
    hip = C.CDLL("libamdhip64.so")
    handle = C.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(handle), C.c_uint32(n_words), words)
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed with error {rc}")
    st = torch.cuda.ExternalStream(handle.value, device=device)
    _KEEP.append((handle, st))
    # tell the XCD-aware kernels how many XCDs this queue has (include/gsdf_hip.h: gsdf_stream_set_xcds)
    from . import capi
    n_xcds = sum(1 for k in range(N_XCD) if (bits >> (k * (total // N_XCD))) & ((1 << (total // N_XCD)) - 1))
    capi.check(capi.lib().gsdf_stream_set_xcds(C.c_void_p(handle.value), n_xcds), "stream_set_xcds")
    return st


def xcd_partition_streams(scatter_xcds=2, n_compute_streams=2, device=None):
    """-> ([compute streams on the other XCDs], scatter stream on the first `scatter_xcds` XCDs).
    NOTE: these are blocking HIP streams: do not mix them with work on the null stream (it would serialise with them)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    total = torch.cuda.get_device_properties(device).multi_processor_count
    per = total // N_XCD
    if not 1 <= scatter_xcds < N_XCD:
        raise ValueError("scatter_xcds must be in [1, 7]")
    lo = (1 << (per * scatter_xcds)) - 1
    hi = ((1 << total) - 1) ^ lo
    return [_masked_stream(hi, total, device) for _ in range(n_compute_streams)], _masked_stream(lo, total, device)


def destroy_all():
    """Synchronises the device and destroys every stream created here.  Call it after the last use (and after
    torch.cuda.set_stream() has been pointed back at a torch-owned stream): a CU-masked stream that is still alive when
    the process runs its exit handlers crashes rocprofv3's finalizer."""
    torch.cuda.synchronize()
    hip = C.CDLL("libamdhip64.so")
    from . import capi
    while _KEEP:
        handle, _ = _KEEP.pop()
        capi.lib().gsdf_stream_set_xcds(C.c_void_p(handle.value), 0)
        hip.hipStreamDestroy(handle)


def launch_on(stream, fn):
    """Runs `fn()` (kernel launches) on `stream`, ordered after the current stream's work so far; the current stream then
    waits for it.  stream=None: plain call."""
    if stream is None:
        return fn()
    cur = torch.cuda.current_stream()
    stream.wait_stream(cur)
    with torch.cuda.stream(stream):
        r = fn()
    cur.wait_stream(stream)
    return r
