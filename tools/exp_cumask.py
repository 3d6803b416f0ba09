"""Experiment (run on the GPU box): what does the atomic-bound hash-grid scatter do to kernels that run beside it on another
stream, and which CU mask (hipExtStreamCreateWithCUMask) cures it?  This is the measurement behind gs_sdf_amd/streams.py.

For each configuration three "victim" jobs (4 x 256 MB copies, L1 + D-SSIM forward + backward at 1080p, a 2.5 M-key
radix sort) are timed alone and beside one scatter launch of 458 K points:
  * unmasked side stream                       -> the victim finishes when the scatter finishes
  * strided CU mask (every XCD shared)         -> same
  * whole-XCD partition (scatter on the first k XCDs, victims on the rest) -> victims run at full speed
Usage: python tools/exp_cumask.py"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
import gs_sdf_amd.ops as ops          # noqa: E402
import gs_sdf_amd.sdf as sdfm         # noqa: E402

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
ALL = (1 << 256) - 1


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words) == 0
    return torch.cuda.ExternalStream(s.value)


lm = sdfm.LocalMap([0.0, 0.0, 0.0], 2.0, decoder_implementation=1, device=dev, seed=1)
lm.flatten(accumulate_table_grad_in_place=True)
x = torch.rand(458000, 3, device=dev)
big_a, big_b = torch.rand(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
img = torch.rand(1080, 1920, 3, device=dev, requires_grad=True)
tgt = torch.rand(1080, 1920, 3, device=dev)
keys = torch.randint(0, 1 << 40, (2_500_000,), device=dev)


def job_copy():
    for _ in range(4):
        big_b.copy_(big_a)


def job_loss():
    img.grad = None
    ops.l1_dssim_loss(img, tgt, 0.8, 0.2).backward()


def job_sort():
    torch.sort(keys)


JOBS = dict(copy4=job_copy, l1_dssim=job_loss, sort=job_sort)


def run(side, label):
    out = []
    for name, job in JOBS.items():
        job(); torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(); job(); c1.record(); torch.cuda.synchronize()
        alone = c0.elapsed_time(c1)
        best = (1e9, 1e9)
        for _ in range(3):
            with torch.cuda.stream(side):
                f = lm.encoder.forward(x)
                g = torch.ones_like(f)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(side):
                e0.record(); f.backward(g); e1.record()
            c0.record(); job(); c1.record()
            torch.cuda.synchronize()
            best = min(best, (c0.elapsed_time(c1), e0.elapsed_time(e1)))
        out.append(f"{name}: {alone:6.3f} -> {best[0]:6.3f} ms (scatter {best[1]:5.2f})")
    print(f"{label:>28} | " + " | ".join(out), flush=True)


torch.cuda.set_stream(torch.cuda.Stream())          # not the null stream: masked streams are blocking streams
run(torch.cuda.Stream(), "unmasked side stream")
strided = sum(1 << i for i in range(0, 256, 8))
torch.cuda.set_stream(masked_stream(ALL ^ strided))
run(masked_stream(strided), "strided 32 CUs, main rest")
for n_xcd in (1, 2, 4):
    lo = (1 << (32 * n_xcd)) - 1
    torch.cuda.set_stream(masked_stream(ALL ^ lo))
    run(masked_stream(lo), f"scatter on {n_xcd} XCD(s), main rest")
