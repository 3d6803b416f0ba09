"""Experiment (GPU box): does restricting the atomic-bound hash-grid backward to a CU subset (hipExtStreamCreateWithCUMask)
keep its speed while leaving the rest of the chip usable by memory-bound kernels running beside it?"""
import ctypes as C, sys, time
import torch
sys.path.insert(0, ".")
import gs_sdf_amd.sdf as sdfm

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def pattern(n, kind):
    if kind == "first":
        return (1 << n) - 1
    step = 256 // n
    return sum(1 << i for i in range(0, 256, step))


lm = sdfm.LocalMap([0.0, 0.0, 0.0], 2.0, decoder_implementation=1, device=dev, seed=1)
grp = lm.flatten(accumulate_table_grad_in_place=True)
x = torch.rand(458000, 3, device=dev)
big_a, big_b = torch.rand(64 << 20, device=dev), torch.empty(64 << 20, device=dev)   # 256 MB copy = 512 MB traffic


def hg_bwd_job():
    f = lm.encoder.forward(x)
    return f


import gs_sdf_amd.ops as ops
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
        for it in range(3):
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
        out.append(f"{name}: {alone:6.3f} -> {best[0]:6.3f} (bwd {best[1]:5.2f})")
    print(f"{label:>22} | " + " | ".join(out), flush=True)


ALL = (1 << 256) - 1
for n in (32, 64, 128):
    lo = (1 << n) - 1
    torch.cuda.set_stream(masked_stream(ALL ^ lo))
    run(masked_stream(lo), f"side first {n}, main rest")
for n in (32, 64):
    m = pattern(n, "strided")
    torch.cuda.set_stream(masked_stream(ALL ^ m))
    run(masked_stream(m), f"side strided {n}, main rest")
sys.exit(0)


def run_old(side, label):
    main = torch.cuda.current_stream()
    # build graph once per iteration on the side stream; time only the backward kernel with events
    res = []
    for it in range(4):
        with torch.cuda.stream(side):
            f = lm.encoder.forward(x)
            g = torch.ones_like(f)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f.backward(g)
            e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1))
    alone = min(res)
    # concurrent: copy on main while backward on side
    res2 = []
    for it in range(4):
        with torch.cuda.stream(side):
            f = lm.encoder.forward(x)
            g = torch.ones_like(f)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            e0.record(); f.backward(g); e1.record()
        c0.record()
        for _ in range(4):
            big_b.copy_(big_a)
        c1.record()
        torch.cuda.synchronize()
        res2.append((e0.elapsed_time(e1), c0.elapsed_time(c1) / 4))
    print(f"{label:>22}: hashgrid_bwd alone {alone:7.3f} ms | beside copies: bwd {min(r[0] for r in res2):7.3f} ms, 256MB copy {min(r[1] for r in res2):7.3f} ms", flush=True)


c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
big_b.copy_(big_a); torch.cuda.synchronize()
c0.record()
for _ in range(4):
    big_b.copy_(big_a)
c1.record(); torch.cuda.synchronize()
print(f"256MB copy alone: {c0.elapsed_time(c1) / 4:.3f} ms")
run(torch.cuda.Stream(), "unmasked side stream")
for n in (128, 64, 32, 16):
    for kind in ("first", "strided"):
        run(masked_stream(pattern(n, kind)), f"{n} CUs {kind}")
