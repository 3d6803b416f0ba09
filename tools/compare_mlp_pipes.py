"""Evidence for the decoder's arithmetic (run on the GPU box): ONE joint iteration's flat gradients (bench.py --dump-grads: no
optimizer update) with the decoder on the bf16 MFMA pipe (exact 3-term operand splits, default) twice, and on the fp32 MFMA pipe
(GSDF_MLP_MFMA=f32) once.  Two default runs differ only by the order of fp32 atomics; the split pipe is indistinguishable from the
fp32 pipe if the second distance is of the same size.  (Parameters after tens of Adam steps are NOT a usable metric: two identical
runs drift 3-4 % apart in the SDF parameters, Adam amplifies sign noise of near-zero gradients.)
Usage: python tools/compare_mlp_pipes.py"""
import os, subprocess, sys, tempfile
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
with tempfile.TemporaryDirectory() as d:
    for name, pipe in (("split_a", "bf16x3"), ("split_b", "bf16x3"), ("fp32", "f32")):
        path = os.path.join(d, name + ".pt")
        subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dump-grads", path, "--no-overlap"],
                       env=dict(os.environ, GSDF_MLP_MFMA=pipe), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out[name] = torch.load(path)


def dist(a, b):
    res = {}
    n_dec = 32 * 64 + 64 * 64 * 2 + 64 * 2          # the bench's 4-layer bias-free decoder follows the table in the flat group
    sa, sb = a["sdf"][0], b["sdf"][0]
    big, small = (lambda t: t[:t.numel() - n_dec]), (lambda t: t[t.numel() - n_dec:])
    for k, (x, y) in {"splat": (a["splat"], b["splat"]), "sdf table": (big(sa), big(sb)), "sdf decoder": (small(sa), small(sb))}.items():
        x, y = x.double(), y.double()
        e = (x - y).abs() / torch.clamp(y.abs(), min=float(y.abs().mean()))
        res[k] = {"rel_l2": float((x - y).norm() / y.norm()), "max scaled": float(e.max()), "frac > 1e-4": float((e > 1e-4).double().mean())}
    return res


print("one joint iteration (cfg3, one stream), flat gradients:")
for name, r in (("split pipe vs split pipe (run to run)", dist(out["split_b"], out["split_a"])), ("fp32 pipe  vs split pipe", dist(out["fp32"], out["split_a"]))):
    print(" ", name)
    for k, v in r.items():
        print("     %-12s rel-L2 %.2e   max scaled %.2e   fraction above 1e-4 %.2e" % (k, v["rel_l2"], v["max scaled"], v["frac > 1e-4"]))
