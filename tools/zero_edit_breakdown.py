"""profiles/rNN_zero_edit_loop_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --reference-loop`, reduced by summarize_rocprof.py)
-> profiles/rNN_zero_edit_loop_breakdown.json: where the step of the UNCHANGED caller goes, by who owns the kernel.
Usage: zero_edit_breakdown.py r05 [steps in the profiled run = warm-up + timed, default 25]"""
import collections, csv, json, os, sys

tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
rows = list(csv.DictReader(open(os.path.join(root, f"{tag}_zero_edit_loop_kernel_stats.csv"))))


def owner(n):
    if "gsdf::" in n:
        return "drop-in HIP kernels (gsdf::): the replaced submodules"
    if any(k in n for k in ("naive_conv", "miopen", "Im2d2Col", "Col2Im")) or n.startswith("Cijk_") or "conv" in n.lower():
        return "MIOpen / CK / Tensile convolutions: the reference's in-tree loss_utils::ssim (5 depthwise 11x11 conv2d + their backward) through libtorch"
    if "multi_tensor_apply" in n:
        return "torch::optim::Adam"
    if "rocclr" in n:
        return "runtime copies / fills"
    return "libtorch elementwise / reduce / index kernels: the reference's in-tree losses, activations, update_state, get_gradient glue"


tot, calls, tuning = collections.Counter(), collections.Counter(), 0.0
for r in rows:
    t, k = float(r["TotalDurationNs"]), int(r["Calls"])
    if k <= 8 and float(r["AverageNs"]) > 5e7:      # MIOpen's find step during the first warm-up iteration
        tuning += t
        continue
    tot[owner(r["Name"])] += t
    calls[owner(r["Name"])] += k
total = sum(tot.values())
out = {"steps_in_profile": steps, "kernel_ms_per_step": round(total / steps / 1e6, 3),
       "by_owner": [{"owner": k, "ms_per_step": round(v / steps / 1e6, 3), "share": round(v / total, 3), "launches_per_step": round(calls[k] / steps, 1)}
                    for k, v in tot.most_common()],
       "excluded": {"miopen_find_during_warmup_ms": round(tuning / 1e6, 1)},
       "largest_dropin_kernels": [{"kernel": r["Name"][:110], "ms_per_step": round(float(r["TotalDurationNs"]) / steps / 1e6, 3),
                                   "launches_per_step": round(int(r["Calls"]) / steps, 1)}
                                  for r in sorted((r for r in rows if "gsdf::" in r["Name"]), key=lambda r: -float(r["TotalDurationNs"]))[:10]]}
json.dump(out, open(os.path.join(root, f"{tag}_zero_edit_loop_breakdown.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
