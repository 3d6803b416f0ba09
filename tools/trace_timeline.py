"""Print the device timeline of the LAST training step found in a rocprofv3 --kernel-trace CSV (run on the GPU box):
start offset, duration, queue/stream and short kernel name, so that stream overlap and idle gaps are visible.
Usage: trace_timeline.py <dir> <out.txt> [once-per-step-kernel-substring]"""
import csv, glob, os, sys

d, out = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else "proj_cull_kernel"
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]      # one launch per step
# the last step that used more than one queue (bench.py ends with a one-stream pass of the same step for the dominant kernel's own
# rate: that is not the step the headline times); a one-stream trace falls back to the last complete step
a, b = marks[-3], marks[-2]
multi = lambda k: len({r.get("Queue_Id") for r in rows[marks[k - 1]:marks[k]]}) > 1   # noqa: E731
for k in range(len(marks) - 2, 0, -1):
    if multi(k) and multi(k + 1):     # not the last two-stream step either: its tail is the teardown between the passes
        a, b = marks[k - 1], marks[k]
        break
t0 = int(rows[a]["Start_Timestamp"])
busy = 0
with open(out, "w") as fo:
    last_end = t0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[-70:]
        qs = r.get("Queue_Id", "?") + "/" + r.get("Stream_Id", "?")
        fo.write(f"{s/1e3:9.1f} {(e-s)/1e3:8.1f} gap{(s-(last_end-t0))/1e3:7.1f} q{qs:>6} {name}\n")
        last_end = max(last_end, int(r["End_Timestamp"]))
    fo.write(f"step span {(last_end - t0)/1e3:.1f} us, {b-a} kernels\n")
print(open(out).read()[-12000:])
