"""The tools that turn rocprofv3 output into the committed evidence (profiles/): small synthetic traces, no GPU.

`tools/trace_timeline.py` must print a step of the TIMED region — one that used both queues — not a step of the one-stream pass `bench.py`
ends with, nor the last two-stream step (whose tail is the teardown between the passes)."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, steps):
    """steps: list of 'two' / 'one'; every step = proj_cull on queue 1, a long kernel on queue 1, and (two) one kernel on queue 3"""
    rows, t = [], 1000
    for k, kind in enumerate(steps):
        rows.append(dict(Kernel_Name="gsdf::proj_cull_kernel(long)", Start_Timestamp=t, End_Timestamp=t + 10, Queue_Id=1, Stream_Id=0))
        rows.append(dict(Kernel_Name=f"gsdf::raster_step{k}(int)", Start_Timestamp=t + 20, End_Timestamp=t + 500, Queue_Id=1, Stream_Id=0))
        if kind == "two":
            rows.append(dict(Kernel_Name=f"gsdf::hashgrid_step{k}(long)", Start_Timestamp=t + 100, End_Timestamp=t + 900, Queue_Id=3, Stream_Id=2))
        t += 1000
    os.makedirs(os.path.join(path, "run"), exist_ok=True)
    with open(os.path.join(path, "run", "1_kernel_trace.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)


def _run(tmp_path, steps):
    _trace(str(tmp_path), steps)
    out = tmp_path / "timeline.txt"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_timeline.py"), str(tmp_path), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out.read_text()


def test_timeline_is_a_two_stream_step_of_the_timed_region(tmp_path):
    text = _run(tmp_path, ["two"] * 6 + ["one"] * 4)      # the bench: timed two-stream steps, then the one-stream pass
    assert "hashgrid_step4" in text and "raster_step4" in text, text       # the last two-stream step that is not the last one (5)
    assert "step5" not in text and "step9" not in text and "step8" not in text
    assert "q   3/2" in text and "q   1/0" in text


def test_timeline_of_a_one_stream_trace_is_its_last_complete_step(tmp_path):
    text = _run(tmp_path, ["one"] * 5)
    assert "raster_step2" in text and "step3" not in text and "step4" not in text      # marks[-3] .. marks[-2], as before
