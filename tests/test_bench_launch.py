"""bench.py's launch contract without a GPU: `--gpus N` must never silently run fewer ranks than it reports."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "GSDF_BENCH_BACKEND")):
    e = {k: v for k, v in os.environ.items() if k not in drop}
    e.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_n_without_enough_devices_refuses_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "--gpus 2 needs 2 visible GPUs" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())          # no JSON line, in particular no n_gpus: 1 line


def test_gpus_flag_must_match_the_launcher():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
