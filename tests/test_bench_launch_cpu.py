"""`python bench.py --gpus N` must start N ranks by itself (the driver's scaling run uses that form too when it does not wrap
it in torch.distributed.run): rehearsed here on the CPU with --dry-launch (gloo, no GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bare_gpus_2_starts_two_ranks():
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["config"]["parallelism"] == "dp2"


def test_under_torchrun_no_second_launch():
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29611", "bench.py", "--gpus", "2", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2


def test_world_mismatch_is_an_error():
    r = _run([sys.executable, "bench.py", "--gpus", "1", "--dry-launch"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
