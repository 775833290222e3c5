"""bench.py as the driver starts it: `python bench.py --gpus N ...` with no launcher in front.

For N > 1 the parent must start the N ranks itself (before anything touches a GPU), relay rank 0's JSON line and the
children's exit code.  The CPU test checks the mechanics on a box without a GPU (every rank then ends with the
"no GPU" code, which must come back through the parent); the GPU test is the one-device rehearsal of a two-rank run
(both ranks on device 0, gloo in RCCL's place -- RCCL refuses two ranks on one device): one JSON line, n_gpus = 2 and
the world size the process group itself reported."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_spawns_its_own_ranks_and_relays_their_exit_code():
    import torch
    if torch.cuda.is_available():
        pytest.skip("the no-GPU exit path needs a box without a GPU")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], timeout=300)
    # the children ran (the first to give up says why; the parent then stops the other by its PID), none printed a JSON line,
    # and their exit code (3: no GPU, no CPU fallback) came back
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert 1 <= r.stderr.count("no GPU visible") <= 2 and "stopping the other ranks" in r.stderr, r.stderr[-2000:]
    assert r.stdout.strip() == ""


def test_bench_with_a_launcher_environment_does_not_spawn():
    """WORLD_SIZE set (torch.distributed.run's case): the process is a rank, not a parent -- a mismatch with --gpus is the
    usage error it always was."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"},
             timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_without_a_launcher():
    r = _run(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
             env_extra={"CCV_BENCH_DEVICE": "0", "CCV_BENCH_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5
    detail = out["config"]["exchange_detail"]
    assert detail["world_seen"] == 2 and detail["backend_seen"] == "gloo"
    assert out["config"]["samples_per_gpu"] == 65536 and out["value"] > 0
    assert out["scaling"] == "weak" and out["config"]["primed_iterations"] >= 1024
