"""bench.py's launch path on CPU: `python bench.py --gpus 2 --dry-launch` must start TWO ranks by itself (no launcher
around it), rendezvous on 127.0.0.1, shard the candidates, time with a barrier on both sides and print ONE line that
says n_gpus = 2; a launcher that starts a different number of ranks than --gpus must be refused."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_spawns_its_own_ranks():
    out = _run(["--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "1", "--sweep-batch", "6", "--dim", "4"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_launch"] is True and rec["backend"] == "gloo"
    assert len(rec["rank_seconds"]) == 2 and rec["results_complete_on_rank0"]


def test_bench_single_rank_dry_launch():
    out = _run(["--gpus", "1", "--dry-launch", "--steps", "1", "--warmup", "0", "--sweep-batch", "3", "--dim", "2"])
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 1


def test_bench_refuses_a_mislabelled_world():
    out = _run(["--gpus", "8", "--dry-launch"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode == 2 and "refusing" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
