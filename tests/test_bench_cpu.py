"""bench.py's multi-rank launch on CPU: `python bench.py --gpus 2` starts its own two ranks (torch.distributed.run as a
child of a parent that never touches the GPU), they rendezvous on 127.0.0.1 over gloo, and rank 0's single JSON line
comes back through the parent.  The `plumbing` workload runs no kernel and says so in its metric."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_starts_its_own_ranks():
    one = _run("--gpus", "1", "--workload", "plumbing", "--steps", "3", "--warmup", "1")
    two = _run("--gpus", "2", "--workload", "plumbing", "--steps", "3", "--warmup", "1")
    for out, n in ((one, 1), (two, 2)):
        assert out["n_gpus"] == n and out["steps"] == 3 and out["warmup"] == 1
        assert "NOT a benchmark" in out["metric"] and out["scaling"] == "weak"
    assert two["rccl_ranks"] == 2


def test_bench_refuses_a_mismatched_launch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "plumbing"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


@pytest.mark.timeout(900)
def test_bench_under_the_drivers_own_launcher():
    """The form the driver uses for N > 1: torch.distributed.run starts the ranks, bench.py finds RANK / WORLD_SIZE in its
    environment and must not start ranks of its own; rank 0 prints the one JSON line."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--workload", "plumbing", "--steps", "3", "--warmup", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 3
