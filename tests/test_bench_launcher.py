"""`python bench.py --gpus N` starts N ranks by itself (the driver's torch.distributed.run launch stays as it
is).  Driven here without a GPU through --stub: the ranks rendezvous over gloo, rank 0 prints the line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None, timeout=180):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_launches_two_ranks():
    r = _run("--gpus", "2", "--stub", "--envs", "96", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                      # rank 0's line only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["envs_total"] == 192 and out["steps"] == 3 and out["scaling"] == "weak"


def test_gpus_1_is_one_process():
    r = _run("--gpus", "1", "--stub", "--envs", "64")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    """torchrun started 2 ranks but the command line says --gpus 4: refuse instead of printing a mislabelled line."""
    r = _run("--gpus", "4", "--stub", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29512"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_a_failing_rank_fails_the_launch():
    r = _run("--gpus", "2", "--stub", "--envs", "0x")   # argparse error in every rank
    assert r.returncode != 0
