"""`python bench.py --gpus N` starts its own ranks (child torch.distributed.run job) when no launcher
surrounds it.  CPU only: the launch path is driven with gloo and bench.py's --launch-check, which
joins the group and counts the ranks without touching a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*flags, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, BENCH] + list(flags), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          env=env, timeout=300)


def test_plain_call_launches_two_ranks_and_relays_one_json_line():
    r = _run("--gpus", "2", "--backend", "gloo", "--launch-check")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # the contract: ONE JSON line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2


def test_plain_call_with_too_few_gpus_fails_with_one_line_naming_the_count():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a machine with fewer than 2 GPUs")
    r = _run("--gpus", "2")
    assert r.returncode != 0
    err = [ln for ln in r.stderr.decode().splitlines() if ln.strip()]
    assert len(err) == 1 and "only %d GPU(s) visible" % torch.cuda.device_count() in err[0], err
    assert r.stdout.decode().strip() == ""


def test_single_rank_launch_check_needs_no_launcher():
    r = _run("--launch-check")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert json.loads(r.stdout.decode().strip())["n_ranks_seen"] == 1
