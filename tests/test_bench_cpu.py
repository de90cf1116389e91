"""bench.py's launch logic, no GPU needed: a plain `python3 bench.py --gpus N` must start its own N ranks (the
reference's workers are threads of one process, SceneManager.h:392-404; a rank per GPU is their counterpart) and a
WORLD_SIZE that disagrees with --gpus must be an error, never a silent one-rank measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def test_plain_run_with_gpus_n_plans_n_ranks():
    r = _run(["--gpus", "8", "--steps", "5", "--warmup", "2", "--print-launch-plan"])
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])["launch_plan"]
    assert plan[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in plan and "--nnodes=1" in plan
    assert plan[plan.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(plan[plan.index("--master-port") + 1]) < 65536
    i = plan.index(BENCH)
    assert plan[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2", "--print-launch-plan"]     # every flag reaches the ranks


def test_one_gpu_run_needs_no_launcher():
    r = _run(["--gpus", "1", "--print-launch-plan"])
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["launch_plan"] is None


def test_world_size_must_match_gpus():
    # --gpus 8 inside a one-rank environment (round 2's hole: it rendered one rank and printed n_gpus 1)
    r = _run(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""
    # a launcher with more ranks than --gpus
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and r.stdout.strip() == ""


def test_self_launch_starts_the_ranks_and_passes_their_status_on():
    """No GPU here, so every rank ends with bench.py's own "no HIP device" message: what is checked is that a plain
    `--gpus 2` run really started two ranks of this file (both report) and that their failure is the parent's."""
    r = _run(["--gpus", "2", "--backend", "gloo", "--one-device", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert r.stdout.strip() == ""                                     # no JSON line from a run that measured nothing
    assert "torch.distributed.run" in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.stderr.count("no HIP device is visible") >= 2
