"""bench.py's launcher: `--gpus N` without WORLD_SIZE starts its own ranks -- it must refuse
cleanly where the GPUs are missing, relay rank 0's line where they are not, and never hang on plumbing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, timeout, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e)


def test_multi_gpu_request_without_gpus_fails_fast_with_a_message():
    """here (no GPU) and on a one-GPU box alike: exit code 2 and one sentence, no rendezvous, no hang"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the GPUs")
    r = run(["--gpus", "2", "--steps", "5", "--warmup", "1"], timeout=120)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "GPU" in r.stderr and "Traceback" not in r.stderr


def test_world_size_mismatch_is_an_error_not_a_hang():
    r = run(["--gpus", "2", "--steps", "5"], timeout=120, env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


@pytest.mark.gpu
def test_self_launched_ranks_share_one_gpu_and_report_one_line():
    """the N-rank run end to end on ONE device: two child ranks on GPU 0 through the peer-mapped transport (gloo
    barriers), the ring's two schedules compared bitwise inside the run, one JSON line from rank 0"""
    r = run(["--gpus", "2", "--share-gpu", "--rows", "512", "--cols", "1024", "--steps", "10", "--warmup", "5"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["check"]["ring_schedules_agree_bitwise"] is True and d["check"]["ring_status"] == "ok"
    assert abs(d["check"]["total_mass"] - d["check"]["expected_mass"]) < 1e-6 * d["check"]["expected_mass"]
    assert "peer-mapped" in d["config"]["transport"] and len(d["ring_phases"]) == 2
