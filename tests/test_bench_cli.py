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


def supervised(args, env, world=2, timeout=120):
    """start `world` rank processes of bench.py the way torch.distributed.run does (RANK / WORLD_SIZE / MASTER_* in the
    environment); each is a supervisor whose worker is tests/fake_bench_worker.py"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    fake = os.path.join(ROOT, "tests", "fake_bench_worker.py")
    procs = []
    for r in range(world):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.update(env)
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", str(world), "--share-gpu", "--worker-script", fake] + args,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e))
    outs = [p.communicate(timeout=timeout) for p in procs]
    return [(p.returncode, o, e) for p, (o, e) in zip(procs, outs)]


def test_a_rank_that_stalls_before_its_ring_is_up_yields_the_fallback_line_not_a_timeout():
    """rank 1's first worker never brings its ring up (a hang in communicator initialisation); every
    supervisor ends its worker at the ring deadline and starts a fresh one on the peer-mapped transport with the gloo
    control plane; rank 0 prints ONE line that records the fallback; exit code 0"""
    import time
    t0 = time.time()
    res = supervised(["--ring-deadline", "3", "--transport", "rccl"], {"LBM_BENCH_STALL": "1:0"})
    assert time.time() - t0 < 60
    assert [r[0] for r in res] == [0, 0], res
    lines = [ln for ln in res[0][1].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in res[1][1].splitlines() if ln.startswith("{")]
    d = json.loads(lines[0])
    assert d["launcher"]["attempts"] == 2 and d["config"]["transport"] == "ipc" and d["config"]["control_plane"] == "gloo"
    fb = d["launcher"]["transport_fallback"]
    assert len(fb) == 1 and "[1]" in fb[0]["gave_up"] and "ring" in fb[0]["gave_up"]


def test_a_worker_that_fails_while_creating_its_ring_also_falls_back():
    res = supervised(["--ring-deadline", "20"], {"LBM_BENCH_FAKE_CRASH": "0:0"})
    assert [r[0] for r in res] == [0, 0], res
    d = json.loads([ln for ln in res[0][1].splitlines() if ln.startswith("{")][0])
    assert d["launcher"]["attempts"] == 2 and "rank 0 exited with an error at stage 'start'" in d["launcher"]["transport_fallback"][0]["gave_up"]


def test_a_rank_that_stalls_on_every_attempt_ends_the_run_with_its_stderr_and_a_code():
    res = supervised(["--ring-deadline", "2"], {"LBM_BENCH_STALL": "1:*"})
    assert all(r[0] == 1 for r in res), res
    assert "rank(s) [1] had not brought the ring up" in res[0][2] and not [ln for ln in res[0][1].splitlines() if ln.startswith("{")]


def test_a_run_without_a_line_is_given_up_at_the_launch_timeout():
    res = supervised(["--ring-deadline", "30", "--launch-timeout", "3"], {"LBM_BENCH_FAKE_RUN_S": "600"})
    assert all(r[0] == 1 for r in res), res
    assert "no result 3 s after the start" in res[0][2]


def test_an_unsupervised_happy_path_reports_one_attempt():
    res = supervised([], {})
    assert [r[0] for r in res] == [0, 0], res
    d = json.loads([ln for ln in res[0][1].splitlines() if ln.startswith("{")][0])
    assert d["launcher"]["attempts"] == 1 and "transport_fallback" not in d["launcher"]


def test_no_worker_outlives_a_supervisor_that_is_killed(tmp_path):
    """the launcher (or the driver's time limit) ends the rank processes: every worker goes with its supervisor -- SIGTERM is
    handled (the worker is killed first), and a SIGKILLed supervisor's worker gets PR_SET_PDEATHSIG"""
    import signal
    import socket
    import time
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    fake = os.path.join(ROOT, "tests", "fake_bench_worker.py")
    procs = []
    for r in range(2):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 LBM_BENCH_STALL=f"{r}:*", LBM_FAKE_PID_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--share-gpu", "--worker-script", fake, "--ring-deadline", "300"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e))
    t0 = time.time()
    while time.time() - t0 < 60 and not all((tmp_path / f"worker{r}.pid").exists() for r in range(2)):
        time.sleep(0.1)
    pids = [int((tmp_path / f"worker{r}.pid").read_text()) for r in range(2)]
    procs[0].send_signal(signal.SIGTERM)
    procs[1].send_signal(signal.SIGKILL)
    for p in procs:
        p.wait(timeout=30)
    time.sleep(1.0)
    for pid in pids:   # gone, or a zombie waiting for init to reap it
        try:
            state = open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0]
        except OSError:
            state = "gone"
        assert state in ("gone", "Z", "X"), (pid, state)


@pytest.mark.gpu
def test_real_workers_restart_after_a_stalled_ring_on_one_gpu():
    """the same on the GPU with real workers: rank 1's first worker stalls before creating its ring; the second set of
    workers (fresh processes) runs the benchmark; the line records it"""
    r = run(["--gpus", "2", "--share-gpu", "--rows", "512", "--cols", "1024", "--steps", "10", "--warmup", "5", "--ring-deadline", "45"],
            timeout=600, env={"LBM_BENCH_STALL": "1:0"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["launcher"]["attempts"] == 2 and d["check"]["ring_status"] == "ok" and d["check"]["ring_schedules_agree_bitwise"] is True
    assert "had not brought the ring up" in d["launcher"]["transport_fallback"][0]["gave_up"]   # (rank 0 waits for rank 1 in a collective: both are late)


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
