"""Stand-in for bench.py's WORKER in the CPU tests of its supervisor (tests/test_bench_cli.py): takes the worker's
arguments, reports the same stages through --status-file, honours the same LBM_BENCH_STALL fault injection and prints a
canned line on rank 0.  It computes nothing and is never used outside those tests (bench.py --worker-script)."""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--status-file")
ap.add_argument("--attempt", type=int, default=0)
ap.add_argument("--transport", default="rccl")
ap.add_argument("--ctl", default="nccl")
ap.add_argument("--gpus", type=int, default=1)
a, _ = ap.parse_known_args()
rank = os.environ.get("RANK", "0")


def stage(w):
    with open(a.status_file, "a") as fh:
        fh.write(w + "\n")


stage("start")
if os.environ.get("LBM_FAKE_PID_DIR"):      # (the tests check that no worker outlives its supervisor)
    with open(os.path.join(os.environ["LBM_FAKE_PID_DIR"], f"worker{rank}.pid"), "w") as fh:
        fh.write(str(os.getpid()))
stall = os.environ.get("LBM_BENCH_STALL", "")
if stall in (f"{rank}:{a.attempt}", f"{rank}:*"):
    time.sleep(3600)
if os.environ.get("LBM_BENCH_FAKE_CRASH", "") == f"{rank}:{a.attempt}":
    print("fake worker: ncclCommInitRank failed", file=sys.stderr)
    sys.exit(7)
time.sleep(0.3)
stage("ring_up")
time.sleep(float(os.environ.get("LBM_BENCH_FAKE_RUN_S", "0.3")))
stage("timed")
if rank == "0":
    print(json.dumps({"metric": "fake", "value": 1.0, "n_gpus": a.gpus, "config": {"transport": a.transport, "control_plane": a.ctl},
                      "check": {"ring_status": "ok"}}), flush=True)
stage("done")
