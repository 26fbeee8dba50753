#!/usr/bin/env python3
"""Headline benchmark: MLUPS and % of the HBM roofline of the fused D2Q9 BGK collide+stream
step on a synthetic periodic box, 8192 x 8192 f64 nodes PER GPU (BASELINE.json configs[1]),
slab-decomposed along rows over N GPUs (weak scaling) with a halo exchange per launch.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N ...          (no WORLD_SIZE in the environment: starts its N ranks itself)

N > 1, either way: every rank process is a SUPERVISOR that never touches the GPU and runs the benchmark in a WORKER
child of its own.  Workers report stages ("ring_up", "timed", "done") through a file; the supervisors agree over gloo:
a rank that has not brought its ring up within --ring-deadline seconds (an RCCL initialisation or first exchange that
hangs) makes every supervisor end its worker (exact pid) and start a FRESH one on the peer-mapped transport with the
gloo control plane -- no RCCL anywhere --, and the line then carries `launcher.transport_fallback`; a run that has
no line after --launch-timeout seconds ends non-zero with the stalled rank's stderr instead of a time-limit kill.

One step = one pass of the hot path over every node of the box (the default launch fuses 5
steps: `steps` counts time steps, not launches).  Rank 0 prints ONE JSON line.

Timing protocol: the GPU is first driven until >= 0.3 s of launches have run
(clock ramp; on top of --warmup), then the batch of EXACTLY --steps steps is timed `repeats` times
back to back, each repeat bracketed by barrier + synchronize on both sides with the MAX over
ranks taken per repeat; `ms_per_step` / `value` are those of the MEDIAN repeat.

N > 1: the transport is the library's own slab ring (csrc/capi_ring.hip: one message to and from each
neighbour per exchange on the ring's own high-priority stream -- RCCL send/recv by default, peer-mapped
direct stores with --transport ipc --, interior rows on the caller's stream); torch.distributed only
carries the 128-byte ring id, the barriers and the MAX reductions.  `--share-gpu` rehearses the N-rank
run on ONE device (every rank on GPU 0, peer-mapped transport, gloo control plane): plumbing, not a number.

N = 1 adds `secondary`: BASELINE configs 3, 4, 5 (KBC 4096^2, two-phase 8192x2048, BGK + immersed
cylinder 16384x4096) timed after the headline, each with a roofline built the same way.

The timed regions touch nothing under oracle/; the cpu_baseline leg (rank 0, N = 1 only) times
the unmodified reference (oracle/_ref) or, failing that, the CPU restatement on a bounded sample.
"""
import argparse
import ctypes as ct
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "lattice-boltzmann-method_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X spec (MI355X_MICROARCH.md:36); measured copy ceiling 6290
HBM_COPY_CEILING_GBS = 6290.0
BYTES_PER_LUP = 144.0        # 9 f64 reads + 9 f64 writes, SURVEY 8(d)
MIN_WARM_S = 0.3             # launches run before anything is timed, whatever --warmup says
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")
RING_RCCL, RING_IPC = 0, 1


# =====================================================================================================
# launcher: `python bench.py --gpus N` with no WORLD_SIZE starts its N ranks itself
# =====================================================================================================
def self_launch(a, argv):
    """One child process per rank, started BEFORE this process touches a GPU; rank 0's JSON line is relayed.
    A rank that fails ends the others (their exact pids) and its stderr is shown; exit code non-zero."""
    import torch  # device_count() does not initialise the GPU
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        print("bench.py: no GPU visible (no CPU fallback on the product path)", file=sys.stderr)
        return 2
    if a.gpus > n_dev and not a.share_gpu:
        print(f"bench.py: --gpus {a.gpus} but this machine shows {n_dev} GPU(s).  One rank per GPU is the measured "
              f"configuration; `--share-gpu` rehearses the {a.gpus}-rank schedule on GPU 0 (peer-mapped transport) "
              "without producing a scaling number.", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    tmp = tempfile.mkdtemp(prefix="lbm_bench_", dir="/tmp")
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = open(os.path.join(tmp, f"rank{r}.out"), "w")
        err = open(os.path.join(tmp, f"rank{r}.err"), "w")
        procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out, stderr=err), out, err))
    t0, why = time.time(), None
    while any(p.poll() is None for p, _, _ in procs):
        bad = [r for r, (p, _, _) in enumerate(procs) if p.poll() not in (None, 0)]
        if bad:
            why = f"rank {bad[0]} exited with code {procs[bad[0]][0].returncode}"
        elif time.time() - t0 > a.launch_timeout + 60:     # the supervisors enforce --launch-timeout themselves; this is the outer guard
            why = f"no result after {a.launch_timeout + 60} s"
        if why:
            for p, _, _ in procs:
                if p.poll() is None:
                    p.terminate()
            t1 = time.time()
            while any(p.poll() is None for p, _, _ in procs) and time.time() - t1 < 10:
                time.sleep(0.1)
            for p, _, _ in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    for p, out, err in procs:
        p.wait()
        out.close()
        err.close()
    read = lambda r, ext: open(os.path.join(tmp, f"rank{r}.{ext}")).read()
    bad = [r for r, (p, _, _) in enumerate(procs) if p.returncode != 0]
    if why or bad:
        first = bad[0] if bad else 0
        print(f"bench.py --gpus {a.gpus}: {why or 'a rank failed'}\n---- stderr of rank {first} ----\n{read(first, 'err')[-4000:]}",
              file=sys.stderr)
        return procs[first][0].returncode or 1
    sys.stderr.write(read(0, "err")[-2000:])
    lines = [ln for ln in read(0, "out").splitlines() if ln.startswith("{")]
    if not lines:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


# =====================================================================================================
# supervisor: one per rank (under torch.distributed.run or self_launch alike); the benchmark itself runs in a worker
# child, so a hang in RCCL (communicator initialisation, first exchange) costs one deadline, not the run
# =====================================================================================================
STAGES = ["spawned", "start", "ring_up", "timed", "done"]


def _stage_of(path):
    try:
        words = open(path).read().split()
    except OSError:
        return 0
    return max([STAGES.index(w) for w in words if w in STAGES] or [0])


def _end_process(p, grace=10.0):
    """terminate, then kill, the exact child we started"""
    if p.poll() is None:
        p.terminate()
        t1 = time.time()
        while p.poll() is None and time.time() - t1 < grace:
            time.sleep(0.05)
        if p.poll() is None:
            p.kill()
    p.wait()


def _strip_args(argv, names_with_value=(), flags=()):
    out, i = [], 0
    while i < len(argv):
        if argv[i] in names_with_value:
            i += 2
        elif any(argv[i].startswith(n + "=") for n in names_with_value) or argv[i] in flags:
            i += 1
        else:
            out.append(argv[i])
            i += 1
    return out


def _die_with_parent():
    """in the worker, between fork and exec: SIGKILL when the supervisor dies (PR_SET_PDEATHSIG), so that no worker
    outlives a supervisor the launcher has killed and keeps its GPU"""
    try:
        ct.CDLL("libc.so.6", use_errno=True).prctl(1, 9)
    except OSError:
        pass


def supervise(a, argv):
    """Returns the exit code.  Never initialises the GPU (torch.cuda.device_count() does not)."""
    import signal
    current = {"proc": None}

    def _on_signal(signum, _frame):                     # the launcher (or the driver's time limit) ends this rank: take the worker along
        p_ = current["proc"]
        if p_ is not None and p_.poll() is None:
            p_.kill()
        os._exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, _on_signal)
    try:
        return _supervise(a, argv, current)
    finally:
        p_ = current["proc"]
        if p_ is not None and p_.poll() is None:        # an exception on the way (a peer supervisor gone, a collective that timed out)
            _end_process(p_, grace=2.0)


def _supervise(a, argv, current):
    import datetime
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world != a.gpus:
        print(f"--gpus {a.gpus} but WORLD_SIZE={world}: either launch N ranks (torch.distributed.run) or unset WORLD_SIZE "
              "and let bench.py start them", file=sys.stderr)
        return 2
    n_dev = torch.cuda.device_count()
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if not a.share_gpu and local_rank >= n_dev:
        print(f"rank {rank}: LOCAL_RANK {local_rank} but {n_dev} GPU(s) visible (one rank per GPU; --share-gpu puts every rank "
              "on GPU 0 for a rehearsal)", file=sys.stderr)
        return 2
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    t_run = time.time()
    box = [tempfile.mkdtemp(prefix="lbm_bench_sup_", dir="/tmp") if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    base = box[0]
    first = "ipc" if a.share_gpu else a.transport
    plan = [(first, "gloo" if (a.share_gpu or a.ctl == "gloo") else "nccl")]
    if not a.no_fallback:
        plan.append(("ipc", "gloo"))         # the second attempt uses no RCCL at all
    worker_argv = _strip_args(argv, names_with_value=("--transport", "--ctl"))
    script = a.worker_script or os.path.abspath(__file__)
    history, code, line = [], 1, None
    for attempt, (transport, ctlb) in enumerate(plan):
        port = [0]
        if rank == 0:
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                port[0] = s_.getsockname()[1]
        dist.broadcast_object_list(port, src=0)
        stem = os.path.join(base, f"a{attempt}_rank{rank}")
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_") and k != "GLOO_SOCKET_IFNAME"}
        env.update(RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]), HSA_ENABLE_IPC_MODE_LEGACY="0")
        open(stem + ".stage", "w").write("spawned\n")
        with open(stem + ".out", "w") as fo, open(stem + ".err", "w") as fe:
            proc = subprocess.Popen([sys.executable, script] + worker_argv +
                                    ["--worker", "--status-file", stem + ".stage", "--transport", transport, "--ctl", ctlb,
                                     "--attempt", str(attempt)], env=env, stdout=fo, stderr=fe, preexec_fn=_die_with_parent)
        current["proc"] = proc                          # (the signal handler and the exception path end exactly this child)
        t0, verdict, table = time.time(), None, None
        while verdict is None:
            time.sleep(0.2)
            rc = proc.poll()
            mine = [float(_stage_of(stem + ".stage")), 0.0 if rc is None else (1.0 if rc == 0 else 2.0), 0.0, 0.0]
            if rank == 0:                               # rank 0's clock decides for everybody
                mine[2] = float(time.time() - t0 > a.ring_deadline)
                mine[3] = float(time.time() - t_run > a.launch_timeout)
            t = torch.tensor(mine, dtype=torch.float64)
            rows = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(rows, t)
            table = [r_.tolist() for r_ in rows]
            stages = [int(r_[0]) for r_ in table]
            failed = [i for i, r_ in enumerate(table) if r_[1] == 2.0]
            late = [i for i, st in enumerate(stages) if st < STAGES.index("ring_up")]
            if all(r_[1] == 1.0 for r_ in table):
                verdict = ("ok", None)
            elif failed:
                verdict = ("failed", f"rank {failed[0]} exited with an error at stage '{STAGES[stages[failed[0]]]}'", failed[0])
            elif table[0][3] > 0:
                slow = min(range(world), key=lambda i: stages[i])
                verdict = ("timeout", f"no result {a.launch_timeout:.0f} s after the start (rank {slow} at stage '{STAGES[stages[slow]]}')", slow)
            elif table[0][2] > 0 and late:
                # (a rank that waits in a collective for a stalled peer is late too: every late rank is named)
                verdict = ("ring_deadline", f"rank(s) {late} had not brought the ring up {a.ring_deadline:.0f} s after the workers started "
                                            f"(stages {[STAGES[stages[i]] for i in late]})", late[0], late)
        if verdict[0] == "ok":
            proc.wait()
            code = 0
            if rank == 0:
                lines = [ln for ln in open(stem + ".out").read().splitlines() if ln.startswith("{")]
                line = lines[-1] if lines else None
                sys.stderr.write(open(stem + ".err").read()[-2000:])
            break
        _end_process(proc)
        dist.barrier()                                  # every worker of this attempt is gone before the next set starts
        history.append({"attempt": attempt, "transport": transport, "control_plane": ctlb, "gave_up": verdict[1]})
        before_ring = verdict[0] == "ring_deadline" or (verdict[0] == "failed" and int(table[verdict[2]][0]) < STAGES.index("ring_up"))
        if verdict[0] == "timeout" or not before_ring or attempt + 1 == len(plan):
            if rank == 0:
                print(f"bench.py --gpus {world}: {verdict[1]}", file=sys.stderr)
                for r_ in (verdict[3] if len(verdict) > 3 else [verdict[2]]):
                    err = ""
                    try:
                        err = open(os.path.join(base, f"a{attempt}_rank{r_}.err")).read()[-(4000 // max(1, len(verdict[3]) if len(verdict) > 3 else 1)):]
                    except OSError:
                        pass
                    print(f"---- stderr of rank {r_} (attempt {attempt}, transport {transport}) ----\n{err}", file=sys.stderr)
                sys.stderr.flush()
            code = 1
            break
        if rank == 0:
            print(f"bench.py: {verdict[1]}; starting fresh workers on the peer-mapped transport (gloo control plane)", file=sys.stderr, flush=True)
    if code == 0 and rank == 0:
        if not line:
            print("bench.py: rank 0's worker printed no result line", file=sys.stderr)
            code = 1
        else:
            out = json.loads(line)
            out["launcher"] = {"supervised": True, "attempts": len(history) + 1, "ring_deadline_s": a.ring_deadline,
                               "run_s": round(time.time() - t_run, 1)}
            if history:
                out["launcher"]["transport_fallback"] = history
            print(json.dumps(out), flush=True)
            if box_failed(out, 0):
                code = 3
    flag = [code]
    dist.broadcast_object_list(flag, src=0)
    dist.barrier()
    if rank == 0 and flag[0] == 0:      # (a failed run keeps its workers' logs for the post-mortem)
        import shutil
        shutil.rmtree(base, ignore_errors=True)
    dist.destroy_process_group()
    return flag[0]


# =====================================================================================================
# control plane (torch.distributed: id broadcast, barriers, reductions -- nothing of the data path)
# =====================================================================================================
class Ctl:
    def __init__(self, rank, world, dev, backend):
        import torch
        import torch.distributed as dist
        self.rank, self.world, self.dist, self.torch = rank, world, dist, torch
        self.dev = dev if backend == "nccl" else torch.device("cpu")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if backend == "gloo":
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def reduce(self, values, op):
        """list of floats -> list of floats, MAX / MIN / SUM over ranks"""
        if self.world == 1:
            return list(values)
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN, "sum": self.dist.ReduceOp.SUM}[op])
        return t.tolist()

    def gather(self, values):
        if self.world == 1:
            return [list(values)]
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.dev)
        g = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(g, t)
        return [x.tolist() for x in g]

    def bcast_bytes(self, b):
        if self.world == 1:
            return bytes(b)
        t = self.torch.tensor(list(bytes(b)), dtype=self.torch.uint8, device=self.dev)
        self.dist.broadcast(t, 0)
        return bytes(t.cpu().tolist())

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def taylor_green(lib, torch, _ptr, R_local, C, row0, R_global, dev, U=0.04):
    """rho = 1, Taylor-Green vortex on the GLOBAL box; returns SoA f = feq(rho, u) [9,R,C]."""
    r = (torch.arange(R_local, device=dev, dtype=torch.float64) + row0).view(-1, 1)
    c = torch.arange(C, device=dev, dtype=torch.float64).view(1, -1)
    kr, kc = 2 * math.pi / R_global, 2 * math.pi / C
    u = torch.empty((2, R_local, C), dtype=torch.float64, device=dev)
    u[0] = U * torch.sin(kr * r) * torch.cos(kc * c)
    u[1] = -U * torch.cos(kr * r) * torch.sin(kc * c)
    rho = torch.ones((R_local, C), dtype=torch.float64, device=dev)
    f = torch.empty((9, R_local, C), dtype=torch.float64, device=dev)
    lib.equilibrium(_ptr(f), _ptr(u), _ptr(rho), R_local, C, None)
    return f


def cpu_baseline(rows=1024, cols=1024, budget_s=15.0):
    """Reference (oracle/_ref, libtorch CPU, all torch threads) on a bounded sample of the same
    workload; the CPU restatement (OpenMP) beside it.  Checker code: never on the product path."""
    import numpy as np
    from pyoracle import Oracle, Ref
    orc = Oracle()
    rng = np.random.default_rng(0)
    rho = np.ones((rows, cols))
    u = 0.04 * rng.standard_normal((rows, cols, 2))
    f0 = orc.equilibrium(u, rho)

    def timed(fn, threads):
        fn(f0, 1.2, 1)  # warm
        t0 = time.perf_counter(); fn(f0, 1.2, 2); per = (time.perf_counter() - t0) / 2
        n = max(2, min(200, int(budget_s / max(per, 1e-6))))
        t0 = time.perf_counter(); fn(f0, 1.2, n); dt = time.perf_counter() - t0
        return dict(value=round(rows * cols * n / dt / 1e6, 3), unit="MLUPS", cores=threads,
                    sample=f"{rows}x{cols} periodic BGK f64, {n} steps in {dt:.1f} s")

    all_threads = orc.max_threads()
    port = timed(orc.bgk_periodic_steps, all_threads)
    port["kind"] = "port"
    # the restatement on ONE core beside all cores (the OpenMP team is resized in place)
    budget_s = min(budget_s, 6.0)
    orc.set_threads(1)
    try:
        port["one_core"] = timed(orc.bgk_periodic_steps, 1)
    finally:
        orc.set_threads(all_threads)
    out = port
    if Ref.available():
        try:
            ref = Ref()
            budget_s = 15.0
            out = timed(ref.bgk_periodic_steps, ref.num_threads())
            out["kind"] = "reference"
            out["port"] = port
        except OSError as e:  # libtorch not loadable on this host
            out["note"] = f"oracle/_ref unusable: {e}"
    return out


def cpu_secondary(which, budget_s=6.0):
    """CPU figure beside a secondary workload (C3 at 1024^2, C4 at 512 x 256, C5 at 2048 x 512): the
    OpenMP restatement on all host cores and, for KBC, the unmodified reference's `kbc` (oracle/_ref) beside it.
    Checker code, after the timed regions, bounded to a few seconds each."""
    import numpy as np
    from pyoracle import Oracle, Ref, cg_params
    orc = Oracle()
    cores = orc.max_threads()

    def timed(step_n, nodes, threads, what, kind):
        t0 = time.perf_counter(); step_n(1); per = time.perf_counter() - t0       # warm-up and pilot in one (the reference's KBC step takes seconds)
        if per < 0.5:
            t0 = time.perf_counter(); step_n(2); per = (time.perf_counter() - t0) / 2
        n = max(1 if per > 2.0 else 2, min(100, int(budget_s / max(per, 1e-6))))
        t0 = time.perf_counter(); step_n(n); dt = time.perf_counter() - t0
        return dict(value=round(nodes * n / dt / 1e6, 3), unit="MLUPS", cores=threads, kind=kind, sample=f"{what}, {n} steps in {dt:.1f} s")

    if which == "kbc":
        R = C = 1024
        s2 = 1.0 / (0.5 + 3 * 1.70766666e-4)
        m0, m1 = orc.kbc_shear_init(R, C)
        f = orc.kbc_equilibrium(m0, m1)
        out = timed(lambda n: orc.kbc_steps(f, m0, m1, s2, n), R * C, cores, f"{R}x{C} KBC shear layer f64", "port")
        if Ref.available():
            try:
                ref = Ref()
                r_ = timed(lambda n: ref.kbc_steps(f, m0, m1, s2, n), R * C, ref.num_threads(), f"{R}x{C} KBC shear layer f64 (ulbm::d2q9::kbc)", "reference")
                r_["port"] = out
                out = r_
            except OSError as e:
                out["note"] = f"oracle/_ref unusable: {e}"
        return out
    if which == "cg":
        R, C = 512, 256
        p = cg_params(R, C)
        st = orc.cg_init(p)
        return timed(lambda n: orc.cg_steps(p, st, n), R * C, cores, f"{R}x{C} colour-gradient two-phase MRT f64, gamma3 parameters", "port")
    if which == "ibm":
        R, C = 2048, 512
        m = int(round(np.pi * 37.5))
        t = 2 * np.pi * np.arange(m) / m
        x, y = R / 4.0 + 18.75 * np.cos(t), C / 2.0 + 18.75 * np.sin(t)
        u = np.zeros((R, C, 2)); u[..., 0] = 0.04
        f = orc.incomp_equilibrium(u, np.ones((R, C)))
        return timed(lambda n: orc.cylinder_steps(x, y, f, 1.0 / 0.55, 0.04, n), R * C, cores,
                     f"{R}x{C} BGK + immersed cylinder (d = 37.5, {m} markers) f64", "port")
    raise ValueError(which)


class Box:
    """The per-rank slab: two SoA lattices with padded planes and (when split) D ghost rows, the
    launch that advances them, and -- with ghost rows -- the library's slab ring."""

    def __init__(self, lib, a, ctl, dev, with_ring, transport):
        import torch
        import pylbm
        self.torch, self.pylbm = torch, pylbm
        self.lib, self.a, self.ctl, self.dev = lib, a, ctl, dev
        rank, world = ctl.rank, ctl.world
        R, C = a.rows, a.cols
        self.R, self.C = R, C
        self.prm = pylbm.BgkParams(a.omega, 0)
        self.bc = pylbm.Bc.periodic()
        self.depth = a.xn if (a.xn >= 2 and C >= 64 and R >= max(4 * a.xn + 8, 4 * a.edge_rows)) else 1
        # ghost = period x D rows: the ring exchanges once per `period` launches (capi_ring.hip ring_bgk_step)
        self.period = max(1, a.ring_period) if (with_ring and self.depth > 1) else 1
        self.ghost = self.depth * self.period if with_ring else 0
        rows = R + 2 * self.ghost
        pad = a.plane_pad if a.plane_pad is not None else lib.default_plane_pad(rows, C)
        self.plane = rows * C + pad
        self.geom = pylbm.Geom(R, C, self.ghost, self.plane if pad else 0)
        self.buf = [torch.zeros(9 * self.plane, dtype=torch.float64, device=dev) for _ in range(2)]
        self.lat = [b.as_strided((9, rows, C), (self.plane, C, 1)) for b in self.buf]
        self.cur = 0
        self.ring, self.transport, self.transport_note = None, None, None
        if with_ring:
            self.ring = self._make_ring(transport)
            if self.ring is None and transport == RING_RCCL:
                # RCCL could not be brought up on every rank: the peer-mapped transport needs no RCCL at all
                self.transport_note = "RCCL ring creation failed on at least one rank; fell back to the peer-mapped transport"
                self.ring = self._make_ring(RING_IPC)
            if self.ring is None:
                raise SystemExit(f"rank {rank}: no slab ring could be created: {lib.raw.lbm_last_error_string().decode()}")

    def _make_ring(self, transport):
        """collective: every rank tries; the ring is kept only if ALL ranks have one"""
        lib, ctl = self.lib, self.ctl
        ident = (ct.c_ubyte * 128)()
        if ctl.rank == 0:
            lib.ring_unique_id_ex(ident, transport)
        ident = (ct.c_ubyte * 128).from_buffer_copy(ctl.bcast_bytes(bytes(ident)))
        ring = ct.c_void_p()
        rc = lib.raw.lbm_ring_create_ex(ct.byref(ring), ident, ctl.rank, ctl.world, ct.byref(self.geom), 1, transport)
        ok = ctl.reduce([1.0 if rc == 0 else 0.0], "min")[0] > 0
        if not ok:
            if rc == 0:
                lib.ring_destroy(ring)
            elif ctl.rank == 0:
                print(f"bench.py: lbm_ring_create_ex(transport {transport}): {lib.raw.lbm_last_error_string().decode()}", file=sys.stderr, flush=True)
            return None
        self.transport = transport
        return ring

    def stream(self):
        return ct.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def owned(self):
        return self.lat[self.cur][:, self.ghost:self.ghost + self.R, :]

    def load(self, f_pre):
        """f_pre [9,R,C]: pre-collision populations; resident state = collide(f_pre) + ghost fill"""
        _ptr = self.pylbm._ptr
        flat = self.pylbm.Geom(self.R, self.C, 0)
        p = self.torch.empty_like(f_pre)
        self.lib.bgk_collide(_ptr(p), _ptr(f_pre), ct.byref(flat), None, ct.byref(self.prm), None, None, self.stream())
        self.owned().copy_(p)
        if self.ring:
            self.lib.ring_exchange(self.ring, _ptr(self.lat[self.cur]), self.stream())
            self.lib.ring_join(self.ring, self.stream())

    def launch(self, n_steps):
        _ptr = self.pylbm._ptr
        src, dst = self.lat[self.cur], self.lat[self.cur ^ 1]
        lib, g, bc, prm = self.lib, ct.byref(self.geom), ct.byref(self.bc), ct.byref(self.prm)
        if self.ring:
            lib.ring_bgk_step(self.ring, _ptr(dst), _ptr(src), None, prm, n_steps, self.a.edge_rows, self.stream())
        elif n_steps == 1:
            lib.bgk_stream_collide(_ptr(dst), _ptr(src), g, bc, prm, 0, self.R, None, None, self.stream())
        else:
            lib.bgk_stream_collide_xn(_ptr(dst), _ptr(src), g, bc, prm, n_steps, 0, self.R, self.stream())
        self.cur ^= 1

    def selfcheck(self, D):
        """Two launches with the ring's default schedule (ghost = period x D rows: the first without an exchange) and
        the same two with an exchange on every launch, from the same state: owned rows must agree bit for bit on
        every rank.  The state is restored afterwards."""
        torch, _ptr = self.torch, self.pylbm._ptr
        saved, cur0 = [b.clone() for b in self.buf], self.cur

        def restore():
            for b, s_ in zip(self.buf, saved):
                b.copy_(s_)
            self.cur = cur0
            self.lib.ring_exchange(self.ring, _ptr(self.lat[self.cur]), self.stream())
            self.lib.ring_join(self.ring, self.stream())
        self.launch(D)
        self.launch(D)
        torch.cuda.synchronize()
        first = self.owned().clone()
        self.lib.set_tuning(b"ring_period", 1)
        restore()
        self.launch(D)
        self.launch(D)
        torch.cuda.synchronize()
        ok = float(torch.equal(first, self.owned())) if self.lib.raw.lbm_ring_status(self.ring) == 0 else 0.0
        self.lib.set_tuning(b"ring_period", -1)
        restore()
        torch.cuda.synchronize()
        return self.ctl.reduce([ok], "min")[0] > 0

    def advance(self, n):
        """n time steps: n // D window launches, the remainder in single steps"""
        for _ in range(n // self.depth):
            self.launch(self.depth)
        for _ in range(n % self.depth):
            self.launch(1)

    def launches(self, n):
        return n // self.depth + n % self.depth

    def close(self):
        if self.ring:
            self.lib.ring_destroy(self.ring)
            self.ring = None


# =====================================================================================================
# HBM traffic by PMC counters: child runs of this script under rocprofv3 --pmc
# =====================================================================================================
def _pmc_run(counter, child_args, timeout_s):
    """one `rocprofv3 --pmc <counter>` pass over `python bench.py --pmc-child ...`; returns the rows of its
    counter_collection.csv (dispatch order) or raises"""
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise OSError("rocprofv3 not found")
    d = tempfile.mkdtemp(prefix="lbm_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__)] + child_args
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
        rows = []
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows += [row for row in csv.DictReader(open(path)) if row["Counter_Name"] == counter]
        if r.returncode != 0 or not rows:
            raise OSError(f"{counter} pass failed (rc {r.returncode}, {len(rows)} rows): {r.stderr[-300:]}")
        rows.sort(key=lambda row: int(row["Dispatch_Id"]))
        return rows
    finally:
        shutil.rmtree(d, ignore_errors=True)


def power_probe(run_some, seconds=1.5):
    """Shader clock and package power WHILE the workload runs: rocm-smi polled from a thread beside `seconds` of the same
    launches, after the timed region (a host-side query: no GPU work of its own).  These f64 kernels run into the
    package power limit, and the clock the card then settles at -- not 2.4 GHz -- is what the timed region ran on."""
    import re
    import shutil
    import subprocess
    import threading
    import time
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                t = subprocess.run([smi, "-d", "0", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                return
            c = re.search(r"sclk clock level:[^(]*\((\d+)Mhz\)", t)
            w = re.search(r"Power \(W\):\s*([0-9.]+)", t)
            if c and w:
                samples.append((int(c.group(1)), float(w.group(1))))
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        run_some()
    stop.set()
    th.join(timeout=10)
    cap = None
    try:
        t = subprocess.run([smi, "-d", "0", "--showmaxpower"], capture_output=True, text=True, timeout=5).stdout
        m = re.search(r"Power \(W\):\s*([0-9.]+)", t)
        cap = float(m.group(1)) if m else None
    except Exception:
        pass
    busy = [x for x in samples if x[0] > 600]   # samples that caught the card idle between batches do not count
    if not busy:
        return None
    import statistics as st
    return {"sclk_mhz": int(st.median(x[0] for x in busy)), "package_w": round(st.median(x[1] for x in busy), 1),
            "package_limit_w": cap, "samples": len(busy),
            "how": f"rocm-smi polled beside {seconds} s of the same launches, after the timed region"}


def pmc_traffic(argv_tail, kernel_tag="k_stream_collide_sw", timeout_s=150):
    """HBM bytes per launch of the headline's kernel, measured NOW: two child runs of this script under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE each in its own pass, as MI355X_MICROARCH.md prescribes;
    FETCH_SIZE x 2 = the gfx950 correction for wide coalesced reads, calibrated in profiles/).  Returns
    (fetch_bytes, write_bytes, note) or (None, None, reason).  The children only launch the kernel a few
    times (--pmc-child); nothing here is timed."""
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            rows = _pmc_run(counter, ["--pmc-child", "headline"] + argv_tail, timeout_s)
            vals = [float(r["Counter_Value"]) for r in rows if kernel_tag in r["Kernel_Name"]]
            if len(vals) < 3:
                return None, None, f"{counter} pass: {len(vals)} samples"
            out[counter] = statistics.median(vals[2:]) * 1024.0     # KiB; the first launches warm the caches
        except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
            return None, None, f"{counter} pass: {type(e).__name__}: {e}"
    return 2.0 * out["FETCH_SIZE"], out["WRITE_SIZE"], "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (FETCH_SIZE x 2: gfx950)"


def pmc_sq(argv_tail, kernel_tag="k_stream_collide_sw", timeout_s=150):
    """one more child pass with SQ counters: share of wave cycles issuing VALU and VALU wave-instructions per launch of the
    headline's kernel, measured in THIS run (the figure was a committed constant)"""
    try:
        import csv  # noqa: F401
        rows = _pmc_run_multi(["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_ANY"],
                              ["--pmc-child", "headline"] + argv_tail, timeout_s)
        acc = {}
        for r in rows:
            if kernel_tag in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        med = {k: statistics.median(v[2:] if len(v) > 3 else v) for k, v in acc.items()}
        if not all(k in med for k in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU")):
            return None
        return {"valu_issue_frac": round(med["SQ_ACTIVE_INST_VALU"] / med["SQ_WAVE_CYCLES"], 4),
                "valu_wave_instructions_per_launch": med["SQ_INSTS_VALU"],
                "waitcnt_frac": round(med.get("SQ_WAIT_ANY", 0.0) / med["SQ_WAVE_CYCLES"], 4)}
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
        return None


def _pmc_run_multi(counters, child_args, timeout_s):
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise OSError("rocprofv3 not found")
    d = tempfile.mkdtemp(prefix="lbm_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__)] + child_args
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
        rows = []
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows += list(csv.DictReader(open(path)))
        if r.returncode != 0 or not rows:
            raise OSError(f"SQ pass failed (rc {r.returncode})")
        rows.sort(key=lambda row: int(row["Dispatch_Id"]))
        return rows
    finally:
        shutil.rmtree(d, ignore_errors=True)


def pmc_between_markers(which, extra_args=(), timeout_s=240):
    """HBM bytes of EVERYTHING a secondary workload launches between its two lbm_marker kernels (the child runs
    set-up, marker, n steps, marker), per kernel name.  Returns {counter: {kernel_name: [sum_KiB, dispatches]}}."""
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = _pmc_run(counter, ["--pmc-child", which] + list(extra_args), timeout_s)
        marks = [i for i, r in enumerate(rows) if "k_lbm_marker" in r["Kernel_Name"]]
        if len(marks) != 2:
            raise OSError(f"{counter} pass of {which}: {len(marks)} markers in the trace")
        per = {}
        for r in rows[marks[0] + 1:marks[1]]:
            e = per.setdefault(r["Kernel_Name"].split("(")[0][:120], [0.0, 0])
            e[0] += float(r["Counter_Value"])
            e[1] += 1
        out[counter] = per
    return out


# =====================================================================================================
# secondary workloads: BASELINE configs 3, 4, 5 on one GPU
# =====================================================================================================
class Secondary:
    """name, nodes, steps per launch group, algorithmic bytes per node and step, and a step(n) callable"""

    def __init__(self, lib, dev, which):
        import numpy as np
        import torch
        import pylbm
        from pylbm import _ptr
        self.lib, self.which, self.torch = lib, which, torch
        # placement probe: a throw-away allocation of LBM_BENCH_PREALLOC_MB before the solver's own moves its lattices to
        # other (virtual and physical) addresses -- to tell a placement effect from a kernel effect
        self._placement = torch.empty(int(os.environ.get("LBM_BENCH_PREALLOC_MB", "0")) << 20, dtype=torch.uint8, device=dev)
        if which == "kbc":      # config 3: ulbm_double_shear_flow.cpp:42-63 at 4096^2 (s2 = omega, nu = 1.70766666e-4)
            R, C = 4096, int(os.environ.get("LBM_BENCH_KBC_COLS", "4096"))   # (other widths: a probe of row-stride effects)
            self.unit = int(lib.raw.lbm_get_tuning(b"kbc_depth")) or 4
            self.kernel = f"k_stream_collide_sw<KbcFastModel,{self.unit},2,nt>"
            self.bytes_per_update, self.config = 144.0, "ulbm_double_shear_flow 4096x4096 KBC (entropic MRT), periodic"
            self.sv = pylbm.Solver(lib, pylbm.MODEL_KBC, R, C, pylbm.KbcParams(1.0 / (0.5 + 3 * 1.70766666e-4)))
            f = taylor_green(lib, torch, _ptr, R, C, 0, R, dev, U=0.02)
            lib.solver_set_f_soa_dev(self.sv.h, _ptr(f))
            self.step = lambda n: self.sv.step(n)
        elif which == "cg":     # config 4: mrtcg_rayleigh_taylor.cpp:182-210 (init_rho_cosine) at 8192 x 2048
            R, C = 8192, int(os.environ.get("LBM_BENCH_CG_COLS", "2048"))   # (other widths: a probe of row-stride effects)
            self.unit = 1
            self.kernel = ("k_cg_tile_mn<16,64,512,4,parked> (inner rectangle: 16x64 tiles, 2 nodes per thread, patches of 4x2 tiles per XCD) "
                           "+ k_cg_fused<16,32,4,.,2> on the frame")
            self.bytes_per_update, self.config = 288.0, "mrtcg_rayleigh_taylor 8192x2048 colour-gradient two-phase MRT, gamma3 parameters"
            prm = pylbm.cg_params()
            rr = np.arange(R).reshape(-1, 1)
            s = R / 2.0 - 0.1 * C * np.cos(2.0 * 3.141592 * np.arange(C) / C).reshape(1, -1)
            rho_r, rho_b = 3.0 * (rr < s), 1.0 * (rr >= s)
            d_rr, d_rb = (torch.from_numpy(x.astype(np.float64)).to(dev) for x in (rho_r, rho_b))
            d_u = torch.zeros((2, R, C), dtype=torch.float64, device=dev)
            f_r, f_b = (torch.empty((9, R, C), dtype=torch.float64, device=dev) for _ in range(2))
            lib.cg_equilibrium(_ptr(f_r), _ptr(d_rr), _ptr(d_u), ct.byref(prm.red), R, C, ct.c_longlong(0), None)
            lib.cg_equilibrium(_ptr(f_b), _ptr(d_rb), _ptr(d_u), ct.byref(prm.blue), R, C, ct.c_longlong(0), None)
            torch.cuda.synchronize()
            self.sv = pylbm.CgSolver(lib, R, C, prm)
            self.sv.set_state(np.moveaxis(f_r.cpu().numpy(), 0, -1), np.moveaxis(f_b.cpu().numpy(), 0, -1), rho_r, rho_b, np.zeros((R, C, 2)))
            self.step = lambda n: self.sv.step(n)
        elif which == "ibm":    # config 5: cylinder_test.cpp:88-164, diameter 300 at rows / 4
            R, C = 16384, 4096
            self.unit = int(lib.raw.lbm_get_tuning(b"ibm_depth")) or 5
            self.kernel = "k_stream_collide_sw<BgkFastModel,5,2,nt> + wall frame (rows away from the cylinder) beside the forced box chain"
            self.bytes_per_update, self.config = 144.0, "cylinder_test 16384x4096 BGK + immersed cylinder (d = 300, 942 markers), one block"
            omega, u_in = 1.0 / 0.55, 0.04
            m = int(round(np.pi * 300))
            t = 2 * np.pi * np.arange(m) / m
            x, y = R / 4.0 + 150 * np.cos(t), C / 2.0 + 150 * np.sin(t)
            bc = pylbm.Bc(row_lo=pylbm.EDGE_ABB_VELOCITY, row_hi=pylbm.EDGE_ABB_VELOCITY, col_lo=pylbm.EDGE_SPECULAR,
                          col_hi=pylbm.EDGE_SPECULAR, uw_r=u_in)
            self.sv = pylbm.Solver(lib, pylbm.MODEL_BGK, R, C, pylbm.BgkParams(omega, 0, 1), bc=bc)
            self.ib = pylbm.Ibm(lib, x, y, R, C)
            self.sv.attach_ibm(self.ib)
            u = torch.zeros((2, R, C), dtype=torch.float64, device=dev); u[0] = u_in
            rho = torch.ones((R, C), dtype=torch.float64, device=dev)
            f = torch.empty((9, R, C), dtype=torch.float64, device=dev)
            lib.incomp_equilibrium(_ptr(f), _ptr(u), _ptr(rho), R, C, None)
            lib.solver_set_f_soa_dev(self.sv.h, _ptr(f))
            self.step = lambda n: self.sv.step(n)
        else:
            raise SystemExit(f"unknown secondary workload {which}")
        self.R, self.C = R, C
        torch.cuda.synchronize()
        self.step(1)             # the driver's first iteration (collide-only launch / forcing on the initial state)
        torch.cuda.synchronize()

    def close(self):
        self.sv.close()
        if hasattr(self, "ib"):
            self.ib.close()


PMC_GROUPS = {"kbc": 4, "cg": 16, "ibm": 4}   # launch groups between the markers (cg: the LAST step of a call also writes
                                              # the observable fields, 48 B per node: 1 step in 16 keeps that below 1 %)


def secondary_child(lib, dev, which):
    """under rocprofv3 --pmc: set-up, marker, PMC_GROUPS launch groups, marker"""
    import torch
    groups = PMC_GROUPS[which]
    w = Secondary(lib, dev, which)
    w.step(w.unit)                       # one group outside the bracket (first-touch effects)
    torch.cuda.synchronize()
    lib.marker(1, None)
    torch.cuda.synchronize()
    w.step(w.unit * groups)
    torch.cuda.synchronize()
    lib.marker(2, None)
    torch.cuda.synchronize()
    w.close()


def run_secondary(lib, dev, which, a):
    import torch
    w = Secondary(lib, dev, which)
    n = max(w.unit, (a.secondary_steps // w.unit) * w.unit)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < a.min_warm_s:
        w.step(n)
        torch.cuda.synchronize()
    wall, devms = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        w.step(n)
        ev1.record()
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
        devms.append(ev0.elapsed_time(ev1))
    power = None
    if not a.no_power:
        def _some():
            w.step(n)
            torch.cuda.synchronize()
        power = power_probe(_some, 1.0)
    # informative second figure of config 4 (as `reference_order` is for the headline): the same steps through round 3's
    # default, the 16 x 32 tile kernel with one node per thread (tuning cg_big = 0) -- same bits, more bytes
    opt_in = None
    if which == "cg" and not any(kv.startswith("cg_big=") for kv in a.tune):
        lib.set_tuning(b"cg_big", 0)
        try:
            w.step(n)
            torch.cuda.synchronize()
            ws = []
            for _ in range(3):
                t0 = time.perf_counter()
                w.step(n)
                torch.cuda.synchronize()
                ws.append(time.perf_counter() - t0)
            opt_in = {"kernel": "k_cg_fused<16,32,4> (tuning cg_big = 0: the default of rounds 1-3) on the inner rectangle + its frame instantiation",
                      "value": round(w.R * w.C * n / sorted(ws)[1] / 1e6, 1), "unit": "MLUPS", "steps": n, "repeats": 3,
                      "note": "bit-identical to the default kernel (tests/test_gpu_cg.py, tests/test_gpu_fullsize.py)"}
            if not a.no_pmc:
                try:
                    pm = pmc_between_markers(which, ["--tune", "cg_big=0"] + [x for kv in a.tune for x in ("--tune", kv)])
                    g_ = PMC_GROUPS[which]
                    fe = 2.0 * 1024.0 * sum(v[0] for v in pm["FETCH_SIZE"].values()) / g_
                    wr = 1024.0 * sum(v[0] for v in pm["WRITE_SIZE"].values()) / g_
                    opt_in.update(fetch_bytes=fe, write_bytes=wr,
                                  traffic_over_algorithmic=round((fe + wr) / (w.R * w.C * w.bytes_per_update * w.unit), 3))
                except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
                    opt_in["traffic_source"] = f"PMC passes failed: {type(e).__name__}: {e}"
        finally:
            lib.set_tuning(b"cg_big", -1)
    w.close()
    # config 5's second figure: the same block with the collision in the reference's operation order (parameters' form =
    # LBM_FORM_REFERENCE_ORDER) -- BITWISE equal to the oracle (tests/test_gpu_ibm.py, tests/test_gpu_fullsize.py); the
    # default since round 4 is the reassociated collision, as for every other model (1e-10 against the oracle)
    if which == "ibm" and not any(kv.startswith("bgk_fast_delta=") for kv in a.tune):
        lib.set_tuning(b"bgk_fast_delta", 0)
        try:
            w2 = Secondary(lib, dev, which)
            w2.step(n)
            torch.cuda.synchronize()
            ws = []
            for _ in range(3):
                t0 = time.perf_counter()
                w2.step(n)
                torch.cuda.synchronize()
                ws.append(time.perf_counter() - t0)
            w2.close()
            opt_in = {"kernel": "the same block in the reference's operation order: far rows through k_stream_collide_sw<BgkModelT<0,1>,5,2,nt> "
                                "(LBM_FORM_REFERENCE_ORDER / tuning bgk_fast_delta = 0)",
                      "value": round(w.R * w.C * n / sorted(ws)[1] / 1e6, 1), "unit": "MLUPS", "steps": n, "repeats": 3,
                      "note": "bitwise equal to the CPU oracle (tests/test_gpu_fullsize.py::test_config5_fullsize_ibm_block_vs_oracle); "
                              "the default (reassociated) collision: 1e-10 relative after 13 steps (tests/test_gpu_ibm.py)"}
        finally:
            lib.set_tuning(b"bgk_fast_delta", -1)
    mid = sorted(range(5), key=lambda i: wall[i])[2]
    dt, group_ms = wall[mid], devms[mid] / (n // w.unit)
    nodes = w.R * w.C
    alg = nodes * w.bytes_per_update * w.unit           # algorithmic bytes one launch group stands for
    out = {"config": w.config, "value": round(nodes * n / dt / 1e6, 1), "unit": "MLUPS", "steps": n, "repeats": 5,
           "ms_per_step": round(dt / n * 1e3, 4), "dtype": "f64", "kernel": w.kernel, "steps_per_launch": w.unit,
           "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                        "kernel_ms": round(group_ms, 4),
                        "kernel_ms_is": "device time (HIP events on the launch stream) per launch group = all kernels of "
                                        f"{w.unit} step(s), helper-stream launches included",
                        "algorithmic_bytes_per_launch": alg, "algorithmic_GBs": round(alg / (group_ms * 1e-3) / 1e9, 1),
                        "algorithmic_multiple": round(alg / (group_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    if power:
        out["roofline"]["power"] = power
    if opt_in:
        out["opt_in"] = opt_in
    if not a.no_pmc:
        groups = PMC_GROUPS[which]
        try:
            pm = pmc_between_markers(which, [x for kv in a.tune for x in ("--tune", kv)])
            fetch = 2.0 * 1024.0 * sum(v[0] for v in pm["FETCH_SIZE"].values()) / groups
            write = 1024.0 * sum(v[0] for v in pm["WRITE_SIZE"].values()) / groups
            traffic = fetch + write
            ach = traffic / (group_ms * 1e-3) / 1e9
            roof = out["roofline"]
            roof.update(traffic=traffic, achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4),
                        frac_of_copy_ceiling=round(ach / HBM_COPY_CEILING_GBS, 4),
                        traffic_bytes_per_update=round(traffic / (nodes * w.unit), 2),
                        traffic_over_algorithmic=round(traffic / alg, 3),
                        traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run, every launch between two "
                                       "marker kernels (FETCH_SIZE x 2: gfx950), per launch group",
                        pmc={"fetch_bytes": fetch, "write_bytes": write,
                             "per_kernel_KiB_per_group": {k: {"fetch_raw": round(v[0] / groups, 1), "launches": v[1] / groups,
                                                              "write": round(pm["WRITE_SIZE"].get(k, [0.0, 0])[0] / groups, 1)}
                                                          for k, v in pm["FETCH_SIZE"].items()}})
        except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
            out["roofline"]["traffic_source"] = f"PMC passes failed: {type(e).__name__}: {e}"
        # the second roof (SURVEY 8(d) "which roofline": report both where the HBM fraction plateaus): f64 VALU issue.
        # One SQ pass over the same launch groups: VALU wave-instructions x 64 lanes per update, against 256 CUs x 4 SIMDs x
        # 16 lanes per clock at the shader clock sampled beside the same launches.  `bound` names the higher fraction.
        try:
            rows = _pmc_run_multi(["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"],
                                  ["--pmc-child", which] + [x for kv in a.tune for x in ("--tune", kv)], 240)
            marks = sorted({int(r["Dispatch_Id"]) for r in rows if "k_lbm_marker" in r["Kernel_Name"]})
            if len(marks) != 2:
                raise OSError(f"SQ pass of {which}: {len(marks)} markers in the trace")
            acc = {}
            for r in rows:
                if marks[0] < int(r["Dispatch_Id"]) < marks[1]:
                    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            lane_ops = acc["SQ_INSTS_VALU"] * 64.0 / groups / (nodes * w.unit)
            sclk = (power or {}).get("sclk_mhz") or 2400
            ach_v = lane_ops * nodes * w.unit / (group_ms * 1e-3)
            peak_v = 256 * 64 * sclk * 1e6
            roof = out["roofline"]
            roof["valu"] = {"lane_ops_per_update": round(lane_ops, 1), "achieved": round(ach_v / 1e12, 2), "peak": round(peak_v / 1e12, 2),
                            "unit": "T lane-ops/s", "frac": round(ach_v / peak_v, 4), "sclk_mhz": sclk,
                            "sclk_source": "rocm-smi beside the same launches" if (power or {}).get("sclk_mhz") else "nominal 2400 MHz (no sample)",
                            "issue_frac_per_wave": round(acc["SQ_ACTIVE_INST_VALU"] / acc["SQ_WAVE_CYCLES"], 4),
                            "source": "rocprofv3 --pmc SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES pass of this run, every launch "
                                      "between two marker kernels; peak = 256 CUs x 64 f64 lanes per clock x sclk"}
            if roof.get("frac") is not None and roof["valu"]["frac"] > roof["frac"]:
                roof["bound"] = "valu"
                roof["bound_note"] = ("the f64 VALU issue fraction exceeds the HBM fraction: this kernel sits on the vector-ALU roof; "
                                      "achieved / peak / frac above stay the HBM figures, roofline.valu holds the other roof")
        except (subprocess.TimeoutExpired, OSError, KeyError, ValueError, ZeroDivisionError) as e:
            out["roofline"]["valu"] = {"error": f"SQ pass failed: {type(e).__name__}: {e}"}
    return out


def timed_batches(box, steps, repeats, ctl):
    """`repeats` batches of exactly `steps` steps, each bracketed by barrier + synchronize; returns
    per-repeat (wall seconds, device ms between HIP events on the launch stream), MAX over ranks."""
    torch = box.torch
    wall, devms, enq = [], [], []
    for _ in range(repeats):
        torch.cuda.synchronize()
        ctl.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        box.advance(steps)
        ev1.record()
        enq.append(time.perf_counter() - t0)    # host time to ENQUEUE the batch (GPU-bound runs: well below wall)
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
        ctl.barrier()
        torch.cuda.synchronize()
        devms.append(ev0.elapsed_time(ev1))
    return ctl.reduce(wall, "max"), ctl.reduce(devms, "max"), ctl.reduce(enq, "max")


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed batches of --steps steps (0 = auto: 25 for short batches, fewer for long ones, >= 5)")
    ap.add_argument("--min-warm-s", type=float, default=MIN_WARM_S,
                    help="seconds of untimed launches before the timed region, on top of --warmup (profiling passes set 0)")
    ap.add_argument("--rows", type=int, default=8192, help="rows PER GPU")
    ap.add_argument("--cols", type=int, default=8192)
    ap.add_argument("--omega", type=float, default=1.2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune", action="append", default=[], help="key=value for lbm_set_tuning")
    ap.add_argument("--plane-pad", type=int, default=None,
                    help="doubles of padding between planes (default: lbm_default_plane_pad)")
    ap.add_argument("--xn", type=int, default=5,
                    help="D >= 2 (default 5): register sliding-window kernel, D time steps per launch; 1: one step per launch")
    ap.add_argument("--sw-rows", type=int, default=-1,
                    help="rows per wavefront chunk of the sliding-window kernel (-1: fitted by the launcher to the resident wave slots)")
    ap.add_argument("--edge-rows", type=int, default=32, help="rows at each slab end computed ahead of the halo exchange")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure HBM traffic with rocprofv3 --pmc child runs (N = 1); report the committed profile's figure")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--ring-period", type=int, default=2,
                    help="slab ring: launches per halo exchange (ghost rows = period x D; 1: exchange every launch)")
    ap.add_argument("--force-halo", action="store_true",
                    help="N=1 only: run the slab schedule (ghost rows, self send/recv)")
    ap.add_argument("--transport", choices=["rccl", "ipc"], default="rccl",
                    help="slab ring transport: RCCL send/recv (default) or peer-mapped direct stores (hipIpc windows)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal: all N ranks on GPU 0 (peer-mapped transport, gloo control plane) -- checks the N-rank plumbing, not a scaling number")
    ap.add_argument("--launch-timeout", type=float, default=520.0,
                    help="N > 1: seconds after which a run without a result line is given up (exit 1 with the stalled rank's stderr)")
    ap.add_argument("--ring-deadline", type=float, default=120.0,
                    help="N > 1: seconds a worker has to bring its slab ring up before every rank restarts on the peer-mapped transport")
    ap.add_argument("--no-fallback", action="store_true", help="N > 1: no second attempt on the peer-mapped transport")
    ap.add_argument("--ctl", choices=["nccl", "gloo"], default="nccl", help="control plane of the workers (barriers, reductions)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--status-file", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--attempt", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--worker-script", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: skip configs 3 / 4 / 5 after the headline")
    ap.add_argument("--secondary", default="kbc,cg,ibm", help="which secondary workloads (N = 1)")
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi clock / power samples beside the workloads")
    ap.add_argument("--secondary-steps", type=int, default=30, help="time steps per timed batch of a secondary workload")
    ap.add_argument("--secondary-only", action="store_true", help="skip the headline (profiling passes)")
    return ap.parse_args(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    if a.gpus > 1 and not a.pmc_child and not a.worker:
        if "WORLD_SIZE" not in os.environ:
            sys.exit(self_launch(a, argv))
        sys.exit(supervise(a, argv))

    def stage(word):
        if a.status_file:
            with open(a.status_file, "a") as fh:
                fh.write(word + "\n")
    stage("start")
    stall = os.environ.get("LBM_BENCH_STALL")      # fault injection (tests): "rank:attempt" never brings its ring up
    if stall and a.worker and stall == f"{os.environ.get('RANK', '0')}:{a.attempt}":
        time.sleep(3600)

    import torch
    import pylbm
    from pylbm import _ptr

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: either launch N ranks (torch.distributed.run) or unset WORLD_SIZE "
                         "and let bench.py start them")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback on the product path)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible (one rank per GPU; "
                         "--share-gpu puts every rank on GPU 0 for a rehearsal)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctl = Ctl(rank, world, dev, "gloo" if (a.share_gpu or a.ctl == "gloo") else "nccl")

    lib = pylbm.Lib()
    lib.set_device(local_rank)
    tune = dict(kv.split("=") for kv in a.tune)
    for k, v in tune.items():
        lib.set_tuning(k.encode(), int(v))
    lib.set_tuning(b"sw_rows", a.sw_rows)

    if a.pmc_child and a.pmc_child != "headline":
        secondary_child(lib, dev, a.pmc_child)
        return
    if a.secondary_only:
        out = {"secondary": [run_secondary(lib, dev, w, a) for w in a.secondary.split(",") if w]}
        print(json.dumps(out), flush=True)
        return

    R, C = a.rows, a.cols
    transport = RING_IPC if (a.transport == "ipc" or a.share_gpu) else RING_RCCL
    box = Box(lib, a, ctl, dev, with_ring=(world > 1 or a.force_halo), transport=transport)
    D = box.depth
    f0 = taylor_green(lib, torch, _ptr, R, C, rank * R, world * R, dev)
    box.load(f0)
    del f0
    torch.cuda.synchronize()
    ctl.barrier()
    stage("ring_up")        # communicators initialised, ring created, first halo exchange done -- on every rank

    # -- slab ring: the one-exchange-per-`period`-launches schedule against one exchange per launch, here and now ----
    # (if the two ever differ on this machine, the timed run falls back to the plain schedule and says so)
    ring_check = None
    if box.ring and box.period > 1 and not a.pmc_child:
        ring_check = box.selfcheck(D)
        if "ring_period" in tune:
            lib.set_tuning(b"ring_period", int(tune["ring_period"]))
        if not ring_check:
            print("bench.py: ring schedules differ -- falling back to one exchange per launch", file=sys.stderr, flush=True)
            lib.set_tuning(b"ring_period", 1)
            box.period = 1

    if a.pmc_child:   # under rocprofv3 --pmc: a few launches of the dominant kernel, nothing else
        for _ in range(8):
            box.launch(D)
        torch.cuda.synchronize()
        box.close()
        return

    # -- warm-up: --warmup steps, then keep launching until MIN_WARM_S of GPU work has run ------
    box.advance(a.warmup)
    torch.cuda.synchronize()
    t_w, warm_steps = time.perf_counter(), a.warmup
    while True:
        box.advance(4 * D)
        torch.cuda.synchronize()
        warm_steps += 4 * D
        if ctl.reduce([float(time.perf_counter() - t_w >= a.min_warm_s)], "min")[0] > 0:   # all ranks leave the loop together
            break

    # -- timed region --------------------------------------------------------------------------
    repeats = a.repeats
    if repeats <= 0:
        w1, _, _ = timed_batches(box, a.steps, 1, ctl)        # pilot batch (also warm-up)
        repeats = max(5, min(25, int(2.5 / max(w1[0], 1e-6))))
    wall, devms, enq = timed_batches(box, a.steps, repeats, ctl)
    order = sorted(range(repeats), key=lambda i: wall[i])
    mid = order[repeats // 2]
    dt, dev_ms = wall[mid], devms[mid]

    stage("timed")
    mass = ctl.reduce([float(box.owned().sum())], "sum")[0]
    ring_status = int(ctl.reduce([float(lib.raw.lbm_ring_status(box.ring) != 0)], "max")[0]) if box.ring else 0

    # per-rank phase timing of one launch-step of the ring (outside the timed region)
    phases = None
    if box.ring:
        lib.ring_profile(box.ring, 1)
        acc = []
        for _ in range(5):
            for _ in range(box.period):   # every `period` consecutive launches hold one with an exchange: that one is timed
                box.launch(D)
            out4 = (ct.c_double * 4)()
            lib.ring_last_timing(box.ring, out4)
            acc.append(list(out4))
        lib.ring_profile(box.ring, 0)
        med = [statistics.median(x[i] for x in acc) for i in range(4)]
        phases = [dict(rank=i, edge_rows_ms=round(float(x[0]), 4), exchange_ms=round(float(x[1]), 4),
                       interior_ms=round(float(x[2]), 4), launch_span_ms=round(float(x[3]), 4))
                  for i, x in enumerate(ctl.gather(med))]

    # informative second figure (N = 1 only, outside the timed region): the same launch schedule
    # with the collision in the reference's exact operation order (GPU bitwise == CPU oracle)
    ref_order = None
    fast = tune.get("bgk_fast", "1") != "0"
    if world == 1 and D >= 2 and fast:
        lib.set_tuning(b"bgk_fast", 0)
        d_ref = 4 if (D == 5 and not box.ring) else D   # this collision's best depth (132.6 k vs 127.1 k MLUPS at 5)
        box.depth = d_ref
        box.advance(4 * d_ref)
        w2, _, _ = timed_batches(box, a.steps, 5, ctl)
        box.depth = D
        ref_order = {"value": round(R * C * a.steps / statistics.median(w2) / 1e6, 1), "unit": "MLUPS",
                     "steps": a.steps, "repeats": 5, "kernel": f"k_stream_collide_sw<BgkModelT<0,0>,{d_ref},4,nt>",
                     "steps_per_launch": d_ref,
                     "note": "same kernel family, collision in the reference's operation order (bitwise equal to the CPU oracle)"}
        lib.set_tuning(b"bgk_fast", int(tune.get("bgk_fast", "-1")))

    power = None
    if world == 1 and rank == 0 and not a.no_power:
        def _some():
            box.advance(20 * D)
            torch.cuda.synchronize()
        power = power_probe(_some)

    out = None
    if rank == 0:
        lups = R * C * world * a.steps / dt
        launches = box.launches(a.steps)
        kernel = ((f"k_stream_collide_sw<BgkFastModel,{D},2,nt>" if fast else f"k_stream_collide_sw<BgkModelT<0,0>,{D},4,nt>")
                  if D >= 2 else "k_stream_collide_v3<BgkModel,256,1,nt,nt>")
        # average duration of one launch of the dominant kernel (HIP events on the launch stream);
        # only meaningful when the batch holds that kernel alone
        kern_ms = dev_ms / launches if a.steps % D == 0 else None
        alg_bytes = R * C * BYTES_PER_LUP * D            # algorithmic bytes one launch stands for
        traffic, traffic_src, valu = None, None, None
        live, sq_live = None, None
        if world == 1 and not box.ring and D >= 2 and not a.no_pmc:
            tail = ["--rows", str(R), "--cols", str(C), "--omega", str(a.omega), "--xn", str(a.xn), "--sw-rows", str(a.sw_rows)]
            for kv in a.tune:
                tail += ["--tune", kv]
            if a.plane_pad is not None:
                tail += ["--plane-pad", str(a.plane_pad)]
            fb, wb_, note = pmc_traffic(tail)
            live = {"fetch_bytes": fb, "write_bytes": wb_, "note": note}
            sq_live = pmc_sq(tail)
        try:
            tj = json.load(open(TRAFFIC_FILE))
            ent = tj.get(kernel)
            if ent and (R, C) == tuple(ent.get("lattice", (8192, 8192))):
                traffic = ent["traffic_bytes"]
                traffic_src = f"committed profile, not this run ({ent.get('source', 'profiles/')})"
                valu = ent.get("valu_issue_frac")
        except (OSError, ValueError, KeyError):
            pass
        if live and live["fetch_bytes"]:
            traffic = live["fetch_bytes"] + live["write_bytes"]
            traffic_src = live["note"]
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel,
                "kernel_ms": round(kern_ms, 4) if kern_ms else None, "steps_per_launch": D,
                "algorithmic_bytes_per_launch": alg_bytes}
        if kern_ms:
            alg = alg_bytes / (kern_ms * 1e-3) / 1e9
            # with D steps fused per launch the 144 B/LUP figure is no lower bound on traffic any
            # more: this is a multiple of the single-step roofline, NOT a fraction of anything
            roof["algorithmic_GBs"] = round(alg, 1)
            roof["algorithmic_multiple"] = round(alg / HBM_PEAK_GBS, 4)
            if traffic:
                ach = traffic / (kern_ms * 1e-3) / 1e9
                roof["achieved"] = round(ach, 1)
                roof["frac"] = round(ach / HBM_PEAK_GBS, 4)
                roof["frac_of_copy_ceiling"] = round(ach / HBM_COPY_CEILING_GBS, 4)
                roof["traffic_bytes_per_update"] = round(traffic / (R * C * D), 2)
            elif D == 1:   # one step per launch: algorithmic bytes ARE the minimum traffic
                roof["achieved"], roof["frac"] = round(alg, 1), round(alg / HBM_PEAK_GBS, 4)
        if sq_live:
            roof["valu_issue_frac"] = sq_live["valu_issue_frac"]
            roof["valu_issue_frac_source"] = "rocprofv3 --pmc SQ pass of this run (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)"
            roof["waitcnt_frac"] = sq_live["waitcnt_frac"]
            roof["valu_lane_ops_per_update"] = round(sq_live["valu_wave_instructions_per_launch"] * 64.0 / (R * C * D), 1)
        elif valu is not None:
            roof["valu_issue_frac"] = valu
            roof["valu_issue_frac_source"] = "committed SQ pass (profiles/), not this run"
        if power:
            roof["power"] = power
        if live:
            roof["pmc"] = live
            roof["minimum_bytes_per_launch"] = R * C * BYTES_PER_LUP     # one read + one write of the lattice
        tname = {RING_RCCL: "RCCL send + recv", RING_IPC: "peer-mapped message (hipIpc window, direct stores)"}.get(box.transport)
        out = {
            "metric": "MLUPS (million lattice updates/sec), D2Q9 BGK periodic box, f64",
            "value": round(lups / 1e6, 1), "unit": "MLUPS", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "repeats": repeats, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{R}x{C} D2Q9 BGK periodic box per GPU, Taylor-Green init, "
                                   f"fused collide+stream (pull, two SoA lattices"
                                   f"{f', {D} time steps per launch (register sliding window)' if D >= 2 else ''}), omega={a.omega}",
                       "rows_per_gpu": R, "cols": C, "global_rows": R * world,
                       "plane_pad_doubles": box.plane - (R + 2 * box.ghost) * C,
                       "parallelism": f"slab{world}" if world > 1 else "single",
                       "transport": (f"lbm_ring (csrc/capi_ring.hip): one {tname} per neighbour per {box.period} launch(es) on the "
                                     "ring's own stream, interior rows on the caller's stream" if box.ring else None),
                       "halo": ("none" if not box.ghost else
                                f"{9 * (box.ghost - 1) if box.ghost > 1 else 3} rows of C doubles per side per "
                                f"{box.ghost} step(s) ({box.period} launch(es))")},
            "timing": {"protocol": f">= {a.min_warm_s} s of untimed launches after --warmup, then `repeats` batches of `steps` "
                                   "steps, each bracketed by barrier + synchronize, MAX over ranks; value = median batch",
                       "warm_steps_run": warm_steps,
                       "batch_ms": {"min": round(min(wall) * 1e3, 4), "median": round(dt * 1e3, 4), "max": round(max(wall) * 1e3, 4),
                                    "first": round(wall[0] * 1e3, 4), "last": round(wall[-1] * 1e3, 4)},
                       "host_enqueue_ms": round(statistics.median(enq) * 1e3, 4),
                       "timed_region_s": round(sum(wall), 4)},
            "roofline": roof,
            "check": {"total_mass": float(mass), "expected_mass": float(R * C * world),
                      **({"ring_schedules_agree_bitwise": ring_check} if ring_check is not None else {}),
                      **({"ring_status": "ok" if ring_status == 0 else "a bounded wait gave up"} if box.ring else {})},
        }
        if a.share_gpu:
            out["config"]["rehearsal"] = f"all {world} ranks on GPU 0: value is NOT a scaling number"
        if box.transport_note:
            out["config"]["transport_note"] = box.transport_note
        if phases:
            out["ring_phases"] = phases
        if ref_order:
            out["reference_order"] = ref_order
    box.close()
    del box
    torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not a.no_secondary:
            sec = []
            for w in [x for x in a.secondary.split(",") if x]:
                try:
                    sec.append(run_secondary(lib, dev, w, a))
                except Exception as e:  # a secondary figure must never cost the headline its line
                    sec.append({"config": w, "error": f"{type(e).__name__}: {e}"})
                torch.cuda.empty_cache()
            out["secondary"] = sec
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            for ent, w in zip(out.get("secondary", []), [x for x in a.secondary.split(",") if x]):
                try:
                    ent["cpu_baseline"] = cpu_secondary(w)
                    if ent.get("value") and ent["cpu_baseline"].get("value"):
                        ent["cpu_baseline"]["gpu_over_cpu"] = round(ent["value"] / ent["cpu_baseline"]["value"], 1)
                except Exception as e:  # a baseline must never cost the line
                    ent["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    ctl.close()
    stage("done")
    if box_failed(out, rank):
        sys.exit(3)


def box_failed(out, rank):
    """a ring whose bounded waits gave up has produced void numbers: say so with the exit code too"""
    return rank == 0 and out is not None and out.get("check", {}).get("ring_status", "ok") != "ok"


if __name__ == "__main__":
    main()
