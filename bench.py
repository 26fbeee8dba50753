#!/usr/bin/env python3
"""Headline benchmark: MLUPS and % of the HBM roofline of the fused D2Q9 BGK collide+stream
step on a synthetic periodic box, 8192 x 8192 f64 nodes PER GPU (BASELINE.json configs[1]),
slab-decomposed along rows over N GPUs (weak scaling) with a halo exchange per launch.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over every node of the box (the default launch fuses 5
steps: `steps` counts time steps, not launches).  Rank 0 prints ONE JSON line.

Timing protocol: the GPU is first driven until >= 0.3 s of launches have run
(clock ramp; on top of --warmup), then the batch of EXACTLY --steps steps is timed `repeats` times
back to back, each repeat bracketed by barrier + synchronize on both sides with the MAX over
ranks taken per repeat; `ms_per_step` / `value` are those of the MEDIAN repeat.

N > 1: the transport is the library's own slab ring (csrc/capi_ring.hip: RCCL send/recv on packed
halo buffers on the ring's own high-priority stream, interior rows on the caller's stream);
torch.distributed only carries the 128-byte RCCL id, the barriers and the MAX reductions.

The timed region touches nothing under oracle/; the cpu_baseline leg (rank 0, N = 1 only) times
the unmodified reference (oracle/_ref) or, failing that, the CPU restatement on a bounded sample.
"""
import argparse
import ctypes as ct
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "lattice-boltzmann-method_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import pylbm  # noqa: E402
from pylbm import _ptr  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X spec (MI355X_MICROARCH.md:36); measured copy ceiling 6290
HBM_COPY_CEILING_GBS = 6290.0
BYTES_PER_LUP = 144.0        # 9 f64 reads + 9 f64 writes, SURVEY 8(d)
MIN_WARM_S = 0.3             # launches run before anything is timed, whatever --warmup says
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")


def taylor_green(lib, R_local, C, row0, R_global, dev, U=0.04):
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

    port = timed(orc.bgk_periodic_steps, orc.max_threads())
    port["kind"] = "port"
    out = port
    if Ref.available():
        try:
            ref = Ref()
            out = timed(ref.bgk_periodic_steps, ref.num_threads())
            out["kind"] = "reference"
            out["port"] = port
        except OSError as e:  # libtorch not loadable on this host
            out["note"] = f"oracle/_ref unusable: {e}"
    return out


class Box:
    """The per-rank slab: two SoA lattices with padded planes and (when split) D ghost rows, the
    launch that advances them, and -- with ghost rows -- the library's slab ring."""

    def __init__(self, lib, a, rank, world, dev, with_ring):
        self.lib, self.a, self.rank, self.world, self.dev = lib, a, rank, world, dev
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
        self.ring = None
        if with_ring:
            ident = (ct.c_ubyte * 128)()
            if rank == 0:
                lib.ring_unique_id(ident)
            if world > 1:   # torch.distributed carries the id, nothing else of the data path
                t = torch.tensor(list(ident), dtype=torch.uint8, device=dev)
                dist.broadcast(t, 0)
                ident = (ct.c_ubyte * 128)(*t.cpu().tolist())
            self.ring = ct.c_void_p()
            lib.ring_create(ct.byref(self.ring), ident, rank, world, ct.byref(self.geom), 1)

    def stream(self):
        return ct.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def owned(self):
        return self.lat[self.cur][:, self.ghost:self.ghost + self.R, :]

    def load(self, f_pre):
        """f_pre [9,R,C]: pre-collision populations; resident state = collide(f_pre) + ghost fill"""
        flat = pylbm.Geom(self.R, self.C, 0)
        p = torch.empty_like(f_pre)
        self.lib.bgk_collide(_ptr(p), _ptr(f_pre), ct.byref(flat), None, ct.byref(self.prm), None, None, self.stream())
        self.owned().copy_(p)
        if self.ring:
            self.lib.ring_exchange(self.ring, _ptr(self.lat[self.cur]), self.stream())
            self.lib.ring_join(self.ring, self.stream())

    def launch(self, n_steps):
        src, dst = self.lat[self.cur], self.lat[self.cur ^ 1]
        lib, g, bc, prm = self.lib, ct.byref(self.geom), ct.byref(self.bc), ct.byref(self.prm)
        if self.ring:
            lib.ring_bgk_step(self.ring, _ptr(dst), _ptr(src), None, prm, n_steps, self.a.edge_rows, self.stream())
        elif n_steps == 1:
            lib.bgk_stream_collide(_ptr(dst), _ptr(src), g, bc, prm, 0, self.R, None, None, self.stream())
        else:
            lib.bgk_stream_collide_xn(_ptr(dst), _ptr(src), g, bc, prm, n_steps, 0, self.R, self.stream())
        self.cur ^= 1

    def selfcheck(self, D, world):
        """Two launches with the ring's default schedule (ghost = period x D rows: the first without an exchange) and
        the same two with an exchange on every launch, from the same state: owned rows must agree bit for bit on
        every rank.  The state is restored afterwards."""
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
        ok = torch.tensor([float(torch.equal(first, self.owned()))], device=self.dev)
        self.lib.set_tuning(b"ring_period", -1)
        restore()
        torch.cuda.synchronize()
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        return bool(ok.item() > 0)

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


def pmc_traffic(argv_tail, kernel_tag="k_stream_collide_sw", timeout_s=150):
    """HBM bytes per launch of the dominant kernel, measured NOW: two child runs of this script under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE each in its own pass, as MI355X_MICROARCH.md prescribes;
    FETCH_SIZE x 2 = the gfx950 correction for wide coalesced reads, calibrated in profiles/).  Returns
    (fetch_bytes, write_bytes, note) or (None, None, reason).  The children only launch the kernel a few
    times (--pmc-child); nothing here is timed."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None, "rocprofv3 not found"
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="lbm_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child"] + argv_tail
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                               timeout=timeout_s)
            vals = []
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(path)):
                    if kernel_tag in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or len(vals) < 3:
                return None, None, f"{counter} pass failed (rc {r.returncode}, {len(vals)} samples)"
            out[counter] = statistics.median(vals[2:]) * 1024.0     # KiB; the first launches warm the caches
        except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
            return None, None, f"{counter} pass: {type(e).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return 2.0 * out["FETCH_SIZE"], out["WRITE_SIZE"], "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (FETCH_SIZE x 2: gfx950)"


def timed_batches(box, steps, repeats, world, dev):
    """`repeats` batches of exactly `steps` steps, each bracketed by barrier + synchronize; returns
    per-repeat (wall seconds, device ms between HIP events on the launch stream), MAX over ranks."""
    wall, devms, enq = [], [], []
    for _ in range(repeats):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        box.advance(steps)
        ev1.record()
        enq.append(time.perf_counter() - t0)    # host time to ENQUEUE the batch (GPU-bound runs: well below wall)
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        devms.append(ev0.elapsed_time(ev1))
    t = torch.tensor([wall, devms, enq], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t[0].tolist(), t[1].tolist(), t[2].tolist()


def main():
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
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--ring-period", type=int, default=2,
                    help="slab ring: launches per halo exchange (ghost rows = period x D; 1: exchange every launch)")
    ap.add_argument("--force-halo", action="store_true",
                    help="N=1 only: run the slab schedule (ghost rows, RCCL self send/recv every launch)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback on the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    lib = pylbm.Lib()
    lib.set_device(local_rank)
    tune = dict(kv.split("=") for kv in a.tune)
    for k, v in tune.items():
        lib.set_tuning(k.encode(), int(v))
    lib.set_tuning(b"sw_rows", a.sw_rows)

    R, C = a.rows, a.cols
    box = Box(lib, a, rank, world, dev, with_ring=(world > 1 or a.force_halo))
    D = box.depth
    f0 = taylor_green(lib, R, C, rank * R, world * R, dev)
    box.load(f0)
    del f0

    # -- slab ring: the one-exchange-per-`period`-launches schedule against one exchange per launch, here and now ----
    # (N > 1 never ran on hardware before the driver's own scaling run: if the two ever differ on this machine, the
    # timed run falls back to the plain schedule and says so)
    ring_check = None
    if box.ring and box.period > 1 and not a.pmc_child:
        ring_check = box.selfcheck(D, world)
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
        done = torch.tensor([float(time.perf_counter() - t_w >= a.min_warm_s)], device=dev)
        if world > 1:
            dist.all_reduce(done, op=dist.ReduceOp.MIN)   # all ranks leave the loop together
        if done.item() > 0:
            break

    # -- timed region --------------------------------------------------------------------------
    repeats = a.repeats
    if repeats <= 0:
        w1, _, _ = timed_batches(box, a.steps, 1, world, dev)        # pilot batch (also warm-up)
        repeats = max(5, min(25, int(2.5 / max(w1[0], 1e-6))))
    wall, devms, enq = timed_batches(box, a.steps, repeats, world, dev)
    order = sorted(range(repeats), key=lambda i: wall[i])
    mid = order[repeats // 2]
    dt, dev_ms = wall[mid], devms[mid]

    mass = box.owned().sum()
    if world > 1:
        dist.all_reduce(mass, op=dist.ReduceOp.SUM)

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
        t = torch.tensor(med, dtype=torch.float64, device=dev)
        if world > 1:
            g = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(g, t)
        else:
            g = [t]
        phases = [dict(rank=i, edge_rows_ms=round(float(x[0]), 4), exchange_ms=round(float(x[1]), 4),
                       interior_ms=round(float(x[2]), 4), launch_span_ms=round(float(x[3]), 4))
                  for i, x in enumerate(g)]

    # informative second figure (N = 1 only, outside the timed region): the same launch schedule
    # with the collision in the reference's exact operation order (GPU bitwise == CPU oracle)
    ref_order = None
    fast = tune.get("bgk_fast", "1") != "0"
    if world == 1 and D >= 2 and fast:
        lib.set_tuning(b"bgk_fast", 0)
        d_ref = 4 if (D == 5 and not box.ring) else D   # this collision's best depth (132.6 k vs 127.1 k MLUPS at 5)
        box.depth = d_ref
        box.advance(4 * d_ref)
        w2, _, _ = timed_batches(box, a.steps, 5, world, dev)
        box.depth = D
        ref_order = {"value": round(R * C * a.steps / statistics.median(w2) / 1e6, 1), "unit": "MLUPS",
                     "steps": a.steps, "repeats": 5, "kernel": f"k_stream_collide_sw<BgkModelT<0,0>,{d_ref},4,nt>",
                     "steps_per_launch": d_ref,
                     "note": "same kernel family, collision in the reference's operation order (bitwise equal to the CPU oracle)"}
        lib.set_tuning(b"bgk_fast", int(tune.get("bgk_fast", "-1")))

    if rank == 0:
        lups = R * C * world * a.steps / dt
        launches = box.launches(a.steps)
        kernel = ((f"k_stream_collide_sw<BgkFastModel,{D},2,nt>" if fast else f"k_stream_collide_sw<BgkModelT<0,0>,{D},4,nt>")
                  if D >= 2 else "k_stream_collide_v3<BgkModel,256,1,nt,nt>")
        # average duration of one launch of the dominant kernel (HIP events on the launch stream);
        # only meaningful when the batch holds that kernel alone
        kern_ms = dev_ms / launches if a.steps % D == 0 else None
        alg_bytes = R * C * BYTES_PER_LUP * D            # algorithmic bytes one launch stands for
        # HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes) cannot
        # be read from inside this process: the figure is the committed profile of this very kernel
        # on this very lattice, else null
        traffic, traffic_src, valu = None, None, None
        live = None
        if world == 1 and not box.ring and D >= 2 and not a.no_pmc:
            tail = ["--rows", str(R), "--cols", str(C), "--omega", str(a.omega), "--xn", str(a.xn), "--sw-rows", str(a.sw_rows)]
            for kv in a.tune:
                tail += ["--tune", kv]
            if a.plane_pad is not None:
                tail += ["--plane-pad", str(a.plane_pad)]
            fb, wb_, note = pmc_traffic(tail)
            live = {"fetch_bytes": fb, "write_bytes": wb_, "note": note}
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
        if valu is not None:
            roof["valu_issue_frac"] = valu
            roof["valu_issue_frac_source"] = "committed SQ pass (profiles/), not this run"
        if live:
            roof["pmc"] = live
            roof["minimum_bytes_per_launch"] = R * C * BYTES_PER_LUP     # one read + one write of the lattice
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
                       "transport": (f"lbm_ring (csrc/capi_ring.hip): one RCCL send + recv per neighbour per {box.period} launch(es) on the "
                                     "ring's own stream, interior rows on the caller's stream" if box.ring else None),
                       "halo": ("none" if not box.ghost else
                                f"{9 * (box.ghost - 1) if box.ghost > 1 else 3} rows of C doubles per side per "
                                f"{box.ghost} step(s) ({box.period} launch(es)) over RCCL send/recv")},
            "timing": {"protocol": f">= {a.min_warm_s} s of untimed launches after --warmup, then `repeats` batches of `steps` "
                                   "steps, each bracketed by barrier + synchronize, MAX over ranks; value = median batch",
                       "warm_steps_run": warm_steps,
                       "batch_ms": {"min": round(min(wall) * 1e3, 4), "median": round(dt * 1e3, 4), "max": round(max(wall) * 1e3, 4),
                                    "first": round(wall[0] * 1e3, 4), "last": round(wall[-1] * 1e3, 4)},
                       "host_enqueue_ms": round(statistics.median(enq) * 1e3, 4),
                       "timed_region_s": round(sum(wall), 4)},
            "roofline": roof,
            "check": {"total_mass": float(mass), "expected_mass": float(R * C * world),
                      **({"ring_schedules_agree_bitwise": ring_check} if ring_check is not None else {})},
        }
        if phases:
            out["ring_phases"] = phases
        if ref_order:
            out["reference_order"] = ref_order
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    box.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
