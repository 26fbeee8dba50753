#!/usr/bin/env python3
"""Headline benchmark: MLUPS and % of the HBM roofline of the fused D2Q9 BGK collide+stream
step on a synthetic periodic box, 8192 x 8192 f64 nodes PER GPU (BASELINE.json configs[1]),
slab-decomposed along rows over N GPUs (weak scaling) with a one-row halo exchange.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path (lbm_bgk_stream_collide) over every node of the box.
Rank 0 prints ONE JSON line.  The timed region touches nothing under oracle/; the
cpu_baseline leg (rank 0, N = 1 only) times the unmodified reference (oracle/_ref) or,
failing that, the CPU restatement on a bounded sample.
"""
import argparse
import ctypes as ct
import json
import math
import os
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
from pylbm.slab import SlabRing  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X spec (MI355X_MICROARCH.md:36); measured copy ceiling 6290
HBM_COPY_CEILING_GBS = 6290.0
BYTES_PER_LUP = 144.0        # 9 f64 reads + 9 f64 writes, SURVEY 8(d)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--rows", type=int, default=8192, help="rows PER GPU")
    ap.add_argument("--cols", type=int, default=8192)
    ap.add_argument("--omega", type=float, default=1.2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune", action="append", default=[], help="key=value for lbm_set_tuning")
    ap.add_argument("--plane-pad", type=int, default=None,
                    help="doubles of padding between planes (default: lbm_default_plane_pad)")
    ap.add_argument("--x2", type=int, default=1,
                    help="1 (default): temporal blocking, two time steps per launch; 0: one step per launch")
    ap.add_argument("--tb-rows", type=int, default=8, help="tile height of the two-step kernel")
    ap.add_argument("--xn", type=int, default=5,
                    help="D >= 2 (default 5): register sliding-window kernel, D time steps per launch; "
                         "0: fall back to --x2 / single-step launches")
    ap.add_argument("--sw-rows", type=int, default=-1,
                    help="rows per wavefront chunk of the sliding-window kernel (-1: fitted by the launcher to the resident wave slots)")
    ap.add_argument("--edge-rows", type=int, default=32, help="rows at each slab end computed ahead of the halo exchange")
    ap.add_argument("--force-halo", action="store_true",
                    help="N=1 only: run the slab schedule (ghost rows, RCCL self send/recv)")
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
    use_dist = world > 1 or a.force_halo
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    lib = pylbm.Lib()
    lib.set_device(local_rank)
    for kv in a.tune:
        k, v = kv.split("=")
        lib.set_tuning(k.encode(), int(v))

    R, C = a.rows, a.cols
    prm = pylbm.BgkParams(a.omega, 0)
    use_xn = a.xn >= 2 and C >= 64 and R >= max(4 * a.xn + 8, 4 * a.edge_rows)
    use_x2 = (not use_xn) and bool(a.x2) and C % 64 == 0 and R % a.tb_rows == 0 and R >= 4 * a.tb_rows
    lib.set_tuning(b"tb_rows", a.tb_rows)
    lib.set_tuning(b"sw_rows", a.sw_rows)
    ring = SlabRing(lib, R, C, rank, world, dev, periodic=True, plane_pad=a.plane_pad,
                    force_ghost=a.force_halo, depth=a.xn if use_xn else (2 if use_x2 else 1))
    f0 = taylor_green(lib, R, C, rank * R, world * R, dev)
    ring.load_precollision(f0, lambda dst, src, geom: lib.bgk_collide(
        _ptr(dst), _ptr(src), ct.byref(geom), None, ct.byref(prm), None, None, ring.stream_ptr()))
    del f0

    def step_rows(dst, src, geom, bc, r0, r1):
        lib.bgk_stream_collide(_ptr(dst), _ptr(src), ct.byref(geom), ct.byref(bc), ct.byref(prm),
                               r0, r1, None, None, ring.stream_ptr())

    def step_rows_x2(dst, src, geom, bc, r0, r1):
        lib.bgk_stream_collide_x2(_ptr(dst), _ptr(src), ct.byref(geom), ct.byref(bc), ct.byref(prm),
                                  r0, r1, ring.stream_ptr())

    def step_rows_xn(dst, src, geom, bc, r0, r1):
        lib.bgk_stream_collide_xn(_ptr(dst), _ptr(src), ct.byref(geom), ct.byref(bc), ct.byref(prm),
                                  a.xn, r0, r1, ring.stream_ptr())

    def advance(n):
        """n time steps: pairs through the two-step kernel, a trailing odd one singly"""
        if use_xn:
            for _ in range(n // a.xn):
                ring.step(step_rows_xn, edge_rows=a.edge_rows)
            for _ in range(n % a.xn):
                ring.step(step_rows)
        elif use_x2:
            for _ in range(n // 2):
                ring.step(step_rows_x2, edge_rows=a.tb_rows)
            if n % 2:
                ring.step(step_rows)
        else:
            for _ in range(n):
                ring.step(step_rows)

    # picks the overlap schedule (no-op without ghost rows)
    if use_xn:
        ring.autotune(step_rows_xn, edge_rows=a.edge_rows)
    elif use_x2:
        ring.autotune(step_rows_x2, edge_rows=a.tb_rows)
    else:
        ring.autotune(step_rows)
    advance(a.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    advance(a.steps)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the launch stream

    t = torch.tensor([dt, dev_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, dev_ms = float(t[0]), float(t[1])
    mass = ring.mass()
    if world > 1:
        dist.all_reduce(mass, op=dist.ReduceOp.SUM)

    # informative second figure (N = 1 only, outside the timed region): the same launch schedule
    # with the collision in the reference's exact operation order (GPU bitwise == CPU oracle)
    ref_order = None
    tune_now = dict(kv.split("=") for kv in a.tune)
    if world == 1 and use_xn and tune_now.get("bgk_fast", "1") != "0":
        lib.set_tuning(b"bgk_fast", 0)
        advance(2 * a.xn)
        torch.cuda.synchronize()
        n_ref = max(a.xn, (a.steps // 2) // a.xn * a.xn)
        t1 = time.perf_counter()
        advance(n_ref)
        torch.cuda.synchronize()
        ref_order = {"value": round(R * C * n_ref / (time.perf_counter() - t1) / 1e6, 1), "unit": "MLUPS",
                     "steps": n_ref, "kernel": f"k_stream_collide_sw<BgkModelT<0,0>,{a.xn},4,nt>",
                     "note": "same schedule, collision in the reference's operation order (bitwise equal to the CPU oracle)"}
        lib.set_tuning(b"bgk_fast", int(tune_now.get("bgk_fast", "-1")))

    if rank == 0:
        lups = R * C * world * a.steps / dt
        steps_per_launch = a.xn if use_xn else (2 if use_x2 else 1)
        launches = a.steps // steps_per_launch + a.steps % steps_per_launch
        kern_ms = dev_ms / launches                      # avg duration of one launch
        alg_bytes = R * C * BYTES_PER_LUP * steps_per_launch   # algorithmic bytes one launch stands for
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        fast = dict(kv.split("=") for kv in a.tune).get("bgk_fast", "1") != "0"   # library default: 1
        kernel = ((f"k_stream_collide_sw<BgkFastModel,{a.xn},2,nt>" if fast else f"k_stream_collide_sw<BgkModel,{a.xn},4,nt>") if use_xn else
                  f"k_stream_collide_tb2<BgkModel,{a.tb_rows},512,nt>" if use_x2
                  else "k_stream_collide_v3<BgkModel,256,1,nt,nt>")
        # HBM bytes per launch cannot be read live (PMC counters need rocprofv3); report the figure
        # of the committed profile of this very kernel/config when there is one, else null
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if (R, C) == (8192, 8192) and kernel in tj:
                traffic, traffic_src = tj[kernel]["traffic_bytes"], "profiles/r01_traffic.json (rocprofv3 --pmc passes)"
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "MLUPS (million lattice updates/sec), D2Q9 BGK periodic box, f64",
            "value": round(lups / 1e6, 1), "unit": "MLUPS", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{R}x{C} D2Q9 BGK periodic box per GPU, Taylor-Green init, "
                                   f"fused collide+stream (pull, two SoA lattices"
                                   f"{f', {a.xn} time steps per launch (register sliding window)' if use_xn else ', 2 time steps per launch through an LDS tile' if use_x2 else ''}), omega={a.omega}",
                       "rows_per_gpu": R, "cols": C, "global_rows": R * world,
                       "plane_pad_doubles": ring.plane - (R + 2 * ring.ghost) * C,
                       "parallelism": f"slab{world}" if world > 1 else "single",
                       "halo": ("none" if not ring.ghost else
                                f"{9 * (ring.ghost - 1) if ring.ghost > 1 else 3} rows of C doubles per side per "
                                f"{ring.ghost} step(s) over RCCL send/recv"),
                       "overlap_schedule": ring.schedule if ring.ghost else None,
                       "schedule_ms": getattr(ring, "autotune_ms", None)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "frac_of_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel,
                         "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "steps_per_launch": steps_per_launch},
            "check": {"total_mass": float(mass), "expected_mass": float(R * C * world)},
        }
        if ref_order:
            out["reference_order"] = ref_order
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
