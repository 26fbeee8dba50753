"""Row-slab decomposition of one lattice over the GPUs of a node (one process per GPU).

Semantics = the reference's block binding (test/decompose_domain.cpp:181-187): after a step
the last row of block A must receive the populations moving towards -r ({3,6,7}) from the
first row of block B and vice versa ({1,5,8}); the +-1 column shift of the diagonal ones is
done by the pull kernel on the receiving side.  Here that is a one-row ghost layer per side:
every rank sends 3 rows of C doubles to each neighbour per step (24*C bytes per side).

Layout: torch tensor [9, R+2, C] (plane row 0 = ghost row -1, rows 1..R owned, row R+1 = ghost
row R), i.e. lbm_geom{R, C, ghost=1}; rows of one population are contiguous, so the halo rows
are sent and received in place -- no pack/unpack kernels.  torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests) does the transport; compute is
injected (`step_rows`) so the same ring logic is exercised on CPU by tests/test_slab_gloo.py.

Overlap: the two boundary rows are computed first; their halo messages then travel on the
process group's stream while the interior rows are computed on the caller's stream.
"""
import ctypes as ct

import torch
import torch.distributed as dist

from . import EDGE_HALO, EDGE_PERIODIC, Bc, Geom

TO_NEXT = (1, 5, 8)  # c_x = +1: leave through the last owned row
TO_PREV = (3, 6, 7)  # c_x = -1: leave through the first owned row


class SlabRing:
    def __init__(self, lib, R, C, rank, world, dev, periodic=True, bc=None, plane_pad=None,
                 force_ghost=False):
        self.lib, self.R, self.C, self.rank, self.world, self.dev = lib, R, C, rank, world, dev
        self.periodic = periodic
        # force_ghost: keep the ghost rows (and the self-exchange) even on one rank -- lets a
        # single GPU exercise and time the halo path
        self.ghost = 1 if (world > 1 or force_ghost) else 0
        rows = R + 2 * self.ghost
        # plane stride in doubles; a pad keeps the 9 planes off a common power-of-two stride
        if plane_pad is None:
            plane_pad = lib.default_plane_pad(rows, C) if lib is not None else 0
        self.plane = rows * C + plane_pad
        self.geom = Geom(R, C, self.ghost, self.plane if plane_pad else 0)
        self.bc = bc if bc is not None else Bc.periodic()
        if self.ghost:
            first, last = rank == 0, rank == world - 1
            if periodic or not first:
                self.bc.row_lo = EDGE_HALO
            if periodic or not last:
                self.bc.row_hi = EDGE_HALO
        self.buf = [torch.zeros(9 * self.plane, dtype=torch.float64, device=dev) for _ in range(2)]
        self.lat = [b.as_strided((9, rows, C), (self.plane, C, 1)) for b in self.buf]
        self.cur = 0
        self.side = None  # lazily created side stream (GPU only)
        self.schedule = 0
        self.next_rank = (rank + 1) % world if (periodic or rank < world - 1) else None
        self.prev_rank = (rank - 1) % world if (periodic or rank > 0) else None

    # -- plumbing ----------------------------------------------------------------------
    def stream_ptr(self):
        """hipStream_t of torch's current stream as a c_void_p (a bare Python int would be
        marshalled as a 32-bit int and truncate the handle)."""
        if self.dev.type == "cuda":
            return ct.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        return None

    def owned(self, lat=None):
        t = self.lat[self.cur] if lat is None else lat
        return t[:, self.ghost:self.ghost + self.R, :]

    def mass(self):
        return self.owned().sum()

    def exchange(self, lat):
        """Post the halo messages for `lat`; returns the outstanding requests."""
        if not self.ghost:
            return []
        R = self.R
        ops = []
        if self.next_rank is not None:
            ops += [dist.P2POp(dist.isend, lat[q, R], self.next_rank, tag=q) for q in TO_NEXT]
        if self.prev_rank is not None:
            ops += [dist.P2POp(dist.isend, lat[q, 1], self.prev_rank, tag=q) for q in TO_PREV]
        if self.prev_rank is not None:
            ops += [dist.P2POp(dist.irecv, lat[q, 0], self.prev_rank, tag=q) for q in TO_NEXT]
        if self.next_rank is not None:
            ops += [dist.P2POp(dist.irecv, lat[q, R + 1], self.next_rank, tag=q) for q in TO_PREV]
        return dist.batch_isend_irecv(ops) if ops else []

    # -- state ---------------------------------------------------------------------------
    def load_precollision(self, f_soa, collide):
        """f_soa: [9, R, C] pre-collision populations (the reference's f_adve).
        collide(dst, src, geom): P = collide(f) on a ghost-less [9,R,C] lattice."""
        assert tuple(f_soa.shape) == (9, self.R, self.C)
        flat = Geom(self.R, self.C, 0)
        p = torch.empty_like(f_soa)
        collide(p, f_soa.contiguous(), flat)
        self.owned(self.lat[self.cur]).copy_(p)
        for req in self.exchange(self.lat[self.cur]):
            req.wait()

    def step(self, step_rows):
        """One time step.  step_rows(dst, src, geom, bc, r0, r1) updates rows [r0, r1) on the
        CURRENT torch stream (callers pass stream_ptr() at call time)."""
        src, dst = self.lat[self.cur], self.lat[self.cur ^ 1]
        R = self.R
        if not self.ghost:
            step_rows(dst, src, self.geom, self.bc, 0, R)
        elif self.dev.type != "cuda":
            step_rows(dst, src, self.geom, self.bc, 0, 1)
            step_rows(dst, src, self.geom, self.bc, R - 1, R)
            reqs = self.exchange(dst)
            step_rows(dst, src, self.geom, self.bc, 1, R - 1)
            for req in reqs:
                req.wait()
        else:
            # Overlap schedule.  Stream E ("edge"): the two boundary rows, then the halo
            # messages.  Stream I: the interior rows, enqueued BEFORE the exchange is posted so
            # that neither the host-side cost of posting 12 messages nor RCCL's kernel leaves
            # the GPU idle (measured: a 134 us bubble per step otherwise).  Whether RCCL's
            # internal stream shares a hardware queue with torch's current stream or with our
            # side stream is not ours to choose, so both role assignments exist
            # (self.schedule 0: I = current, E = side; 1: I = side, E = current) and
            # autotune() picks the faster one during warm-up.
            cur = torch.cuda.current_stream(self.dev)
            if self.side is None:
                lo, hi = torch.cuda.Stream.priority_range()
                self.side = torch.cuda.Stream(self.dev, priority=hi)
            edge, inner = (self.side, cur) if self.schedule == 0 else (cur, self.side)
            edge.wait_stream(inner)         # previous step (other lattice) fully done
            inner.wait_stream(edge)
            with torch.cuda.stream(edge):
                step_rows(dst, src, self.geom, self.bc, 0, 1)
                step_rows(dst, src, self.geom, self.bc, R - 1, R)
            with torch.cuda.stream(inner):
                step_rows(dst, src, self.geom, self.bc, 1, R - 1)
            with torch.cuda.stream(edge):
                for req in self.exchange(dst):
                    req.wait()              # stream-level wait: edge now trails the transfers
            cur.wait_stream(self.side)      # callers synchronise on the current stream
        self.cur ^= 1

    def autotune(self, step_rows, steps=6):
        """Time both overlap schedules (GPU, ghost rows only) and keep the faster; all ranks
        agree through an all-reduce(MAX) of the timings.  Advances the state by 2*steps+2."""
        if not self.ghost or self.dev.type != "cuda":
            return self.schedule
        times = []
        for sched in (0, 1):
            self.schedule = sched
            self.step(step_rows)            # settle streams / lazy inits
            torch.cuda.synchronize(self.dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                self.step(step_rows)
            e1.record()
            torch.cuda.synchronize(self.dev)
            times.append(e0.elapsed_time(e1))
        t = torch.tensor(times, dtype=torch.float64, device=self.dev)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        self.schedule = int(t[1] < t[0])
        self.autotune_ms = [float(x) / steps for x in t]
        return self.schedule
