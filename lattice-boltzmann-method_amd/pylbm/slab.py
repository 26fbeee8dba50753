"""Row-slab decomposition of one lattice over the GPUs of a node (one process per GPU).

Semantics = the reference's block binding (test/decompose_domain.cpp:181-187): after a step
the last row of block A must receive the populations moving towards -r ({3,6,7}) from the
first row of block B and vice versa ({1,5,8}); the +-1 column shift of the diagonal ones is
done by the pull kernel on the receiving side.  Here that is a ghost layer per side:

  depth 1 (one step per launch):  ghost row -1 needs {1,5,8} of the previous slab's last row,
           ghost row R needs {3,6,7} of the next slab's first row: 3 rows of C doubles per side.
  depth 2 (two steps per launch, temporal blocking): the first of the two fused steps is also
           evaluated on ghost row -1 / R, so ghost row -1 needs {0,1,2,4,5,8} and ghost row -2
           {1,5,8} (mirror image below): 9 rows of C doubles per side per TWO steps.

Layout: torch tensor viewed as [9, R+2g, C] (g ghost rows above and below the R owned rows),
i.e. lbm_geom{R, C, ghost=g}; rows of one population are contiguous, so halo rows are sent and
received in place -- no pack/unpack kernels.  torch.distributed (backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in the CPU tests) does the transport; compute is injected (`step_rows`)
so the same ring logic is exercised on CPU by tests/test_slab_gloo.py.

Overlap: the edge rows are computed first; their halo messages then travel while the interior
rows are computed on another stream.
"""
import ctypes as ct

import torch
import torch.distributed as dist

from . import EDGE_ABB_VELOCITY, EDGE_BOUNCE_BACK, EDGE_HALO, EDGE_PERIODIC, EDGE_SPECULAR, Bc, Geom

TO_NEXT = (1, 5, 8)  # c_x = +1: leave through the last owned row
TO_PREV = (3, 6, 7)  # c_x = -1: leave through the first owned row
REST = (0, 2, 4)     # c_x = 0

# (populations, owned row counted from the sending edge) per ghost depth; the receiver stores
# them in the ghost row at the same distance from ITS edge.
ALL9 = tuple(range(9))


def _halo_table(depth, outward):
    """Which populations of the sender's k-th row from its edge (k = 0 .. depth-1) the
    neighbour's ghost row at the same distance must hold so that `depth` fused steps (or the
    colour-gradient step, depth 3) can be evaluated next to the seam.  The first fused step is
    evaluated on ghost rows up to distance depth-1; a row at distance k supplies its c_x = 0
    populations to itself, the outward-moving ones to the row at k-1 and the inward ones to
    k+1:  k <= depth-3: all 9;  k = depth-2: c_x = 0 and outward;  k = depth-1: outward only."""
    rows = []
    for k in range(depth):
        if k <= depth - 3:
            rows.append((ALL9, k))
        elif k == depth - 2:
            rows.append((REST + outward, k))
        else:
            rows.append((outward, k))
    return rows


class _HaloTables(dict):
    def __init__(self, outward):
        super().__init__()
        self.outward = outward

    def __missing__(self, depth):
        if isinstance(depth, int) and depth >= 100:   # LBM_HALO_FULL(d): all 9 populations of every row
            self[depth] = [(ALL9, k) for k in range(depth - 100)]
        else:
            self[depth] = _halo_table(depth, self.outward)
        return self[depth]


def _halo_table_two_phase(outward):
    """Colour-gradient step: 3 ghost rows as for a 3-step launch, but the second row travels
    complete -- the driver's same-row column copy (mrtcg_rayleigh_taylor.cpp:517-523)
    makes the column-0 / column-(C-1) nodes of ghost row 2 read {2,5,6} / {4,7,8} of their own
    row.  9 + 9 + 3 = 21 rows per colour per side (C ABI: LBM_HALO_TWO_PHASE)."""
    return [(ALL9, 0), (ALL9, 1), (outward, 2)]


TWO_PHASE = "two_phase"
HALO_TO_NEXT = _HaloTables(TO_NEXT)   # depth 1: [({1,5,8}, 0)]; depth 2: [({0,2,4,1,5,8}, 0), ({1,5,8}, 1)]; ...
HALO_TO_PREV = _HaloTables(TO_PREV)
HALO_TO_NEXT[TWO_PHASE] = _halo_table_two_phase(TO_NEXT)
HALO_TO_PREV[TWO_PHASE] = _halo_table_two_phase(TO_PREV)
# depth 3 = the colour-gradient step: pass A recomputes the macroscopic fields on ghost rows
# -2..-1 (R..R+1), whose own streaming reaches one row further out.


def halo_ops(lats, G, R, next_rank, prev_rank, table=None):
    """P2P ops that bring the ghost rows of every lattice in `lats` (views [9, R+2G, C]) up to
    date.  Sends are issued (to next, to prev), receives (from prev, from next): with two ranks
    both neighbours are the same peer and messages match in issue order."""
    ops = []
    T = G if table is None else table
    for f, lat in enumerate(lats):
        tag0 = 100 * f
        if next_rank is not None:   # my last rows -> their ghost rows above row 0
            for pops, k in HALO_TO_NEXT[T]:
                ops += [dist.P2POp(dist.isend, lat[q, G + R - 1 - k], next_rank, tag=tag0 + 10 * k + q) for q in pops]
        if prev_rank is not None:   # my first rows -> their ghost rows below row R-1
            for pops, k in HALO_TO_PREV[T]:
                ops += [dist.P2POp(dist.isend, lat[q, G + k], prev_rank, tag=tag0 + 10 * k + q) for q in pops]
    for f, lat in enumerate(lats):
        tag0 = 100 * f
        if prev_rank is not None:
            for pops, k in HALO_TO_NEXT[T]:
                ops += [dist.P2POp(dist.irecv, lat[q, G - 1 - k], prev_rank, tag=tag0 + 10 * k + q) for q in pops]
        if next_rank is not None:
            for pops, k in HALO_TO_PREV[T]:
                ops += [dist.P2POp(dist.irecv, lat[q, G + R + k], next_rank, tag=tag0 + 10 * k + q) for q in pops]
    return ops


class _PackedHalo:
    """request handle of a packed exchange: wait() = wait for the transfers (stream-level on the
    GPU), then scatter the received messages into the ghost rows on the current stream."""

    def __init__(self, ring, reqs):
        self.ring, self.reqs = ring, reqs

    def wait(self):
        for r in self.reqs:
            r.wait()
        self.ring._unpack()


class SlabRing:
    def __init__(self, lib, R, C, rank, world, dev, periodic=True, bc=None, plane_pad=None,
                 force_ghost=False, depth=1):
        """depth: ghost rows per side when the lattice is split (1, or 2 for two-step launches)."""
        self.lib, self.R, self.C, self.rank, self.world, self.dev = lib, R, C, rank, world, dev
        self.periodic = periodic
        # force_ghost: keep the ghost rows (and the self-exchange) even on one rank -- lets a
        # single GPU exercise and time the halo path
        self.ghost = depth if (world > 1 or force_ghost) else 0
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
        self.halo_buf = None
        # multi-step launches on slabs with wall columns / rows: every ghost row travels complete
        # (the fix-ups of a ghost-row wall node read that node's own populations): LBM_HALO_FULL(depth)
        walls = any(m in (EDGE_BOUNCE_BACK, EDGE_SPECULAR, EDGE_ABB_VELOCITY)
                    for m in (self.bc.row_lo, self.bc.row_hi, self.bc.col_lo, self.bc.col_hi))
        self.halo_code = (100 + self.ghost) if (walls and self.ghost > 1) else self.ghost
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
        """Post the halo messages for `lat`; returns the outstanding requests.  On the GPU the
        rows of one side travel as ONE packed message (lbm_halo_pack / _unpack; per-row messages
        measured ~2 ms of host time + a ~1 ms RCCL kernel per launch at depth 5); the CPU/gloo
        path sends the rows in place."""
        if not self.ghost:
            return []
        if self.lib is None or self.dev.type != "cuda":
            ops = halo_ops([lat], self.ghost, self.R, self.next_rank, self.prev_rank, table=self.halo_code)
            return dist.batch_isend_irecv(ops) if ops else []
        import ctypes
        from . import _ptr
        G = self.halo_code
        if self.halo_buf is None:
            n = self.lib.raw.lbm_halo_rows(G) * self.C
            self.halo_buf = {k: torch.empty(n, dtype=torch.float64, device=self.dev)
                             for k in ("send_next", "send_prev", "recv_prev", "recv_next")}
        hb, st, g = self.halo_buf, self.stream_ptr(), ctypes.byref(self.geom)
        ops = []
        if self.next_rank is not None:
            self.lib.halo_pack(_ptr(hb["send_next"]), _ptr(lat), g, G, 1, st)
            ops.append(dist.P2POp(dist.isend, hb["send_next"], self.next_rank))
        if self.prev_rank is not None:
            self.lib.halo_pack(_ptr(hb["send_prev"]), _ptr(lat), g, G, 0, st)
            ops.append(dist.P2POp(dist.isend, hb["send_prev"], self.prev_rank))
        if self.prev_rank is not None:
            ops.append(dist.P2POp(dist.irecv, hb["recv_prev"], self.prev_rank))
        if self.next_rank is not None:
            ops.append(dist.P2POp(dist.irecv, hb["recv_next"], self.next_rank))
        self._unpack_target = lat
        return [_PackedHalo(self, dist.batch_isend_irecv(ops))] if ops else []

    def _unpack(self):
        import ctypes
        from . import _ptr
        hb, st, g, lat = self.halo_buf, self.stream_ptr(), ctypes.byref(self.geom), self._unpack_target
        if self.prev_rank is not None:
            self.lib.halo_unpack(_ptr(lat), _ptr(hb["recv_prev"]), g, self.halo_code, 0, st)
        if self.next_rank is not None:
            self.lib.halo_unpack(_ptr(lat), _ptr(hb["recv_next"]), g, self.halo_code, 1, st)

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

    def step(self, step_rows, edge_rows=None):
        """One launch-step over the slab (one time step, or two with a two-step `step_rows`).
        step_rows(dst, src, geom, bc, r0, r1) updates rows [r0, r1) on the CURRENT torch stream
        (callers pass stream_ptr() at call time).  edge_rows: how many rows at each end are
        computed ahead of the exchange (default: the ghost depth; two-step kernels pass their
        tile height)."""
        src, dst = self.lat[self.cur], self.lat[self.cur ^ 1]
        R = self.R
        e = self.ghost if edge_rows is None else edge_rows
        if not self.ghost:
            step_rows(dst, src, self.geom, self.bc, 0, R)
        elif self.dev.type != "cuda":
            step_rows(dst, src, self.geom, self.bc, 0, e)
            step_rows(dst, src, self.geom, self.bc, R - e, R)
            reqs = self.exchange(dst)
            step_rows(dst, src, self.geom, self.bc, e, R - e)
            for req in reqs:
                req.wait()
        else:
            # Overlap schedule.  Stream E ("edge"): the boundary rows, then the halo messages.
            # Stream I: the interior rows, enqueued BEFORE the exchange is posted so that
            # neither the host-side cost of posting the messages nor RCCL's kernel leaves the
            # GPU idle (measured: a 134 us bubble per step otherwise).  Whether RCCL's internal
            # stream shares a hardware queue with torch's current stream or with our side
            # stream is not ours to choose, so both role assignments exist (self.schedule
            # 0: I = current, E = side; 1: I = side, E = current) and autotune() picks the
            # faster one during warm-up.
            cur = torch.cuda.current_stream(self.dev)
            if self.side is None:
                lo, hi = torch.cuda.Stream.priority_range()
                self.side = torch.cuda.Stream(self.dev, priority=hi)
            edge, inner = (self.side, cur) if self.schedule == 0 else (cur, self.side)
            edge.wait_stream(inner)         # previous step (other lattice) fully done
            inner.wait_stream(edge)
            with torch.cuda.stream(edge):
                step_rows(dst, src, self.geom, self.bc, 0, e)
                step_rows(dst, src, self.geom, self.bc, R - e, R)
            with torch.cuda.stream(inner):
                step_rows(dst, src, self.geom, self.bc, e, R - e)
            with torch.cuda.stream(edge):
                for req in self.exchange(dst):
                    req.wait()              # stream-level wait: edge now trails the transfers
            cur.wait_stream(self.side)      # callers synchronise on the current stream
        self.cur ^= 1

    def step_without_exchange(self, step_rows, keep):
        """A launch-step that uses up ghost rows instead of refreshing them (the schedule of capi_ring.hip's
        ring_bgk_step on slabs with ghost = m x D rows): ONE call over the owned rows plus `keep` ghost rows per
        side, which stay current for the launches that follow; the caller makes sure the ghost rows still
        current before the call are >= keep + the depth of `step_rows`."""
        src, dst = self.lat[self.cur], self.lat[self.cur ^ 1]
        assert 0 <= keep < self.ghost
        step_rows(dst, src, self.geom, self.bc, -keep, self.R + keep)
        self.cur ^= 1

    def autotune(self, step_rows, steps=6, edge_rows=None):
        """Time both overlap schedules (GPU, ghost rows only) and keep the faster; all ranks
        agree through an all-reduce(MAX) of the timings.  Advances the state by 2*steps+2
        launch-steps."""
        if not self.ghost or self.dev.type != "cuda":
            return self.schedule
        times = []
        for sched in (0, 1):
            self.schedule = sched
            self.step(step_rows, edge_rows)  # settle streams / lazy inits
            torch.cuda.synchronize(self.dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                self.step(step_rows, edge_rows)
            e1.record()
            torch.cuda.synchronize(self.dev)
            times.append(e0.elapsed_time(e1))
        t = torch.tensor(times, dtype=torch.float64, device=self.dev)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        self.schedule = int(t[1] < t[0])
        self.autotune_ms = [float(x) / max(steps, 1) for x in t]
        return self.schedule


class CgSlabRing:
    """Colour-gradient (two-phase) step over a CHAIN of row slabs (the driver's rows 0 / R-1 are
    bounce-back walls, test/mrtcg_rayleigh_taylor.cpp:525-531): two colour lattices with 3 ghost
    rows, macroscopic fields with 2.  Per step: pass A on rows -2..R+1 (where a neighbour exists),
    pass B on the 3 edge rows at each end, ONE exchange of 3 rows x both colours per side,
    pass B on the interior rows meanwhile."""
    G = 3

    def __init__(self, lib, R, C, rank, world, dev, params, plane_pad=None):
        from . import Bc
        import ctypes
        self.lib, self.R, self.C, self.rank, self.world, self.dev, self.params = lib, R, C, rank, world, dev, params
        G = self.G if world > 1 else 0
        self.ghost = G
        rows = R + 2 * G
        if plane_pad is None:
            plane_pad = lib.default_plane_pad(rows, C) if lib is not None else 0
        self.plane = rows * C + plane_pad
        self.geom = Geom(R, C, G, self.plane if plane_pad else 0)
        self.bc = Bc()
        if lib is not None:
            lib.raw.lbm_cg_default_bc(ctypes.byref(self.bc))
        if G:
            if rank > 0:
                self.bc.row_lo = EDGE_HALO
            if rank < world - 1:
                self.bc.row_hi = EDGE_HALO
        self.next_rank = rank + 1 if rank < world - 1 else None
        self.prev_rank = rank - 1 if rank > 0 else None
        z = lambda n: torch.zeros(n, dtype=torch.float64, device=dev)
        self.buf = [[z(9 * self.plane) for _ in range(2)] for _ in range(2)]           # [buffer][colour]
        self.lat = [[b.as_strided((9, rows, C), (self.plane, C, 1)) for b in bb] for bb in self.buf]
        mg = 2 if G else 0
        self.rho_r, self.rho_b, self.u = z((R + 2 * mg) * C), z((R + 2 * mg) * C), z(2 * (R + 2 * mg) * C)
        self.cur = 0
        self.side = None

    def exchange(self, lats):
        if not self.ghost:
            return []
        ops = halo_ops(lats, self.ghost, self.R, self.next_rank, self.prev_rank, table=TWO_PHASE)
        return dist.batch_isend_irecv(ops) if ops else []

    def step(self, moments, collide_rows):
        """moments(rho_r, rho_b, u, src_r, src_b, geom, bc); collide_rows(dst_r, dst_b, src_r, src_b,
        rho_r, rho_b, u, geom, bc, r0, r1) -- both enqueue on the current torch stream."""
        src, dst = self.lat[self.cur], self.lat[self.cur ^ 1]
        R, e = self.R, self.G
        moments(self.rho_r, self.rho_b, self.u, src[0], src[1], self.geom, self.bc)
        args = (dst[0], dst[1], src[0], src[1], self.rho_r, self.rho_b, self.u, self.geom, self.bc)
        if not self.ghost:
            collide_rows(*args, 0, R)
        elif self.dev.type != "cuda":
            collide_rows(*args, 0, e)
            collide_rows(*args, R - e, R)
            reqs = self.exchange(dst)
            collide_rows(*args, e, R - e)
            for req in reqs:
                req.wait()
        else:
            cur = torch.cuda.current_stream(self.dev)
            if self.side is None:
                self.side = torch.cuda.Stream(self.dev, priority=torch.cuda.Stream.priority_range()[1])
            side = self.side
            side.wait_stream(cur)           # pass A done, previous step done
            with torch.cuda.stream(side):   # interior on the side stream (see SlabRing.step)
                collide_rows(*args, e, R - e)
            collide_rows(*args, 0, e)
            collide_rows(*args, R - e, R)
            for req in self.exchange(dst):
                req.wait()
            cur.wait_stream(side)
        self.cur ^= 1
