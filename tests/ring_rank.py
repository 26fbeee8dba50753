"""One RANK of a slab ring, run as a process of its own by tests/test_gpu_ring_ranks.py:

    python tests/ring_rank.py <case> <rank> <nranks> <workdir>

<workdir> holds cfg.json, id.bin (the 128 bytes of lbm_ring_unique_id_ex) and the global input arrays as
.npy; the rank writes out_<rank>.npz (its owned rows, post-collision, SoA) and exits 0.  Several ranks
share GPU 0 through the peer-mapped transport (LBM_RING_IPC) -- the C++ ring of capi_ring.hip across REAL
process boundaries, which RCCL cannot do on one device.  Every compute call goes through the C ABI."""
import ctypes as ct
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, os.path.join(ROOT, "lattice-boltzmann-method_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    case, rank, n, work = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    cfg = json.load(open(os.path.join(work, "cfg.json")))
    import torch
    import pylbm
    from pylbm import _ptr

    lib = pylbm.Lib()
    lib.raw.lbm_slab_pressure_msg_doubles.restype = ct.c_longlong
    for k, v in cfg.get("tuning", {}).items():
        lib.set_tuning(k.encode(), int(v))
    d = torch.device("cuda:0")
    ident = (ct.c_ubyte * 128).from_buffer_copy(open(os.path.join(work, "id.bin"), "rb").read())
    transport = int(cfg.get("transport", 1))

    def load(name):
        return torch.from_numpy(np.load(os.path.join(work, name + ".npy"))).to(d)

    def struct(cls, key):
        return cls.from_buffer_copy(bytes.fromhex(cfg[key]))

    def make_ring(geom, periodic):
        ring = ct.c_void_p()
        lib.ring_create_ex(ct.byref(ring), ident, rank, n, ct.byref(geom), int(periodic), transport)
        assert lib.raw.lbm_ring_transport(ring) == transport
        assert lib.raw.lbm_ring_window_cached(ring) == int(cfg.get("expect_cached", 0))
        return ring

    def zeros(R, G, C):
        return torch.zeros((9, R + 2 * G, C), dtype=torch.float64, device=d)

    out = {}
    if case in ("bgk", "kbc"):
        R, C, G = cfg["R"], cfg["C"], cfg["ghost"]
        geom = pylbm.Geom(R, C, G)
        bc = struct(pylbm.Bc, "bc")
        walls = bool(cfg.get("walls", 0))
        prm = struct(pylbm.BgkParams if case == "bgk" else pylbm.KbcParams, "prm")
        p0 = load("p0")
        lat = [zeros(R, G, C), zeros(R, G, C)]
        lat[0][:, G:G + R] = p0[:, rank * R:(rank + 1) * R]
        ring = make_ring(geom, 1)
        if cfg.get("desert") and rank == 1:      # joins the ring, then leaves without a word
            os._exit(0)
        (lib.ring_exchange_full if walls else lib.ring_exchange)(ring, _ptr(lat[0]), None)
        lib.ring_join(ring, None)
        step = lib.ring_bgk_step if case == "bgk" else lib.ring_kbc_step
        cur = 0
        import time
        t0 = time.time()
        for depth in cfg["depths"]:
            step(ring, _ptr(lat[cur ^ 1]), _ptr(lat[cur]), ct.byref(bc), ct.byref(prm), int(depth), cfg.get("edge_rows", 16), None)
            cur ^= 1
        torch.cuda.synchronize()
        open(os.path.join(work, f"drain_{rank}.txt"), "w").write(repr(time.time() - t0))   # how long the queue took to drain
        lib.ring_status(ring)
        out["P"] = lat[cur][:, G:G + R].cpu().numpy()
        lib.ring_destroy(ring)
    elif case == "cg":
        R, C, G = cfg["R"], cfg["C"], 3
        geom = pylbm.Geom(R, C, G)
        pg = pylbm.cg_params()
        p0 = [load("p0r"), load("p0b")]
        lat = [[zeros(R, G, C), zeros(R, G, C)] for _ in range(2)]   # [buffer][colour]
        for k in range(2):
            lat[0][k][:, G:G + R] = p0[k][:, rank * R:(rank + 1) * R]
        ring = make_ring(geom, 0)            # a chain: bounce-back rows on the outer slabs (the driver's walls)
        lib.ring_exchange2(ring, _ptr(lat[0][0]), _ptr(lat[0][1]), None)
        lib.ring_join(ring, None)
        cur = 0
        for _ in range(cfg["steps"]):
            src, dst = lat[cur], lat[cur ^ 1]
            lib.ring_cg_step(ring, _ptr(dst[0]), _ptr(dst[1]), _ptr(src[0]), _ptr(src[1]), None, ct.byref(pg),
                             cfg.get("edge_rows", 16), None)
            cur ^= 1
        torch.cuda.synchronize()
        lib.ring_status(ring)
        out["Pr"] = lat[cur][0][:, G:G + R].cpu().numpy()
        out["Pb"] = lat[cur][1][:, G:G + R].cpu().numpy()
        lib.ring_destroy(ring)
    elif case == "ibm":
        X, Y, D = cfg["X"], cfg["Y"], cfg["D"]
        R = X // n
        geom = pylbm.Geom(R, Y, D)
        bc, prm = struct(pylbm.Bc, "bc"), struct(pylbm.BgkParams, "prm")
        x, y = np.load(os.path.join(work, "x.npy")), np.load(os.path.join(work, "y.npy"))
        f0 = load("f0")
        sl = pylbm.SlabIbm(lib, geom, rank * R, X, bc, prm, D, x, y)
        pre = zeros(R, D, Y)
        pre[:, D:R + D] = f0[:, rank * R:(rank + 1) * R]
        lat = [zeros(R, D, Y), zeros(R, D, Y)]
        ring = make_ring(geom, 0)
        lib.ring_ibm_start(ring, sl.h, _ptr(lat[0]), _ptr(pre), None)
        cur = 0
        for _ in range(cfg["blocks"]):
            lib.ring_bgk_block_ibm(ring, sl.h, _ptr(lat[cur ^ 1]), _ptr(lat[cur]), cfg.get("edge_rows", 16), None)
            cur ^= 1
        torch.cuda.synchronize()
        lib.ring_status(ring)
        out["P"] = lat[cur][:, D:R + D].cpu().numpy()
        out["roles"] = np.array([sl.owner, sl.straddle_prev, sl.straddle_next])
        out["Fs"] = sl.surface_force() if sl.owner else np.zeros(2)
        lib.ring_destroy(ring)
        sl.close()
    elif case == "pressure":
        R, W, D = cfg["R"], cfg["W"], cfg["D"]
        H = R * n
        geom = pylbm.Geom(R, W, D)
        bc, prm = struct(pylbm.Bc, "bc"), struct(pylbm.BgkParams, "prm")
        f0 = load("f0")
        h = ct.c_void_p()
        lib.slab_pressure_create(ct.byref(h), ct.byref(geom), rank * R, H, ct.byref(bc), ct.byref(prm), D)
        pre = zeros(R, D, W)
        pre[:, D:R + D] = f0[:, rank * R:(rank + 1) * R]
        lat = [zeros(R, D, W), zeros(R, D, W)]
        ring = make_ring(geom, 1)
        lib.ring_pressure_start(ring, h, _ptr(lat[0]), _ptr(pre), None)
        cur = 0
        for _ in range(cfg["blocks"]):
            lib.ring_bgk_block_pressure(ring, h, _ptr(lat[cur ^ 1]), _ptr(lat[cur]), None)
            cur ^= 1
        torch.cuda.synchronize()
        lib.ring_status(ring)
        out["P"] = lat[cur][:, D:R + D].cpu().numpy()
        lib.ring_destroy(ring)
        lib.slab_pressure_destroy(h)
    elif case == "pressure_kbc":
        R, W, D = cfg["R"], cfg["W"], 2
        H = R * n
        geom = pylbm.Geom(R, W, D)
        bc, prm = struct(pylbm.Bc, "bc"), struct(pylbm.KbcParams, "prm")
        m0, m1 = load("m0")[rank * R:(rank + 1) * R].contiguous(), load("m1")[:, rank * R:(rank + 1) * R].contiguous()
        h = ct.c_void_p()
        lib.slab_pressure_create_kbc(ct.byref(h), ct.byref(geom), rank * R, H, ct.byref(bc), ct.byref(prm))
        pre = zeros(R, D, W)              # the driver starts from adve_f = 0 (ulbm_poiseuille.cpp:85-86)
        lat = [zeros(R, D, W), zeros(R, D, W)]
        ring = make_ring(geom, 1)
        lib.ring_pressure_start_kbc(ring, h, _ptr(lat[0]), _ptr(pre), _ptr(m0), _ptr(m1), None)
        cur = 0
        for _ in range(cfg["blocks"]):
            lib.ring_bgk_block_pressure(ring, h, _ptr(lat[cur ^ 1]), _ptr(lat[cur]), None)
            cur ^= 1
        torch.cuda.synchronize()
        lib.ring_status(ring)
        out["P"] = lat[cur][:, D:R + D].cpu().numpy()
        lib.ring_destroy(ring)
        lib.slab_pressure_destroy(h)
    else:
        raise SystemExit(f"unknown case {case}")
    np.savez(os.path.join(work, f"out_{rank}.npz"), **out)


if __name__ == "__main__":
    main()
