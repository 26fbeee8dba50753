"""GPU suite: the C++ drivers (reference mains restated on the facade + C ABI) end to end."""
import os
import subprocess

import numpy as np
import pytest
from conftest import golden, relerr

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lattice-boltzmann-method_amd")
BIN = os.path.join(PKG, "drivers", "bin")


def run(name, *args, cwd=None, tune=None):
    exe = os.path.join(BIN, name)
    assert os.path.exists(exe), f"{exe} missing: run __graft_entry__.build()"
    env = dict(os.environ, LBM_TUNE=tune) if tune else None    # process-wide tuning of the library, through the environment
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, cwd=cwd, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return dict(ln.split("=", 1) for ln in r.stdout.splitlines() if "=" in ln and " " not in ln.split("=")[0])


def test_poiseuille_driver_reproduces_reference_assertion(tmp_path):
    g = golden("hpt_21x21.npz")
    out = run("horizontal_poiseuille_test", "--dump", tmp_path / "hpt")
    assert int(out["steps"]) == int(g["T"])                # no early exit, like the reference
    assert float(out["L2"]) <= 1e-11                       # horizontal_poiseuille_test.cpp:172
    assert abs(float(out["L2"]) - float(g["l2_printed"])) < 5e-14  # rounding-level agreement
    u = np.fromfile(tmp_path / "hpt-u.f64").reshape(21, 21, 2)
    ka = [0.009585127953269833, 0.07972021053814406, 0.10309857139975714, 0.079720210538144,
          0.00958512795326982]                             # SURVEY 8c known answers
    assert np.allclose(u[10, [0, 5, 10, 15, 20], 0], ka, rtol=1e-11, atol=0)


def test_shear_layer_driver_vs_reference_snapshot(tmp_path):
    g = golden("dsf_128.npz")
    k = list(g["snap_index"]).index(5)                      # snapshot 5 <-> 50 iterations
    run("ulbm_double_shear_flow", "--T", 50, "--dump", tmp_path / "dsf")
    u = np.fromfile(tmp_path / "dsf-u.f64").reshape(128, 128, 2)
    rho = np.fromfile(tmp_path / "dsf-rho.f64").reshape(128, 128)
    assert relerr(u[..., 0], g["ux"][..., k]) < 1e-11
    assert relerr(u[..., 1], g["uy"][..., k]) < 1e-11
    assert relerr(rho, g["rho"][..., k]) < 1e-12


def test_rayleigh_taylor_driver_vs_oracle(tmp_path, oracle):
    import pyoracle
    toml = open(os.path.join(PKG, "examples", "mrtcg-rayleigh-taylor-gamma3.toml")).read()
    toml = toml.replace("rows = 256", "rows = 64").replace("columns = 128", "columns = 32")
    (tmp_path / "rt.toml").write_text(toml)
    run("mrtcg_rayleigh_taylor", tmp_path / "rt.toml", "--steps", 40, "--dump", tmp_path / "rt")
    po = pyoracle.cg_params(64, 32)
    want = oracle.cg_steps(po, oracle.cg_init(po), 40)
    for name, key, shape in (("rho_r", "rho_r", (64, 32)), ("rho_b", "rho_b", (64, 32)), ("u", "u", (64, 32, 2)),
                             ("phase", "psi", (64, 32)), ("snu", "s_nu", (64, 32))):
        got = np.fromfile(tmp_path / f"rt-{name}.f64").reshape(shape)
        assert relerr(got, want[key]) < 1e-12, name


def test_cylinder_driver_vs_oracle(tmp_path, oracle):
    # a small lattice through the params.toml surface: l = 13 -> X = 8*13 = 104, Y = 6*13 = 78
    (tmp_path / "p.toml").write_text(
        "[flow]\ninitial_density = 1e3\nkinematic_viscosity = 1.0E-6\ncharacteristic_length = 2.6E-4\n"
        "characteristic_velocity = 0.2\n[lattice]\nrelaxation_time = 0.56\nlattice_spacing = 2.0E-5\n"
        "x_multiplier = 8\ny_multiplier = 6\n[simulation]\nstop_time = 1.0\nsnapshot_period = 1.0\n"
        'file_prefix = "t-"\n')
    n = 40
    t = 2 * np.pi * np.arange(n) / n
    x, y = 26.3 + 6.5 * np.cos(t), 39.2 + 6.5 * np.sin(t)
    (tmp_path / "b.toml").write_text('["cylinder-a"]\nx = [' + ", ".join(f"{v!r}" for v in x.tolist())
                                     + "]\ny = [" + ", ".join(f"{v!r}" for v in y.tolist()) + "]\n")
    # the reference's operation order ("bgk_fast_delta" = 0: what LBM_FORM_DEFAULT means is process-wide state) holds the
    # driver to the oracle at rounding level; the library's default -- the reassociated collision -- at 1e-9 (below)
    out = run("cylinder_test", tmp_path / "p.toml", tmp_path / "b.toml", "--steps", 25, "--dump", tmp_path / "c", tune="bgk_fast_delta=0")
    X, Y = 104, 78
    # lattice parameters as params::lattice derives them (src/params.cpp:52-65)
    Re = 0.2 * 2.6e-4 / 1.0e-6
    tau = 0.56
    u_in = Re * (1.0 / 3.0) * (tau - 0.5) / 13
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    fo, uo, rhoo, Fso = oracle.cylinder_steps(x, y, f0, 1.0 / tau, u_in, 25)
    f = np.fromfile(tmp_path / "c-f.f64").reshape(X, Y, 9)
    assert relerr(f, fo) < 1e-12
    assert np.allclose([float(out["Fs_r"]), float(out["Fs_c"])], Fso, rtol=1e-10, atol=1e-16)
    out = run("cylinder_test", tmp_path / "p.toml", tmp_path / "b.toml", "--steps", 25, "--dump", tmp_path / "d")
    f = np.fromfile(tmp_path / "d-f.f64").reshape(X, Y, 9)
    assert relerr(f, fo) < 1e-9
    assert np.allclose([float(out["Fs_r"]), float(out["Fs_c"])], Fso, rtol=1e-7, atol=1e-14)


def test_specular_boundary_driver_vs_reference(tmp_path):
    """C++ restatement of test/specular_boundary_test.cpp vs the unmodified main"""
    g = golden("sbt_51x51.npz")
    k = list(g["steps"]).index(1000)
    run("specular_boundary_test", "--T", 1000, "--dump", tmp_path / "sbt")
    f = np.fromfile(tmp_path / "sbt-f.f64").reshape(51, 51, 9)
    assert relerr(f, g["fs"][..., k]) < 1e-12


def test_gravity_driver_vs_reference(tmp_path):
    """C++ restatement of test/gravity_test.cpp: same early exit (t = 8301) and final state"""
    g = golden("gt_21x21.npz")
    out = run("gravity_test", "--dump", tmp_path / "gt")
    assert int(out["steps"]) == int(g["last_t"]) == 8301
    f = np.fromfile(tmp_path / "gt-f.f64").reshape(21, 21, 9)
    assert relerr(f, g["fs"][..., -1]) < 1e-11
    u = np.fromfile(tmp_path / "gt-u.f64").reshape(21, 21, 2)
    assert relerr(u[..., 0], g["ux"][..., -1]) < 1e-10


def test_free_stream_driver_vs_oracle(tmp_path, oracle):
    (tmp_path / "p.toml").write_text(
        "[flow]\ninitial_density = 1e3\nkinematic_viscosity = 1.0E-6\ncharacteristic_length = 2.2E-4\n"
        "characteristic_velocity = 0.2\n[lattice]\nrelaxation_time = 0.6\nlattice_spacing = 2.0E-5\n"
        "x_multiplier = 6\ny_multiplier = 4\n[simulation]\nstop_time = 1.0\nsnapshot_period = 1.0\n"
        'file_prefix = "t-"\n')
    run("free_stream_test", tmp_path / "p.toml", "--steps", 50, "--dump", tmp_path / "fs")
    X, Y = 66, 44                                           # l = 11 -> 6*11, 4*11
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = 0.1
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    fo, uo, rhoo = oracle.free_stream_steps(f0, 1.0 / 0.6, 0.1, 50)
    f = np.fromfile(tmp_path / "fs-f.f64").reshape(X, Y, 9)
    assert relerr(f, fo) < 1e-13


def test_static_droplet_driver_vs_oracle(tmp_path, oracle):
    """C++ restatement of test/mrtcg_static_droplet.cpp, run straight from the reference-style TOML"""
    import pyoracle
    toml = open(os.path.join(PKG, "examples", "mrtcg-rayleigh-taylor-gamma3.toml")).read()
    toml = toml.replace("rows = 256", "rows = 96").replace("columns = 128", "columns = 96")
    (tmp_path / "d.toml").write_text(toml)
    run("mrtcg_static_droplet", tmp_path / "d.toml", "--steps", 25, "--dump", tmp_path / "sd")
    po = pyoracle.cg_params(96, 96, sigma=0.1, gravity=0.0, gravity_c=-6.25e-6, add_source=0)
    want = oracle.cg_steps(po, oracle.cg_init_droplet(po), 25)
    for name, key, shape in (("rho_r", "rho_r", (96, 96)), ("rho_b", "rho_b", (96, 96)), ("u", "u", (96, 96, 2)),
                             ("phase", "psi", (96, 96))):
        got = np.fromfile(tmp_path / f"sd-{name}.f64").reshape(shape)
        assert relerr(got, want[key]) < 1e-12, name


def test_slab_ring_box_driver_self_ring(tmp_path):
    """C++ host of the multi-GPU path (drivers/slab_ring_box.cpp on lbm_ring_*): one forked rank whose
    ring neighbours are itself, so every launch-step runs the packed ncclSend/ncclRecv exchange on the
    edge stream; --check compares with the same box advanced as one ghost-free block, bit for bit."""
    import json
    exe = os.path.join(BIN, "slab_ring_box")
    assert os.path.exists(exe)
    for depth, edge, model, period in ((5, 16, "bgk", 2), (5, 16, "bgk", 1), (3, 16, "bgk", 3), (1, 8, "bgk", 2), (3, 16, "kbc", 2)):
        r = subprocess.run([exe, "--spawn", "1", "--rows", "160", "--cols", "256", "--steps", "3", "--model", model,
                            "--warmup", "1", "--depth", str(depth), "--edge-rows", str(edge), "--check", "1",
                            "--period", str(period), "--id-file", str(tmp_path / f"id{depth}{model}{period}")],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["check"] == "bitwise equal to one block" and line["depth"] == depth
        # window launches: ghost = period x depth rows, one exchange per `period` launches
        assert line["ghost_rows"] == (depth * period if depth > 1 else depth)


def test_slab_ring_rt_driver(tmp_path):
    """C++ host of config 4 over slabs (drivers/slab_ring_rt.cpp on lbm_ring_cg_step): with one rank the
    chain has no neighbour, so this checks the slab-geometry path (3 ghost rows, walls on the slab)
    against the ghost-free single block bit for bit; the exchange itself is covered by
    test_cg_fused_two_slabs_equal_single_block and the BGK self-ring tests."""
    import json
    exe = os.path.join(BIN, "slab_ring_rt")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "--spawn", "1", "--rows", "128", "--cols", "64", "--steps", "8", "--warmup", "2",
                        "--check", "1", "--id-file", str(tmp_path / "id")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["check"] == "bitwise equal to one block"


def test_slab_ring_rt_emulated_chain_of_four_slabs(tmp_path):
    """BASELINE config 4's decomposition in small (mrtcg_rayleigh_taylor.cpp:413-478 with its walls :495-533 on the
    outer slabs): 4 slabs in turn on one GPU, the two-colour 21-row messages by device copies, edge rows before
    interior rows; == the single block bit for bit"""
    import json
    exe = os.path.join(BIN, "slab_ring_rt")
    r = subprocess.run([exe, "--emulate", "4", "--rows", "64", "--cols", "96", "--steps", "9", "--warmup", "2", "--edge-rows", "8",
                        "--check", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["check"] == "bitwise equal to one block" and line["slabs"] == 4 and line["message_rows_per_colour_and_side"] == 21


def test_slab_ring_rt_with_row_padded_slabs(tmp_path):
    """1024 columns: slab_geom pads the rows of the slabs' lattices (lbm_default_row_pitch) -- init through
    lbm_lattice_copy_rows, ring pack / unpack, the two-part step and the check all on padded lattices; the emulated chain
    and two real rank processes, both bitwise against the dense single block"""
    import json
    exe = os.path.join(BIN, "slab_ring_rt")
    r = subprocess.run([exe, "--emulate", "3", "--rows", "48", "--cols", "1024", "--steps", "7", "--warmup", "2", "--edge-rows", "16",
                        "--check", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["check"] == "bitwise equal to one block"
    r = subprocess.run([exe, "--spawn", "2", "--transport", "ipc", "--one-gpu", "1", "--check", "1", "--id-file", str(tmp_path / "id"),
                        "--rows", "48", "--cols", "1024", "--steps", "6", "--warmup", "2", "--edge-rows", "16"],
                       capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["check"] == "bitwise equal to one block"


@pytest.mark.parametrize("driver,args", [
    ("slab_ring_box", ["--rows", "96", "--cols", "160", "--steps", "3", "--warmup", "1", "--depth", "5", "--period", "2", "--edge-rows", "16"]),
    ("slab_ring_box", ["--rows", "96", "--cols", "128", "--steps", "3", "--warmup", "1", "--depth", "3", "--model", "kbc", "--edge-rows", "16"]),
    ("slab_ring_rt", ["--rows", "64", "--cols", "96", "--steps", "6", "--warmup", "2", "--edge-rows", "8"]),
    ("slab_ring_cylinder", ["--rows", "128", "--cols", "160", "--diameter", "30", "--steps", "10", "--warmup", "5", "--edge-rows", "8"]),
])
def test_ring_drivers_with_three_real_ranks_on_one_gpu(tmp_path, driver, args):
    """the C++ hosts of the multi-GPU path as they run on a node -- `--spawn 3`: three forked rank processes --, here all on
    GPU 0 through the peer-mapped transport (--transport ipc --one-gpu 1); --check 1: rank 0 recomputes the domain as one
    block and compares every rank's rows bit for bit"""
    import json
    exe = os.path.join(BIN, driver)
    r = subprocess.run([exe, "--spawn", "3", "--transport", "ipc", "--one-gpu", "1", "--check", "1", "--id-file", str(tmp_path / "id")] + args,
                       capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["check"] == "bitwise equal to one block" and line["n_gpus"] == 3


def test_ring_driver_reports_a_neighbour_that_left(tmp_path):
    """the C++ ring hosts read lbm_ring_status after their timed launches.  `--desert 1`: rank 1 joins the ring and
    leaves; rank 0's bounded waits give up (1.5 s here), its lattices and timings are void -- it says so as JSON and the
    driver returns 4 instead of a rate"""
    import json
    exe = os.path.join(BIN, "slab_ring_box")
    env = dict(os.environ, LBM_TUNE="ring_ipc_timeout_ms=1500")
    r = subprocess.run([exe, "--spawn", "2", "--transport", "ipc", "--one-gpu", "1", "--desert", "1", "--rows", "96", "--cols", "160",
                        "--steps", "3", "--warmup", "1", "--depth", "5", "--period", "2", "--edge-rows", "16", "--id-file", str(tmp_path / "id")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 4, (r.returncode, r.stdout[-1000:], r.stderr[-1000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["rank"] == 0 and "never delivered" in line["error"] and "mlups" not in line


def test_slab_ring_cylinder_driver(tmp_path):
    """C++ host of config 5 over slabs (drivers/slab_ring_cylinder.cpp on lbm_ring_ibm_start /
    lbm_ring_bgk_block_ibm): 5-step blocks, the band around the ROI in a compact replica beside the far
    rows.  --check: populations and surface force equal the single-block solver-context run bit for bit.
    (One rank: the chain has no neighbour; the exchange is covered by the self-ring tests and by the
    emulated chain below.)"""
    import json
    exe = os.path.join(BIN, "slab_ring_cylinder")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "--spawn", "1", "--rows", "192", "--cols", "160", "--diameter", "30", "--steps", "12",
                        "--warmup", "3", "--edge-rows", "8", "--check", "1", "--id-file", str(tmp_path / "id")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["check"] == "bitwise equal to one block"
    assert line["markers"] == 94 and abs(line["Fs"][0]) > 0
    assert line["this_rank"]["owner"] == 1 and line["steps_per_block"] == 5


def test_slab_ring_cylinder_emulated_chain_with_the_cylinder_on_a_seam(tmp_path):
    """the BASELINE layout in small: 4 slabs, the cylinder centred at rows / 4 = exactly the seam between
    slabs 0 and 1 -- both co-own the band; every slab in turn on one GPU, messages by device copies;
    == the single block bit for bit (populations of every slab, surface force of both co-owners)"""
    import json
    exe = os.path.join(BIN, "slab_ring_cylinder")
    r = subprocess.run([exe, "--emulate", "4", "--uniform", "1", "--rows", "96", "--cols", "160", "--diameter", "30", "--steps", "15",
                        "--warmup", "5", "--check", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["check"] == "bitwise equal to one block"
    roles = [(s["owner"], s["straddle_prev"], s["straddle_next"]) for s in line["per_slab"]]
    assert roles == [(1, 0, 1), (1, 1, 0), (0, 0, 0), (0, 0, 0)], roles


def test_slab_ring_cylinder_emulated_chain_with_the_reassociated_collision(tmp_path):
    """--form reassociated (lbm_bgk_params.form on every slab and on the single block): the cylinder on the seam, both
    co-owners; slabs == the single block bit for bit in this operation order too"""
    import json
    exe = os.path.join(BIN, "slab_ring_cylinder")
    r = subprocess.run([exe, "--emulate", "4", "--uniform", "1", "--rows", "96", "--cols", "160", "--diameter", "30", "--steps", "15",
                        "--warmup", "5", "--check", "1", "--form", "reassociated"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["check"] == "bitwise equal to one block"
    r = subprocess.run([exe, "--emulate", "2", "--form", "fast"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "--form" in r.stderr


def test_slab_ring_cylinder_emulated_chain_with_planned_slab_heights(tmp_path):
    """the default: the LIBRARY cuts the domain (lbm_slab_ibm_plan_rows) -- the forced band lands inside ONE short slab,
    nobody straddles; == the single block bit for bit"""
    import json
    exe = os.path.join(BIN, "slab_ring_cylinder")
    r = subprocess.run([exe, "--emulate", "4", "--rows", "96", "--cols", "160", "--diameter", "30", "--steps", "15",
                        "--warmup", "5", "--check", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["check"] == "bitwise equal to one block" and line["slab_heights"].startswith("planned")
    heights = [s["rows"] for s in line["per_slab"]]
    assert sum(heights) == 384 and min(heights) >= 28
    roles = [(s["owner"], s["straddle_prev"], s["straddle_next"]) for s in line["per_slab"]]
    assert sum(o for o, _, _ in roles) == 1 and all(sp == 0 and sn == 0 for _, sp, sn in roles), roles
    owner = [s for s in line["per_slab"] if s["owner"]][0]
    assert owner["rows"] == min(heights)          # the band's slab is the short one


def test_slab_ring_cylinder_emulated_chain_with_unequal_slab_heights(tmp_path):
    """--slab-rows: each slab brings its own height (a chain runs at its slowest slab's pace, so the slabs that carry
    the forced band get fewer rows).  The band inside one short slab, and straddling two short ones; == the single
    block bit for bit."""
    import json
    exe = os.path.join(BIN, "slab_ring_cylinder")
    for heights, centre, roles_want in (("112,72,104,96", 148, [(0, 0, 0), (1, 0, 0), (0, 0, 0), (0, 0, 0)]),
                                        ("104,80,88,112", 184, [(0, 0, 0), (1, 0, 1), (1, 1, 0), (0, 0, 0)])):
        r = subprocess.run([exe, "--emulate", "4", "--slab-rows", heights, "--cols", "160", "--diameter", "22", "--centre-row", str(centre),
                            "--steps", "15", "--warmup", "5", "--check", "1"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["check"] == "bitwise equal to one block" and line["global_rows"] == 384
        assert [s["rows"] for s in line["per_slab"]] == [int(h) for h in heights.split(",")]
        roles = [(s["owner"], s["straddle_prev"], s["straddle_next"]) for s in line["per_slab"]]
        assert roles == roles_want, roles


def test_ulbm_poiseuille_driver_vs_oracle(tmp_path, oracle):
    """drivers/ulbm_poiseuille.cpp (KBC + pressure rows + bounce-back columns from the driver's zero
    start) at the reference's 128 x 128 for 300 iterations: moments as the driver holds them."""
    run("ulbm_poiseuille", "--T", 300, "--dump", tmp_path / "upo")
    nu = 1e-4
    s2 = 1.0 / (0.5 + 3.0 * nu)
    rin = 3.0 * 127 * (8.0 * nu * 0.05 / (128 * 128)) + 1.0
    _, m0, m1 = oracle.upo_steps(128, 128, s2, rin, 1.0, 300)
    u = np.fromfile(tmp_path / "upo-u.f64").reshape(128, 128, 2)
    rho = np.fromfile(tmp_path / "upo-rho.f64").reshape(128, 128)
    assert relerr(rho, m0) < 1e-14 and np.abs(u - m1).max() < 1e-15
    assert u[64, 64, 0] > 0           # the pressure drop drives the flow along +r


def test_decompose_domain_driver_vs_unmodified_main(tmp_path, oracle):
    """SURVEY a18, test/decompose_domain.cpp (the halo contract's specification): two blocks glued by
    cross-block pressure rows (lbm_pressure_row) and three populations per interface row each way
    (lbm_links_*), operator-level collide/advect.  Against the snapshots of the unmodified main
    (tests/golden/ddm_21x21.npz: adve_f, m_0, m_1 at the top of iterations 1, 2, 10, 100, 499), and
    bitwise against the oracle restatement (no reassociated path at the operator level)."""
    g = golden("ddm_21x21.npz")
    H = W = 21
    for k, t in enumerate(g["steps"]):
        out = run("decompose_domain", "--T", int(t), "--dump", tmp_path / "ddm")
        got = {}
        for blk in "AB":
            got["f" + blk] = np.fromfile(tmp_path / f"ddm-{blk}-f.f64").reshape(H, W, 9)
            got["u" + blk] = np.fromfile(tmp_path / f"ddm-{blk}-u.f64").reshape(H, W, 2)
            got["rho" + blk] = np.fromfile(tmp_path / f"ddm-{blk}-rho.f64").reshape(H, W)
            assert relerr(got["f" + blk], g[f"{blk}_fs"][..., k]) < 1e-13
            assert np.abs(got["u" + blk][..., 0] - g[f"{blk}_ux"][..., k]).max() < 1e-14
            assert np.abs(got["u" + blk][..., 1] - g[f"{blk}_uy"][..., k]).max() < 1e-14
            assert np.abs(got["rho" + blk] - g[f"{blk}_rho"][..., k]).max() < 1e-14
        o = oracle.ddm_run(H, W, int(t), float(out["omega"]), float(out["rho_inlet"]), 1.0)
        for key in ("fA", "fB", "uA", "uB", "rhoA", "rhoB"):
            assert np.array_equal(got[key], o[key]), key


@pytest.mark.parametrize("fast", [0, 1])
def test_decompose_domain_loop_driver_vs_oracle(tmp_path, oracle, fast):
    """SURVEY 8(f) row 4, test/decompose_domain_loop.cpp: four blocks closed into a loop channel by
    column-seam bindings, walls as slice assignments, momentum source on a row window of A --
    drivers/decompose_domain_loop.cpp on lbm_links_* (one gather launch for the ~70 slice assignments
    of a step).  L = 128, 300 steps: bitwise vs the oracle in the reference operation order; 1e-12
    with the default reassociated collision (blocks B, C, D; A's delta form always runs in order)."""
    env = dict(os.environ)
    exe = os.path.join(BIN, "decompose_domain_loop")
    assert os.path.exists(exe)
    # the tuning switch is process-wide state of the library: hand it over through the environment
    env["LBM_TUNE"] = f"bgk_fast={fast}"
    # fast = 1 also replays the step as a captured HIP graph (lbm_graph_*): same bits as plain launches
    r = subprocess.run([exe, "--L", "128", "--T", "300", "--graph", str(fast), "--dump", str(tmp_path / "ddl")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    want = oracle.ddl_run(128, 300)
    shapes = [(128, 32), (32, 64), (128, 32), (32, 64)]
    for k, name in enumerate("ABCD"):
        R, C = shapes[k]
        f = np.fromfile(tmp_path / f"ddl-{name}-f.f64").reshape(R, C, 9)
        rho = np.fromfile(tmp_path / f"ddl-{name}-rho.f64").reshape(R, C)
        u = np.fromfile(tmp_path / f"ddl-{name}-u.f64").reshape(R, C, 2)
        if fast:
            assert relerr(f, want["f"][k]) < 1e-12 and relerr(rho, want["rho"][k]) < 1e-12
            assert np.abs(u - want["u"][k]).max() < 1e-13
        else:
            assert np.array_equal(f, want["f"][k]), name
            assert np.array_equal(rho, want["rho"][k]) and np.array_equal(u, want["u"][k]), name
    assert want["u"][0][32 + 30, 16, 0] > 1e-3      # the source drives the flow along +r in A


def test_decompose_domain_loop_driver_vs_unmodified_main(tmp_path):
    """drivers/decompose_domain_loop.cpp at the reference's L = 512 against the snapshots of the
    unmodified main (tests/golden/ddl_512.npz): t = 50 ... 49950 iterations.  Default (reassociated)
    collision; 1e-12 absolute on u up to 500 iterations, 1e-9 at 5000 / 49950 (stationary state)."""
    g = golden("ddl_512.npz")
    L, L4 = 512, 128
    shapes = dict(A=(512, 128), B=(128, 256), C=(512, 128), D=(128, 256))
    cases = [("full", j, int(i)) for j, i in enumerate(g["full_index"])] + \
            [("strided", j, int(i)) for j, i in enumerate(g["strided_index"])]
    for kind, j, i in cases:
        T = 50 * i
        run("decompose_domain_loop", "--L", L, "--T", T, "--dump", tmp_path / "ddl")
        tol = 1e-12 if T <= 500 else 1e-9
        for blk, (R, C) in shapes.items():
            u = np.fromfile(tmp_path / f"ddl-{blk}-u.f64").reshape(R, C, 2)
            rho = np.fromfile(tmp_path / f"ddl-{blk}-rho.f64").reshape(R, C)
            ux = u[..., 0].copy()
            if blk == "A":
                ux[L4 + 5:L4 + 55] += 3e-3                      # :114, visible in the snapshot only
            uy = u[..., 1]
            if kind == "strided":
                ux, uy, rho = ux[::4, ::4], uy[::4, ::4], rho[::4, ::4]
            assert np.abs(ux - g[f"{blk}_ux_{kind}"][..., j]).max() < tol, (T, blk)
            assert np.abs(uy - g[f"{blk}_uy_{kind}"][..., j]).max() < tol, (T, blk)
            assert np.abs(rho - g[f"{blk}_rho_{kind}"][..., j]).max() < tol, (T, blk)


@pytest.mark.parametrize("fast", [0, 1])
def test_rectangle_sedimentation_driver_vs_oracle(tmp_path, oracle, fast):
    """SURVEY 8(f) row 4, test/rectangle_sedimentation_test.cpp: fluid + sediment distributions, anti-
    bounce-back columns with per-row wall terms, zero-gradient copies, the hard-coded obstacle --
    drivers/rectangle_sedimentation_test.cpp on lbm_links_* (affine links) vs the oracle (PARITY
    UNPINNED: the reference driver needs toml++).  540 x 420 lattice, 60 steps: bitwise in the
    reference operation order, 1e-12 with the default reassociated collision."""
    import json
    toml = open(os.path.join(PKG, "examples", "parameters.toml")).read()
    toml = toml.replace("characteristic_velocity = 0.5", "characteristic_velocity = 0.03")
    toml = toml.replace("lattice_spacing = 2.0E-5", "lattice_spacing = 1.0E-4")
    (tmp_path / "sed.toml").write_text(toml)
    r = subprocess.run([os.path.join(BIN, "params_dump"), str(tmp_path / "sed.toml")], capture_output=True, text=True)
    lp = json.loads(r.stdout)["lattice"]
    X, Y = lp["X"], lp["Y"]
    assert (X, Y) == (540, 420) and lp["u"] < 0.06
    env = dict(os.environ, LBM_TUNE=f"bgk_fast={fast}")
    steps = 60
    r = subprocess.run([os.path.join(BIN, "rectangle_sedimentation_test"), str(tmp_path / "sed.toml"), "--steps", str(steps),
                        "--dump", str(tmp_path / "sed")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    want = oracle.sed_steps(X, Y, lp["omega"], lp["u"], steps)
    got = dict(f=np.fromfile(tmp_path / "sed-f.f64").reshape(X, Y, 9), g=np.fromfile(tmp_path / "sed-g.f64").reshape(X, Y, 9),
               rho=np.fromfile(tmp_path / "sed-rho.f64").reshape(X, Y), u=np.fromfile(tmp_path / "sed-u.f64").reshape(X, Y, 2),
               C=np.fromfile(tmp_path / "sed-C.f64").reshape(X, Y))
    for k in ("f", "g", "rho", "u", "C"):
        if fast:
            assert relerr(got[k], want[k]) < 1e-12, (k, relerr(got[k], want[k]))
        else:
            assert np.array_equal(got[k], want[k]), (k, float(np.abs(got[k] - want[k]).max()))
    assert want["C"].max() > 1e-3 * 0.5 and np.isfinite(want["f"]).all()
