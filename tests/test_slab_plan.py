"""lbm_slab_ibm_plan_rows: slab heights for config 5 over a chain (cylinder_test.cpp:88-164 over the block binding of
decompose_domain.cpp:181-187).  Host arithmetic only: runs without a GPU."""
import ctypes as ct

import numpy as np
import pytest

import pylbm

_dp = ct.POINTER(ct.c_double)


@pytest.fixture(scope="module")
def lib():
    return pylbm.Lib()


def plan(lib, n, Rg, C, D, x, costs=None):
    rows, pred = (ct.c_int * n)(), ct.c_double()
    x = np.ascontiguousarray(x, dtype=np.float64)
    c = None if costs is None else (ct.c_double * 3)(*costs)
    lib.slab_ibm_plan_rows(rows, n, Rg, C, D, x.ctypes.data_as(_dp), len(x), c, ct.byref(pred))
    return list(rows), pred.value


def circle_rows(centre, diameter):
    m = int(round(np.pi * diameter))
    return centre + 0.37 + 0.5 * diameter * np.cos(2 * np.pi * np.arange(m) / m)


def test_baseline_geometry_puts_the_band_into_one_short_slab(lib):
    """BASELINE config 5: 16384 x 4096 over 8 slabs, diameter 300 at row 4096 (ON the seam of 8 equal slabs)"""
    x = circle_rows(4096, 300)
    rows, pred = plan(lib, 8, 16384, 4096, 5, x)
    assert sum(rows) == 16384 and min(rows) >= 28
    starts = np.cumsum([0] + rows)
    v0, v1 = int(np.floor(x.min())) - 2 - 5, int(np.floor(x.max())) + 3 + 5     # the band's valid rows
    k = int(np.searchsorted(starts, v0, side="right") - 1)
    assert starts[k] <= v0 and v1 <= starts[k + 1]                              # inside ONE slab
    assert rows[k] == min(rows) and rows[k] < 16384 // 8
    far = 0.180                                # the built-in table (capi_slab_ibm.hip, re-fitted at the end of round 3)
    uniform = 70.2 * 5 + far * 2048            # an equal cut leaves an owner with 2048 rows AND the chain
    assert pred < 0.8 * uniform
    assert max(r for i, r in enumerate(rows) if i != k) * far <= pred + 1e-9    # nobody slower than the prediction


@pytest.mark.parametrize("n,Rg,centre", [(2, 512, 100), (3, 900, 450), (4, 384, 96), (8, 4096, 3900), (5, 2000, 40)])
def test_plans_are_valid_layouts(lib, n, Rg, centre):
    x = circle_rows(centre, 30)
    rows, pred = plan(lib, n, Rg, 256, 5, x)
    assert len(rows) == n and sum(rows) == Rg and min(rows) >= 28 and pred > 0
    starts = np.cumsum([0] + rows)
    v0, v1 = int(np.floor(x.min())) - 7, int(np.floor(x.max())) + 8
    holders = [k for k in range(n) if starts[k] < v1 and starts[k + 1] > v0]
    assert len(holders) == 1, (rows, v0, v1)


def test_costs_move_the_cut(lib):
    """an owner whose chain costs nothing is just another slab; an expensive chain shrinks it to the band"""
    x = circle_rows(1000, 60)
    free, _ = plan(lib, 4, 4000, 512, 5, x, costs=(0.1, 0.0, 0.1))
    dear, _ = plan(lib, 4, 4000, 512, 5, x, costs=(0.1, 500.0, 0.03))
    assert max(free) <= 1020                   # nobody far above the equal share (heights move in steps of 4 rows)
    assert min(dear) <= 60 + 5 + 10 + 8 and min(dear) < min(free)


def test_one_slab_and_bad_arguments(lib):
    x = circle_rows(100, 30)
    assert plan(lib, 1, 300, 128, 5, x)[0] == [300]
    with pytest.raises(pylbm.LbmError):
        plan(lib, 8, 100, 128, 5, x)           # 100 rows cannot hold 8 slabs of 28
