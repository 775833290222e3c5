"""Host prologue of the product (libccv_mppi_hip.so: window builder, path generators, plant) against the
oracle's restatement of the reference -- bit-exact, CPU only."""
import numpy as np
import pytest

import helpers
import ccv_mppi_path_tracker_amd as amd
from oracle import oracle_lib as O


@pytest.mark.parametrize("kind", ["straight", "sinusoid", "dkan"])
def test_path_generators_bit_exact(kind):
    px, py = amd.make_path(kind)
    ox, oy = helpers.oracle_path(kind)
    np.testing.assert_array_equal(px, ox)
    np.testing.assert_array_equal(py, oy)


@pytest.mark.parametrize("kind,v_ref,H", [("straight", 0.8, 15), ("sinusoid", 1.2, 50), ("dkan", 2.0, 80),
                                          ("sinusoid", 0.8, 30), ("dkan", 1.2, 128)])
def test_ref_window_bit_exact(kind, v_ref, H):
    px, py = amd.make_path(kind)
    rng = np.random.default_rng(1)
    poses = [(px[0], py[0]), (px[-1], py[-1]), (1e3, -1e3), (px[40] + 0.3, py[40] - 0.2)]
    poses += [(px[i] + dx, py[i] + dy) for i, dx, dy in zip(rng.integers(0, len(px), 40), rng.normal(0, 0.5, 40),
                                                          rng.normal(0, 0.5, 40))]
    for dt in (0.1, 0.093, 0.21):
        for x, y in poses:
            i1, xr1, yr1, yaw1 = amd.calc_ref_path(px, py, x, y, v_ref, dt, 0.1, H)
            i2, xr2, yr2, yaw2 = O.calc_ref_path(px, py, x, y, v_ref, dt, 0.1, H)
            assert i1 == i2
            np.testing.assert_array_equal(xr1, xr2)
            np.testing.assert_array_equal(yr1, yr2)
            np.testing.assert_array_equal(yaw1[:H - 1], yaw2[:H - 1])


def test_ref_window_rejects_bad_arguments():
    px, py = amd.make_path("straight")
    with pytest.raises(amd.controller.MPPIError):
        amd.calc_ref_path(px[:0], py[:0], 0.0, 0.0, 1.0, 0.1, 0.1, 10)
    # the window index start + i * v_ref * dt / resolution is the truncation of a double (dd:160-163) and the node takes dt
    # from its clock (dd:346-348): anything but a finite dt >= 0 and a finite non-negative stride would index before
    # path_[0] (undefined in the reference) and is refused
    for dt in (-0.1, float("inf"), float("nan")):
        with pytest.raises(amd.controller.MPPIError):
            amd.calc_ref_path(px, py, 0.0, 0.0, 1.0, dt, 0.1, 10)
    for v_ref, res in ((-1.0, 0.1), (1.0, -0.1), (float("nan"), 0.1), (1.0, 0.0), (1e308, 1e-308)):
        with pytest.raises(amd.controller.MPPIError):
            amd.calc_ref_path(px, py, 0.0, 0.0, v_ref, 0.1, res, 10)
    # dt = 0 (two ticks inside one clock tick) is defined in the reference -- stride 0 like v_ref = 0 -- and admitted
    i0, xr0, yr0, yaw0 = amd.calc_ref_path(px, py, 0.31, 0.0, 1.0, 0.0, 0.1, 10)
    j0, xq0, yq0, yawq0 = O.calc_ref_path(px, py, 0.31, 0.0, 1.0, 0.0, 0.1, 10)
    assert i0 == j0 and np.array_equal(xr0, xq0) and np.array_equal(yr0, yq0) and np.all(xr0 == xr0[0])
    idx, xr, yr, yaw = amd.calc_ref_path(px, py, 0.0, 0.0, 0.0, 0.1, 0.1, 10)   # v_ref = 0: stride 0, every point the same
    assert idx == 0 and np.all(xr == px[0])


@pytest.mark.parametrize("model,n", [("diff_drive", 3), ("steering_diff_drive", 3), ("full_body", 5)])
def test_plant_step_matches_predict_next_state(model, n):
    rng = np.random.default_rng(2)
    for _ in range(20):
        s, u = rng.normal(size=n), rng.normal(size=5)
        # sin / cos of the heading are the device's specified polynomial evaluation (<= 1 ulp from libm), so that the host
        # plant and the resident loop's plant (tests/test_gpu_resident.py) give the same bits
        np.testing.assert_allclose(amd.plant_step(model, s, u, 0.1), helpers.plant(model, s, u, 0.1), rtol=0, atol=1e-15)
    with pytest.raises(amd.controller.MPPIError):
        amd.plant_step(model, np.array([0.0, 0.0, 2.0e5, 0.0, 0.0][:n]), np.zeros(5), 0.1)   # outside the specified range
    # angles that leave +-1e4 rad are taken modulo 2 pi (the node reads them from tf in [-pi, pi]); below, untouched
    s = np.array([0.0, 0.0, 1.0e4 - 0.05, 0.0, 0.0][:n])
    u = np.array([0.0, 1.0, 0.0, 0.0, 0.0])
    s1 = amd.plant_step(model, s, u, 0.1)
    assert abs(s1[2]) <= np.pi and abs(np.sin(s1[2]) - np.sin(1.0e4 + 0.05)) < 1e-11 and abs(np.cos(s1[2]) - np.cos(1.0e4 + 0.05)) < 1e-11
    s2 = amd.plant_step(model, s, -u, 0.1)
    assert s2[2] == s[2] - 1.0 * 0.1
