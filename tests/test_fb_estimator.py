"""CPU tests of the full-body state estimator: the host mirror of imuCallback / wrenchCallback /
calc_true_ZMP / get_CurrentState (src/full_body_mppi.cpp:115-156,199-237,528-596; csrc/host/mppi_node.cpp) against the
oracle's restatement, bit for bit, and both against independent numpy formulas.  Parity unpinned: the reference holds no
fixture for these functions, and their tf pieces (Matrix3x3::getRPY, Matrix3x3 * Vector3) are restated from tf's published
source (ros/geometry, noetic), which is absent from /root/reference."""
import numpy as np
import pytest

import ccv_mppi_path_tracker_amd as amd
from oracle import oracle_lib as O

MASS, L, ALPHA = 60.0, 0.8075 / 2, 0.3
IXX = MASS * (0.208**2 + 0.8075**2) / 12 + MASS * L * L
CONTACT = np.array([[0.0, 0.225, 0.075], [0.0, -0.225, 0.075], [0.245, 0.167, -0.003], [0.245, -0.167, -0.004],
                    [-0.245, -0.167, -0.004], [-0.245, 0.167, -0.003]])   # fb:57-63 in the order of fb:49-56


def quat_from_rpy(roll, pitch, yaw):
    """tf::Quaternion::setRPY convention: R = Rz(yaw) Ry(pitch) Rx(roll); returns (x, y, z, w)."""
    cr, sr, cp, sp, cy, sy = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw / 2), np.sin(yaw / 2)
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])


def rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]


def test_estimator_matches_the_oracle_bit_for_bit():
    """(6 000 ticks: with 200 the test once missed a last-place difference in the model ZMP that showed up about every 300th tick --
    std::sin / std::cos of one angle became one sincos call under g++ and two calls under clang; both sides call sincos now)"""
    rng = np.random.default_rng(11)
    host, orc = amd.FullBodyStateEstimator(), O.FbEstimator()
    for tick in range(6000):
        rpy = rng.uniform([-0.5, -0.3, -3.1], [0.5, 0.3, 3.1])
        q = quat_from_rpy(*rpy) * rng.uniform(0.5, 2.0)        # (tf normalises through s = 2 / |q|^2: any scale)
        w, a = rng.normal(0, 0.4, 3), rng.normal(0, 2.0, 3) + np.array([0, 0, 9.8])
        basis = rot("z", rng.normal(0, 0.1)) @ rot("y", rng.normal(0, 0.05))
        host.imu(q, w, a, basis)
        orc.imu(q, w, a, basis)
        for s in range(6):
            f = rng.normal(0, 30.0, 3) + np.array([0.0, 0.0, 100.0 if rng.random() < 0.8 else -50.0])   # some lift off
            b = rot("y", rng.normal(0, 0.3)) if s < 2 else None
            host.wrench(s, f, b)
            orc.wrench(s, f, b)
        x, y, yaw, dt = rng.normal(0, 3), rng.normal(0, 3), rng.uniform(-3, 3), rng.uniform(0.05, 0.2)
        assert host.update(x, y, yaw, dt) == orc.update(x, y, yaw, dt)
        np.testing.assert_array_equal(host.read(), orc.read())
    assert np.all(np.isfinite(host.read()))


def test_imu_angles_and_gravity_compensation():
    host = amd.FullBodyStateEstimator()
    for rpy in [(0.2, -0.1, 1.0), (-0.4, 0.25, -2.5), (0.0, 0.0, 0.0), (0.01, 1.2, 3.0)]:
        host.imu(quat_from_rpy(*rpy), [0, 0, 0], [0.3, -0.2, 9.8])
        out = host.read()
        np.testing.assert_allclose(out[10:13], rpy, atol=1e-14)                         # getRPY inverts setRPY
        assert out[13] == 0.3 - (-9.81) * np.sin(out[11]) and out[14] == -0.2 and out[15] == 9.8   # fb:233, g = -9.81 (fb.h:32)
    # gimbal lock: pitch = +-90 degrees -> yaw is reported as 0 and roll takes the whole rotation about the vertical
    for sgn in (1.0, -1.0):
        host.imu(quat_from_rpy(0.3, sgn * np.pi / 2, 0.0), [0, 0, 0], [0, 0, 0])
        out = host.read()
        assert abs(abs(out[11]) - np.pi / 2) < 1e-7 and (out[12] == 0.0 or abs(np.cos(out[11])) > 0)
    # the rotation into the robot frame: a 90 degree yaw of the IMU frame maps its x axis onto the robot's y axis
    host.imu(quat_from_rpy(0, 0, 0), [0, 0, 0], [1.0, 0.0, 0.0], rot("z", np.pi / 2))
    np.testing.assert_allclose(host.read()[13:16], [0.0, 1.0, 0.0], atol=1e-15)


def test_model_zmp_and_low_pass():
    """get_CurrentState(): ZMP from computeZMPfromModel (fb:597-603; closed form with accel.z = 0),
    H_Gdot = I_O (omega - omega_last) / dt with last_HG carried over, then alpha = 0.3 low-pass of zmp_x / zmp_y."""
    host = amd.FullBodyStateEstimator()
    zx = zy = 0.0
    w_last = np.zeros(3)
    rng = np.random.default_rng(3)
    for _ in range(30):
        roll, pitch = rng.uniform(-0.4, 0.4), rng.uniform(-0.2, 0.2)
        w, a, dt = rng.normal(0, 0.5, 3), rng.normal(0, 1.5, 3), 0.1
        host.imu(quat_from_rpy(roll, pitch, 0.7), w, a)
        host.update(1.0, 2.0, 0.7, dt)
        out = host.read()
        ax, ay = a[0] - (-9.81) * np.sin(out[11]), a[1]
        hx, hy = IXX * (w[0] - w_last[0]) / dt, IXX * (w[1] - w_last[1]) / dt      # Ixx == Iyy (width == depth)
        w_last = w
        com = np.array([L * np.sin(out[11]), -L * np.sin(out[10]), L * np.cos(out[11]) * np.cos(out[10])])
        mo = np.cross(com, [0, 0, MASS * -9.8]) - np.cross(com, [MASS * ax, MASS * ay, 0.0]) - np.array([hx, hy, 0.0])
        zmp = np.cross([0, 0, 1.0], mo) / (MASS * -9.8)
        zx, zy = ALPHA * zmp[0] + (1 - ALPHA) * zx, ALPHA * zmp[1] + (1 - ALPHA) * zy
        np.testing.assert_allclose(out[5:7], [zx, zy], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(out[:5], [1.0, 2.0, 0.7, out[10], out[11]], atol=0)
    # level and at rest: the ZMP relaxes to the origin
    for _ in range(60):
        host.imu(quat_from_rpy(0, 0, 0), [0, 0, 0], [0, 0, 0])
        host.update(0, 0, 0, 0.1)
    assert np.all(np.abs(host.read()[5:7]) < 1e-8)


def test_true_zmp_from_force_sensors():
    """calc_true_ZMP(): centre of pressure of the sensors in contact (f.z > 0), low-passed; 'denom is too small' keeps the
    previous value (fb:588-592)."""
    host = amd.FullBodyStateEstimator()
    forces = np.zeros((6, 3))
    forces[:, 2] = [200.0, 100.0, 50.0, 0.0, -20.0, 30.0]     # sensor 3: no load; sensor 4: pulls (ignored)
    for s in range(6):
        host.wrench(s, forces[s])
    assert host.update(0, 0, 0, 0.1) == 1
    act = forces[:, 2] > 0
    cop = (CONTACT[act, :2] * forces[act, 2:3]).sum(axis=0) / forces[act, 2].sum()
    np.testing.assert_allclose(host.read()[7:9], ALPHA * cop, rtol=1e-13)
    assert host.read()[9] == 0.0
    before = host.read()[7:10].copy()
    for s in range(6):
        host.wrench(s, [5.0, 5.0, -1.0])                      # nothing in contact
    assert host.update(0, 0, 0, 0.1) == 0
    np.testing.assert_array_equal(host.read()[7:10], before)
    for s in range(6):
        host.wrench(s, [0.0, 0.0, 1e-8])                      # in contact, but |sum f.z| < 1e-6
    assert host.update(0, 0, 0, 0.1) == 0
    np.testing.assert_array_equal(host.read()[7:10], before)
    # the wheel sensors (0, 1) arrive in the wheel frame and are rotated; the casters are not
    host2 = amd.FullBodyStateEstimator()
    host2.wrench(0, [10.0, 0.0, 0.0], rot("y", -np.pi / 2))   # wheel x axis points up
    host2.wrench(2, [10.0, 0.0, 0.0], rot("y", -np.pi / 2))   # caster: basis ignored -> no vertical load
    assert host2.update(0, 0, 0, 0.1) == 1
    np.testing.assert_allclose(host2.read()[7:9], ALPHA * CONTACT[0, :2], atol=1e-15)
    with pytest.raises(amd.controller.MPPIError):
        host2.wrench(6, [0, 0, 1])
