"""GPU tests of the ROS-free controller node classes (include/ccv_mppi_node.hpp): one run() pass per call, closed loop
against the oracle, stage-wise vs fused call paths, command post-processing (publish_CmdVel / publish_CmdPos)."""
import numpy as np
import pytest

import helpers
import ccv_mppi_path_tracker_amd as amd
from ccv_mppi_path_tracker_amd import ControllerNode, capi, configs

pytestmark = pytest.mark.gpu
DEG = np.pi / 180.0


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(gpu_required):
    capi.load()


def _oracle_params(model, over):
    base = {"diff_drive": configs.diff_drive_defaults, "steering_diff_drive": configs.steering_defaults,
            "full_body": configs.full_body_defaults}[model](int(over["num_samples"]), int(over["horizon"]))
    kw = {}
    if "v_ref" in over:
        kw["v_ref"] = over["v_ref"]
    if "path_weight" in over:
        kw["path_weight"] = over["path_weight"]
    if "v_max" in over:
        kw["u_max"] = (over["v_max"],) + base.u_max[1:]
    for k in ("zmp_weight", "roll_v_weight", "back_weight", "yaw_weight"):
        if k in over:
            kw[k] = over[k]
    return base.with_(**kw)


CASES = [
    ("diff_drive", "sinusoid", {"num_samples": 512, "horizon": 30, "v_ref": 1.2, "v_max": 2.0, "path_weight": 10.0}),
    ("steering_diff_drive", "sinusoid", {"num_samples": 512, "horizon": 20, "v_ref": 1.2, "v_max": 2.0, "path_weight": 10.0}),
    ("full_body", "dkan", {"num_samples": 512, "horizon": 24, "v_ref": 2.0, "v_max": 2.0, "path_weight": 10.0,
                           "zmp_weight": 10.0, "roll_v_weight": 0.5, "yaw_weight": 2.0}),
]


@pytest.mark.parametrize("model,kind,over", CASES)
def test_node_closed_loop_matches_oracle(model, kind, over):
    p = _oracle_params(model, over)
    px, py = amd.make_path(kind)
    node = ControllerNode(model, over, seed=7)
    assert node.run_once(0.1) is None          # gated on the first path message (dd:338)
    node.set_path(px, py)
    o = helpers.oracle_for(p)
    s_n = np.zeros(p.nstate)
    s_n[:2] = px[0], py[0]
    s_o = s_n.copy()
    for it in range(12):
        dt = 0.1 if it % 3 else 0.093          # dt_ is the measured loop period (Q7)
        s_run = s_n.copy()
        node.set_state(s_n)
        out = node.run_once(dt)
        xr, yr, yaw = O_window(p, (px, py), s_o, dt)
        u_o = o.iterate(s_o, dt, xr, yr, yaw[0], seed=7, rng="philox", iteration=it)
        u_n = node.optimal_solution()
        assert helpers.rel_err(u_n, u_o) < 1e-8
        np.testing.assert_array_equal(node.ref_path()[:, 0], xr)
        assert out["cmd_vel"] == (u_n[0, 0], u_n[0, 1])           # dd:250-251
        o.set_nominal(u_n)
        s_n = amd.plant_step(model, s_n, u_n[0], dt)
        s_o = helpers.plant(model, s_o, u_n[0], dt)
    # optimal path = the optimal controls re-rolled through the plant model from the pose the iteration ran with, with its
    # dt_ (dd:295-312: predict_NextState(optimal_solution, i) fills state i+1, pose i of the message is state i -- the state
    # BEFORE step i; sd:328-345 the same with the steered heading; fb:332 has the call commented out, the mirror keeps it)
    op = node.optimal_path()
    assert op.shape == (p.horizon - 1, 3)
    s_roll = s_run.copy()
    want = np.zeros((p.horizon - 1, 3))
    for i in range(p.horizon - 1):
        want[i] = s_roll[:3]
        s_roll = helpers.plant(model, s_roll, u_n[i], dt)
    np.testing.assert_array_equal(op[0], s_run[:3])
    np.testing.assert_allclose(op, want, rtol=1e-12, atol=1e-12)   # (the mirror's specified sin / cos vs numpy's: a few ulp)
    assert np.max(np.abs(op[-1, :2] - op[0, :2])) > 0.1             # (it is a path, not H-1 copies of the pose)


@pytest.mark.parametrize("fused", [True, False])
def test_node_refuses_the_tick_when_the_window_is_refused(fused):
    """ccv_mppi_calc_ref_path refuses a negative or non-finite loop period and writes no window: the node must not iterate
    against the stale window and must publish no command (host-prologue paths: fused and stage-wise; the device-prologue
    path refuses in ccv_mppi_resident_step_enqueue).  dt_ = 0 is defined (stride 0, dd:160-163) and runs."""
    from ccv_mppi_path_tracker_amd.controller import MPPIError
    over = {"num_samples": 256, "horizon": 20}
    px, py = amd.make_path("sinusoid")
    node = ControllerNode("diff_drive", over, seed=5, fused=fused)
    node.set_path(px, py)
    node.set_state([px[3], py[3] + 0.05, 0.1])
    assert node.run_once(0.1) is not None
    u_before, win_before = node.optimal_solution(), node.ref_path()
    for bad in (-0.1, float("inf"), float("nan")):
        with pytest.raises(MPPIError) as e:
            node.run_once(bad)
        assert e.value.code == capi.ERR_INVALID_ARG
        np.testing.assert_array_equal(node.optimal_solution(), u_before)     # nothing ran, nothing was published
        np.testing.assert_array_equal(node.ref_path(), win_before)
    out = node.run_once(0.0)                                                   # two ticks inside one clock tick
    assert out is not None and np.all(np.isfinite(node.optimal_solution()))
    win = node.ref_path()
    assert np.all(win[:, 0] == win[0, 0]) and np.all(win[:, 1] == win[0, 1])   # stride 0: H copies of the nearest pose
    assert node.run_once(0.1) is not None                                      # and the node carries on


def O_window(p, path, state, dt):
    from oracle import oracle_lib as O
    _, xr, yr, yaw = O.calc_ref_path(path[0], path[1], state[0], state[1], p.v_ref, dt, p.resolution, p.horizon)
    return xr, yr, yaw


@pytest.mark.parametrize("model,kind,over", CASES)
def test_stagewise_and_fused_nodes_agree(model, kind, over):
    px, py = amd.make_path(kind)
    a, b = ControllerNode(model, over, seed=3, fused=True), ControllerNode(model, over, seed=3, fused=False)
    s = np.zeros(configs.NSTATE[model])
    s[:2] = px[0], py[0] + 0.1
    for n in (a, b):
        n.set_path(px, py)
    for it in range(4):
        for n in (a, b):
            n.set_state(s)
        ra, rb = a.run_once(0.1), b.run_once(0.1)
        np.testing.assert_array_equal(a.optimal_solution(), b.optimal_solution())
        assert ra == rb
        s = amd.plant_step(model, s, a.optimal_solution()[0], 0.1)


@pytest.mark.parametrize("model,kind,over", CASES)
def test_device_prologue_node_agrees_with_host_prologue_node(model, kind, over):
    """run_once() with get_CurrentIndex() + calc_RefPath() on the device (ccv_mppi_resident_*): same window, same u*, same
    commands; full body within rounding (yaw_ref[0] from the device atan2 enters fb:408)."""
    px, py = amd.make_path(kind)
    a, b = ControllerNode(model, over, seed=3), ControllerNode(model, over, seed=3, device_prologue=True)
    s = np.zeros(configs.NSTATE[model])
    s[:2] = px[0], py[0] + 0.1
    for n in (a, b):
        n.set_path(px, py)
    for it in range(5):
        for n in (a, b):
            n.set_state(s)
        ra, rb = a.run_once(0.1), b.run_once(0.1)
        np.testing.assert_array_equal(a.ref_path()[:, :2], b.ref_path()[:, :2])
        np.testing.assert_allclose(a.ref_path()[:-1, 2], b.ref_path()[:-1, 2], rtol=0, atol=1e-15)
        if model == "full_body":
            np.testing.assert_allclose(a.optimal_solution(), b.optimal_solution(), rtol=1e-9, atol=1e-12)
        else:
            np.testing.assert_array_equal(a.optimal_solution(), b.optimal_solution())
            assert ra == rb
        s = amd.plant_step(model, s, a.optimal_solution()[0], 0.1)
    if it == 4:   # a new path replaces the uploaded one
        for n in (a, b):
            n.set_path(px[::-1].copy(), py[::-1].copy())
            n.set_state(s)
        a.run_once(0.1), b.run_once(0.1)
        np.testing.assert_array_equal(a.ref_path()[:, :2], b.ref_path()[:, :2])


def test_cmd_pos_post_processing():
    tread = 0.501
    px, py = amd.make_path("sinusoid")
    # diff drive: steer 0, fore/rear = pitch_offset (dd:255-263)
    n = ControllerNode("diff_drive", {"num_samples": 256, "horizon": 10})
    n.set_path(px, py)
    n.set_state([0.0, 0.0, 0.0])
    out = n.run_once(0.1)
    assert out["cmd_pos"] == (0.0, 0.0, 3.0 * DEG, 3.0 * DEG, 0.0)
    # steering: inner/outer wheel angles (sd:275-291)
    n = ControllerNode("steering_diff_drive", {"num_samples": 256, "horizon": 10})
    n.set_path(px, py)
    n.set_state([0.0, 0.0, 0.0])
    out = n.run_once(0.1)
    v, w, st = n.optimal_solution()[0]
    R = abs(v / w)
    s_in = np.arctan2(R * np.sin(st), R * np.cos(st) - tread / 2)
    s_out = np.arctan2(R * np.sin(st), R * np.cos(st) + tread / 2)
    want = (s_in, s_out) if w > 0 else (s_out, s_in)
    np.testing.assert_allclose(out["cmd_pos"][:2], want, rtol=1e-14)
    # full body: roll command = roll + roll_v*dt clamped to +-30 deg, zero when roll_off (fb:266-269)
    n = ControllerNode("full_body", {"num_samples": 256, "horizon": 10})
    n.set_path(px, py)
    n.set_state([0.0, 0.0, 0.0, 29.9 * DEG, 0.0])
    out = n.run_once(0.1)
    rv = n.optimal_solution()[0, 3]
    assert abs(out["cmd_pos"][4] - min(max(29.9 * DEG + rv * 0.1, -30 * DEG), 30 * DEG)) < 1e-15
    n = ControllerNode("full_body", {"num_samples": 256, "horizon": 10, "roll_off": 1, "steer_off": 1})
    n.set_path(px, py)
    n.set_state([0.0, 0.0, 0.0, 0.1, 0.0])
    out = n.run_once(0.1)
    assert out["cmd_pos"][0] == 0.0 and out["cmd_pos"][1] == 0.0 and out["cmd_pos"][4] == 0.0
    assert np.all(n.optimal_solution()[:, 2] == 0.0)            # steer_off zeroes the direction samples (fb:517)


def _quat_from_rpy(roll, pitch, yaw):   # tf::Quaternion::setRPY, (x, y, z, w)
    cr, sr, cp, sp, cy, sy = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw / 2), np.sin(yaw / 2)
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])


def test_full_body_loop_closed_through_the_state_estimator():
    """the FullBodyMPPI mirror produces current_state_ itself.  The plant's pose reaches the node only as
    sensor messages -- the Gazebo model pose, an IMU orientation / rates / acceleration, six force sensors -- and
    run_once() does what run() does (fb:621-644): calc_true_ZMP(), get_CurrentState(), then the four hot methods.  The
    oracle controller gets the same state through the oracle's restatement of the estimator; both loops must agree, and
    the estimated roll / pitch must be the plant's."""
    from oracle import oracle_lib as O
    model, kind, over = CASES[2]
    p = _oracle_params(model, over)
    px, py = amd.make_path(kind)
    node = ControllerNode(model, over, seed=7)
    node.set_path(px, py)
    o, est = helpers.oracle_for(p), O.FbEstimator()
    s = np.zeros(p.nstate)
    s[:5] = px[0], py[0], 0.0, 0.02, -0.01
    u_prev = np.zeros(p.udim)
    for it in range(15):
        dt = 0.1
        # sensors as a simulator would publish them from the plant state and the last command
        q = _quat_from_rpy(s[3], s[4], s[2])
        rates = [u_prev[3], u_prev[4], u_prev[1]]                       # roll, pitch, yaw rate
        acc = [0.3 * np.cos(it), u_prev[0] * u_prev[1], 9.8]
        load = 60.0 * 9.8 / 6.0
        forces = [[0.0, 0.0, load * (1.0 + 0.2 * np.sin(0.3 * it + k))] for k in range(6)]
        node.fb_imu(q, rates, acc)
        est.imu(q, rates, acc)
        for k in range(6):
            node.fb_wrench(k, forces[k])
            est.wrench(k, forces[k])
        node.fb_pose(s[0], s[1], s[2])
        out = node.run_once(dt)                                          # calc_true_ZMP + get_CurrentState + iteration
        est.update(s[0], s[1], s[2], dt)
        e_o, e_n = est.read(), node.fb_read()
        np.testing.assert_array_equal(e_n, e_o)                          # estimator: host mirror == oracle restatement
        np.testing.assert_allclose(e_n[:5], s, atol=1e-14)               # and it recovers the plant's pose and attitude
        xr, yr, yaw = O_window(p, (px, py), e_o[:5], dt)
        u_o = o.iterate(e_o[:5], dt, xr, yr, yaw[0], seed=7, rng="philox", iteration=it)
        u_n = node.optimal_solution()
        assert helpers.rel_err(u_n, u_o) < 1e-8
        assert out["cmd_vel"] == (u_n[0, 0], u_n[0, 1])
        roll_cmd = min(max(e_n[3] + u_n[0, 3] * dt, -30 * DEG), 30 * DEG)   # fb:266-269 from the ESTIMATED roll
        assert abs(out["cmd_pos"][4] - roll_cmd) < 1e-15
        o.set_nominal(u_n)
        u_prev = u_n[0]
        s = amd.plant_step(model, s, u_n[0], dt)
    assert s[0] > px[0] + 1.0 and np.all(np.abs(e_n[7:9]) < 0.3)          # drove along the path; true ZMP inside the footprint
    # a node of another model has no estimator
    with pytest.raises(amd.controller.MPPIError):
        ControllerNode("diff_drive", CASES[0][2]).fb_pose(0, 0, 0)
