"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Parity chain (the reference itself cannot be built or run here, see DESIGN.md "Oracle"):
  O->G   oracle in mt19937 mode (the reference's own generator) -> its clamped samples are INJECTED into the GPU
         (ccv_mppi_inject_controls) -> rollout / cost / weights / update must match the oracle:
         costs <= 1e-9 rel, u* <= 1e-5 rel (north_star tolerance; observed ~1e-12).
  R'->G  oracle in philox mode vs GPU philox: noise bit-exact, u* <= 1e-5 rel (observed ~1e-10).
Full BASELINE sizes are covered through size-independent properties (contiguous sample blocks re-scored by the
oracle, host recomputation of the weighted mean, determinism, shard invariance).
"""
import os

import numpy as np
import pytest

import helpers
import ccv_mppi_path_tracker_amd as amd
from ccv_mppi_path_tracker_amd import capi, configs
from ccv_mppi_path_tracker_amd.controller import MPPIController, MPPIError

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {name: (p, kind) for name, p, kind in helpers.small_cases()}
TOL_U = 1e-5      # north_star: controls within 1e-5 relative of the reference CPU loop
TOL_COST = 1e-9   # SURVEY.md 8c O->G


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(gpu_required):
    capi.load()


def start_state(p, path, lateral=0.0):
    s = np.zeros(p.nstate)
    s[0], s[1] = path[0][0], path[1][0] + lateral
    return s


# --------------------------------------------------------------------------------------------------------------
# noise: GPU Philox + Box-Muller == CPU restatement, bit for bit
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model", ["diff_drive", "steering_diff_drive", "full_body"])
def test_noise_bit_exact(model):
    mk = {"diff_drive": configs.diff_drive_defaults, "steering_diff_drive": configs.steering_defaults,
          "full_body": configs.full_body_defaults}[model]
    K, H, off = 1000, 50, 123457
    p = mk(K, H).with_(control_noise=1.0, u_min=tuple([-1e30] * configs.UDIM[model]),
                       u_max=tuple([1e30] * configs.UDIM[model]))
    g = MPPIController(p, sample_offset=off)
    seed, it = 0x0123456789ABCDEF, (1 << 33) + 5
    g.sampling(seed, it)
    got = g.read_controls()
    o = helpers.oracle_for(p)
    o.sampling(seed, rng="philox", iteration=it, k_offset=off)
    np.testing.assert_array_equal(got, o.get_controls())
    z = got.ravel()
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.02


def test_sampling_uses_nominal_sigma_and_clamp():
    p = configs.workload("C3").params.with_(num_samples=513, horizon=20)
    g, o = MPPIController(p), helpers.oracle_for(p)
    nominal = np.random.default_rng(0).normal(0, 0.4, size=(p.horizon - 1, p.udim))
    g.set_nominal(nominal)
    o.set_nominal(nominal)
    np.testing.assert_array_equal(g.get_nominal(), nominal)
    g.sampling(5, 9)
    o.sampling(5, rng="philox", iteration=9)
    u = g.read_controls()
    np.testing.assert_array_equal(u, o.get_controls())
    for d in range(p.udim):
        assert u[..., d].min() >= p.u_min[d] and u[..., d].max() <= p.u_max[d]
        assert (u[..., d] == p.u_max[d]).any() or (u[..., d] == p.u_min[d]).any()


# --------------------------------------------------------------------------------------------------------------
# O->G: injected reference-generator samples, every small case, three consecutive closed-loop iterations
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(CASES))
def test_injected_controls_match_oracle(name):
    p, kind = CASES[name]
    path = helpers.oracle_path(kind)
    o, g = helpers.oracle_for(p), MPPIController(p)
    state = start_state(p, path)
    for it in range(3):
        xr, yr, yaw = helpers.oracle_window(p, path, state)
        u_in = o.get_nominal()
        g.set_nominal(u_in)
        # reference call order (dd:352-358)
        o.sampling(1000 + it, rng="mt19937")
        g.inject_controls(o.get_controls())
        o.predict_States(state, p.dt)
        g.predict_States(state, p.dt)
        o.calc_Weights(xr, yr, yaw[0])
        g.calc_Weights(xr, yr, yaw[0])
        u_o = o.determine_OptimalSolution()
        u_g, st = g.determine_OptimalSolution(want_stats=True)

        xy = g.read_candidates()
        np.testing.assert_allclose(xy[..., 0], o.states("x"), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(xy[..., 1], o.states("y"), rtol=1e-12, atol=1e-13)
        c_o, c_g = o.costs(), g.read_costs()
        assert np.max(np.abs(c_g - c_o) / np.abs(c_o)) < TOL_COST
        assert abs(st.sum_w - o.sum_w()) <= 1e-9 * o.sum_w()
        np.testing.assert_allclose(g.read_weights(), o.weights(), rtol=1e-8, atol=1e-300)
        assert st.min_cost == c_g.min() and st.max_cost == c_g.max()
        assert st.n_zero_weight == int((np.exp(-c_g / p.lam) == 0).sum()) and st.nonfinite == 0
        assert helpers.rel_err(u_g, u_o) < TOL_U
        assert helpers.rel_err(u_g, u_o) < 1e-9          # what the fp64 path actually delivers
        np.testing.assert_array_equal(g.get_nominal(), u_g)
        state = helpers.plant(p.model, state, u_o[0], p.dt)


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_vectors_through_the_gpu(name):
    """The committed fixtures (inputs + expected outputs) replayed through the HIP path."""
    p, kind = CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    o, g = helpers.oracle_for(p), MPPIController(p)
    sel = [0, 1, p.num_samples - 1]
    for it in range(3):
        pre = "it%d_" % it
        o.set_nominal(gold[pre + "u_in"])
        o.sampling(int(gold[pre + "seed"]), rng="mt19937")
        ctrl = o.get_controls()
        np.testing.assert_allclose(ctrl[sel], gold[pre + "controls_sel"], rtol=1e-14, atol=1e-16)
        g.set_nominal(gold[pre + "u_in"])
        g.inject_controls(ctrl)
        g.predict_States(gold[pre + "x0"], p.dt)
        g.calc_Weights(gold[pre + "x_ref"], gold[pre + "y_ref"], float(gold[pre + "yaw_ref0"]))
        u_g, st = g.determine_OptimalSolution(want_stats=True)
        assert np.max(np.abs(g.read_costs() - gold[pre + "costs"]) / gold[pre + "costs"]) < TOL_COST
        assert abs(st.sum_w - gold[pre + "sum_w"]) <= 1e-9 * gold[pre + "sum_w"]
        assert helpers.rel_err(u_g, gold[pre + "u_out"]) < 1e-9
        xy = g.read_candidates()[sel]
        np.testing.assert_allclose(xy[..., 0], gold[pre + "x_sel"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(xy[..., 1], gold[pre + "y_sel"], rtol=1e-12, atol=1e-13)


# --------------------------------------------------------------------------------------------------------------
# R'->G: the fused production iteration (device Philox) against the oracle's philox mode
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(CASES))
def test_fused_iteration_matches_oracle_philox(name):
    p, kind = CASES[name]
    path = helpers.oracle_path(kind)
    o, g = helpers.oracle_for(p), MPPIController(p)
    state = start_state(p, path, lateral=0.07)
    for it in range(4):
        xr, yr, yaw = helpers.oracle_window(p, path, state)
        u_o = o.iterate(state, p.dt, xr, yr, yaw[0], seed=77, rng="philox", iteration=it)
        u_g, st = g.iterate(state, p.dt, xr, yr, yaw[0], 77, it)
        np.testing.assert_array_equal(g.read_controls(), o.get_controls())
        assert np.max(np.abs(g.read_costs() - o.costs()) / o.costs()) < TOL_COST
        assert helpers.rel_err(u_g, u_o) < 1e-8
        # keep the two loops on the same nominal so the next iteration's noise means are bit-identical
        o.set_nominal(u_g)
        state = helpers.plant(p.model, state, u_g[0], p.dt)


@pytest.mark.parametrize("name", ["dd_K256_H50_sinusoid_C2", "sd_K256_H50_sinusoid_C3", "fb_K128_H80_dkan_C4"])
def test_fused_equals_stagewise_bitwise(name):
    p, kind = CASES[name]
    path = helpers.oracle_path(kind)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    a, b = MPPIController(p), MPPIController(p)
    u_a, st_a = a.iterate(state, p.dt, xr, yr, yaw[0], 3, 0)
    b.sampling(3, 0)
    b.predict_States(state, p.dt)
    b.calc_Weights(xr, yr, yaw[0])
    u_b, st_b = b.determine_OptimalSolution(want_stats=True)
    np.testing.assert_array_equal(a.read_controls(), b.read_controls())
    np.testing.assert_array_equal(a.read_candidates(), b.read_candidates())
    np.testing.assert_array_equal(a.read_costs(), b.read_costs())
    np.testing.assert_array_equal(u_a, u_b)
    assert st_a.sum_w == st_b.sum_w


@pytest.mark.parametrize("name", ["dd_K256_H50_sinusoid_C2", "fb_K128_H80_dkan_C4"])
def test_blocking_result_mailbox_equals_copy_and_synchronise(monkeypatch, name):
    """The blocking calls (ccv_mppi_iterate, ccv_mppi_update) get u* and the statistics from a mailbox in pinned host memory
    that the update kernel writes itself (self-validating packets, polled by the host) instead of two device-to-host copies
    and a stream synchronisation (CCV_MPPI_MAILBOX=0): the same bits, call after call, with and without a statistics
    pointer, fused and stage-wise, and ccv_mppi_get_nominal (a copy) agrees with what the mailbox delivered."""
    p, kind = CASES[name]
    path = helpers.oracle_path(kind)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    a = MPPIController(p)
    monkeypatch.setenv("CCV_MPPI_MAILBOX", "0")
    b = MPPIController(p)
    for it in range(6):
        if it % 2:
            u_a, u_b = (g.iterate(state, p.dt, xr, yr, yaw[0], 3, it, want_stats=False) for g in (a, b))
        else:
            (u_a, st_a), (u_b, st_b) = (g.iterate(state, p.dt, xr, yr, yaw[0], 3, it) for g in (a, b))
            assert (st_a.sum_w, st_a.min_cost, st_a.max_cost, st_a.n_zero_weight, st_a.nonfinite) == \
                   (st_b.sum_w, st_b.min_cost, st_b.max_cost, st_b.n_zero_weight, st_b.nonfinite)
            assert st_a.min_cost == a.read_costs().min() and st_a.max_cost == a.read_costs().max()
        np.testing.assert_array_equal(u_a, u_b)
        np.testing.assert_array_equal(u_a, a.get_nominal())
    for g in (a, b):
        g.sampling(3, 9)
        g.predict_States(state, p.dt)
        g.calc_Weights(xr, yr, yaw[0])
    (u_a, st_a), (u_b, st_b) = (g.determine_OptimalSolution(want_stats=True) for g in (a, b))
    np.testing.assert_array_equal(u_a, u_b)
    assert st_a.sum_w == st_b.sum_w and np.all(np.isfinite(u_a))


@pytest.mark.parametrize("name", ["dd_K256_H50_sinusoid_C2", "sd_K256_H50_sinusoid_C3", "fb_K128_H80_dkan_C4"])
def test_kernel_variants_agree(monkeypatch, name):
    """The production kernels (two or four waves share 64 samples) against the plain one-sample-per-lane variants kept for
    experiments (CCV_MPPI_KERNEL=v1, LDS or scalar-load window): same samples, costs equal up to summation order."""
    p, kind = CASES[name]
    path = helpers.oracle_path(kind)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    a = MPPIController(p)
    monkeypatch.setenv("CCV_MPPI_KERNEL", "v1")
    b = MPPIController(p)
    monkeypatch.setenv("CCV_MPPI_WINDOW", "scalar")
    c = MPPIController(p)
    res = [g.iterate(state, p.dt, xr, yr, yaw[0], 3, 0, want_stats=False) for g in (a, b, c)]
    np.testing.assert_array_equal(b.read_costs(), c.read_costs())
    np.testing.assert_array_equal(a.read_controls(), b.read_controls())
    # the production kernel uses a branch-free sin/cos (<= 1-2 ulp from OCML's): states agree to rounding
    np.testing.assert_allclose(a.read_candidates(), b.read_candidates(), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(a.read_costs(), b.read_costs(), rtol=1e-12)
    np.testing.assert_allclose(res[0], res[1], rtol=1e-10, atol=1e-14)
    np.testing.assert_array_equal(res[1], res[2])


@pytest.mark.parametrize("wl,K,H", [("C2", 256, 50), ("C2", 1000, 50), ("C2", 130, 9), ("C2", 64, 3), ("C2", 4097, 128),
                                    ("C2", 777, 17), ("C2", 640, 25), ("C3", 1000, 50), ("C3", 321, 9), ("C3", 64, 3),
                                    ("C3", 2049, 128), ("C2", 1000, 15), ("C3", 1000, 15)])   # (H = 15: the reference's default --
                                    # a last block of six steps, the four-wave kernel's compile-time instantiation of the masked batch)
def test_multi_wave_kernels_agree(monkeypatch, wl, K, H):
    """Diff-drive and steering run the four-wave kernel (noise / dynamics / distance / store wave, mppi_rollout_r4.h) by
    default; the three-wave kernel (mppi_rollout_r3.h) splits the same arithmetic over the same cost parts and must give
    the same bits everywhere -- samples, states, per-sample costs, weights, controls -- over two iterations; the two-wave
    kernel (mppi_rollout_pc.h, the one full body uses) the same samples and states bit for bit and the same costs and
    controls up to the order in which the per-wave cost parts and the rows' partial sums are added.  The wave-priority
    rotation must not change anything.  Horizons with full blocks only, with a one-step tail, shorter than the
    pipeline is deep."""
    w = configs.workload(wl)
    p = w.params.with_(num_samples=K, horizon=H)
    path = helpers.oracle_path(w.path)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    monkeypatch.delenv("CCV_MPPI_KERNEL", raising=False)
    a = MPPIController(p)
    monkeypatch.setenv("CCV_MPPI_PRIO", "0")
    a0 = MPPIController(p)
    monkeypatch.delenv("CCV_MPPI_PRIO", raising=False)
    monkeypatch.setenv("CCV_MPPI_KERNEL", "r3")
    r3 = MPPIController(p)
    monkeypatch.setenv("CCV_MPPI_KERNEL", "pc")
    b = MPPIController(p)
    for it in range(2):   # (the second iteration starts from the first one's update)
        res = [g.iterate(state, p.dt, xr, yr, yaw[0], 11, 4 + it, want_stats=False) for g in (a, a0, r3, b)]
        np.testing.assert_array_equal(res[0], res[1])
        np.testing.assert_array_equal(res[0], res[2])
        np.testing.assert_allclose(res[0], res[3], rtol=1e-10, atol=1e-13)
    np.testing.assert_array_equal(a.read_costs(), a0.read_costs())
    np.testing.assert_array_equal(a.read_costs(), r3.read_costs())
    np.testing.assert_array_equal(a.read_weights(), r3.read_weights())
    np.testing.assert_array_equal(a.read_controls(), r3.read_controls())
    np.testing.assert_array_equal(a.read_candidates(), r3.read_candidates())
    # (second iteration: the warm starts of the two already differ by the summation order of the first update)
    np.testing.assert_allclose(a.read_controls(), b.read_controls(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(a.read_candidates(), b.read_candidates(), rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(a.read_costs(), b.read_costs(), rtol=1e-10)


@pytest.mark.parametrize("wl,K,H", [("C4", 256, 80), ("C4", 1000, 80), ("C4", 130, 9), ("C4", 64, 3), ("C4", 2049, 128),
                                    ("C2", 1000, 50), ("C3", 321, 50)])
def test_one_wave_kernel_agrees_with_the_multi_wave_kernels(monkeypatch, wl, K, H):
    """Full body with two or more blocks of 64 samples per SIMD runs k_rollout_solo (one wave per 64 samples, no barrier
    in the time loop, mppi_rollout_solo.h).  Same building blocks as the multi-wave kernels: same samples and
    states bit for bit, costs and controls equal up to the order in which the cost terms are added; and equal to the
    oracle within the north_star tolerance."""
    w = configs.workload(wl)
    p = w.params.with_(num_samples=K, horizon=H)
    path = helpers.oracle_path(w.path)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    monkeypatch.delenv("CCV_MPPI_KERNEL", raising=False)
    a = MPPIController(p)
    monkeypatch.setenv("CCV_MPPI_KERNEL", "solo")
    b = MPPIController(p)
    for it in range(2):   # (the second iteration starts from the first one's update)
        ua = a.iterate(state, p.dt, xr, yr, yaw[0], 11, it, want_stats=False)
        ub = b.iterate(state, p.dt, xr, yr, yaw[0], 11, it, want_stats=False)
        np.testing.assert_allclose(ua, ub, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(a.read_controls(), b.read_controls(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(a.read_candidates(), b.read_candidates(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(a.read_costs(), b.read_costs(), rtol=1e-12)
    o = helpers.oracle_for(p)
    for it in range(2):
        uo = o.iterate(state, p.dt, xr, yr, yaw[0], seed=11, rng="philox", iteration=it)
    np.testing.assert_allclose(ub, uo, rtol=TOL_U, atol=1e-12)


@pytest.mark.parametrize("K,H", [(10000, 15), (256, 80), (1000, 23), (130, 9), (64, 3), (777, 128), (2049, 12)])
def test_full_body_four_wave_kernel_agrees_with_the_two_wave_kernel(monkeypatch, K, H):
    """Round 3: full body up to one block of 64 samples per CU -- the reference's own operating point, K = 10 000, H = 15 --
    runs the four-wave kernel (mppi_rollout_r4.h, one wave per SIMD); beyond, the two-wave kernel (mppi_rollout_pc.h).  The
    same building blocks: the same samples and states bit for bit, costs and controls equal up to the order in which the
    waves' cost parts and the rows' partial sums are added; both equal to the oracle.  Horizons with a full-block tail, a
    masked partial tail (7, 6 steps), a short tail taken step by step (3 steps) and shorter than one block."""
    p = configs.workload("C4").params.with_(num_samples=K, horizon=H)
    path = helpers.oracle_path("dkan")
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    monkeypatch.delenv("CCV_MPPI_KERNEL", raising=False)
    a = MPPIController(p)               # default: the four-wave kernel at these sizes
    monkeypatch.setenv("CCV_MPPI_KERNEL", "pc")
    b = MPPIController(p)
    monkeypatch.setenv("CCV_MPPI_KERNEL", "r4")
    c = MPPIController(p)               # forced: the same kernel as the default must have been
    for it in range(2):
        ua, ub, uc = (g.iterate(state, p.dt, xr, yr, yaw[0], 5, it, want_stats=False) for g in (a, b, c))
        np.testing.assert_array_equal(ua, uc)
        np.testing.assert_allclose(ua, ub, rtol=1e-10, atol=1e-13)
        if it == 0:
            np.testing.assert_array_equal(a.read_controls(), b.read_controls())
            np.testing.assert_array_equal(a.read_candidates(), b.read_candidates())
            np.testing.assert_allclose(a.read_costs(), b.read_costs(), rtol=1e-12)
    np.testing.assert_array_equal(a.read_costs(), c.read_costs())
    o = helpers.oracle_for(p, min(K, 2048))
    kcmp = min(K, 2048)
    for it in range(2):
        if it == 1:
            o.set_nominal(ua_first)
        o.iterate(state, p.dt, xr, yr, yaw[0], seed=5, rng="philox", iteration=it)
        if it == 0:
            # (the oracle only rolls the first 2 048 samples at the large K: their costs and controls; u* from the device)
            h = MPPIController(p)
            ua_first = h.iterate(state, p.dt, xr, yr, yaw[0], 5, 0, want_stats=False)
            np.testing.assert_array_equal(h.read_controls(0, kcmp), o.get_controls())
            assert np.max(np.abs(h.read_costs(0, kcmp) - o.costs()) / o.costs()) < TOL_COST
    np.testing.assert_array_equal(a.read_controls(0, kcmp), o.get_controls())
    assert np.max(np.abs(a.read_costs(0, kcmp) - o.costs()) / o.costs()) < TOL_COST


# --------------------------------------------------------------------------------------------------------------
# edge cases
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [1, 2, 63, 64, 65, 255, 257, 2047, 2049, 4097])
def test_ragged_sample_counts(K):
    p = configs.workload("C2").params.with_(num_samples=K, horizon=17)
    path = helpers.oracle_path("sinusoid")
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    o, g = helpers.oracle_for(p), MPPIController(p)
    u_o = o.iterate(state, p.dt, xr, yr, yaw[0], seed=1, rng="philox", iteration=0)
    u_g, st = g.iterate(state, p.dt, xr, yr, yaw[0], 1, 0)
    assert np.max(np.abs(g.read_costs() - o.costs()) / o.costs()) < TOL_COST
    assert helpers.rel_err(u_g, u_o) < 1e-9
    assert abs(st.sum_w - o.sum_w()) <= 1e-9 * o.sum_w()


@pytest.mark.parametrize("model,H", [("diff_drive", 3), ("diff_drive", 4), ("diff_drive", 9), ("diff_drive", 128),
                                     ("steering_diff_drive", 3), ("steering_diff_drive", 127), ("full_body", 3),
                                     ("full_body", 4), ("full_body", 10), ("full_body", 128)])
def test_horizon_extremes(model, H):
    mk = {"diff_drive": configs.diff_drive_defaults, "steering_diff_drive": configs.steering_defaults,
          "full_body": configs.full_body_defaults}[model]
    p = mk(96, H)
    path = helpers.oracle_path("dkan")
    state = start_state(p, path, lateral=-0.1)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    o, g = helpers.oracle_for(p), MPPIController(p)
    u_o = o.iterate(state, p.dt, xr, yr, yaw[0], seed=2, rng="philox", iteration=1)
    u_g, st = g.iterate(state, p.dt, xr, yr, yaw[0], 2, 1)
    c_o = o.costs()
    assert np.max(np.abs(g.read_costs() - c_o) / np.maximum(np.abs(c_o), 1e-300)) < TOL_COST
    assert helpers.rel_err(u_g, u_o) < 1e-9


def test_all_weights_underflow_gives_nan_like_the_reference():
    """dd:219-222 has no min-cost shift: when every exp underflows the update is 0/0."""
    p = configs.workload("C2").params.with_(num_samples=128, horizon=20, path_weight=1e4)
    path = helpers.oracle_path("sinusoid")
    state = np.array([3.0, 40.0, 0.0])    # far from the path: cost >> 745
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    o, g = helpers.oracle_for(p), MPPIController(p)
    u_o = o.iterate(state, p.dt, xr, yr, yaw[0], seed=1, rng="philox", iteration=0)
    u_g, st = g.iterate(state, p.dt, xr, yr, yaw[0], 1, 0)
    assert o.sum_w() == 0.0 and np.all(np.isnan(u_o))
    assert st.sum_w == 0.0 and np.all(np.isnan(u_g)) and st.nonfinite == 1 and st.n_zero_weight == 128
    # the flagged, non-reference min-shift mode stays finite and equals the shifted softmax of the same costs
    g2 = MPPIController(p, min_shift=True)
    u2, st2 = g2.iterate(state, p.dt, xr, yr, yaw[0], 1, 0)
    c, u = o.costs(), o.get_controls()
    w = np.exp(-(c - c.min()) / p.lam)
    assert st2.nonfinite == 0
    np.testing.assert_allclose(u2, np.einsum("i,itd->td", w / w.sum(), u), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("wl,K,H", [("C2", 1000, 50), ("C3", 640, 50), ("C4", 512, 80), ("C2", 200, 9)])
def test_two_instruction_clamp_equals_compare_and_select(monkeypatch, wl, K, H):
    """The rollout kernels clamp with v_max / v_min where no control can be NaN (sigma finite, bounds in order -- checked on
    the host -- and no NaN in the staged warm start -- checked by the kernel) and with the reference's compare-and-select
    otherwise (dd:98-99 passes a NaN through).  CCV_MPPI_FAST_CLAMP=0 forces the second form: the controls must be the same
    bits -- with bounds that really bite -- and everything downstream equal up to the sin / cos formulation of the block
    path the forced run takes."""
    w = configs.workload(wl)
    p = w.params.with_(num_samples=K, horizon=H, control_noise=1.5)   # (wide noise: many samples at a bound)
    path = helpers.oracle_path(w.path)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    monkeypatch.delenv("CCV_MPPI_FAST_CLAMP", raising=False)
    a = MPPIController(p)
    monkeypatch.setenv("CCV_MPPI_FAST_CLAMP", "0")
    b = MPPIController(p)
    ua = a.iterate(state, p.dt, xr, yr, yaw[0], 5, 0, want_stats=False)
    ub = b.iterate(state, p.dt, xr, yr, yaw[0], 5, 0, want_stats=False)
    ca, cb = a.read_controls(), b.read_controls()
    np.testing.assert_array_equal(ca, cb)
    lo, hi = np.array(p.u_min[:p.udim]), np.array(p.u_max[:p.udim])
    assert np.mean((ca == lo) | (ca == hi)) > 0.05 and np.all(ca >= lo) and np.all(ca <= hi)
    np.testing.assert_allclose(a.read_candidates(), b.read_candidates(), rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(a.read_costs(), b.read_costs(), rtol=1e-9)
    np.testing.assert_allclose(ua, ub, rtol=1e-8, atol=1e-11)
    o = helpers.oracle_for(p)
    uo = o.iterate(state, p.dt, xr, yr, yaw[0], seed=5, rng="philox", iteration=0)
    np.testing.assert_array_equal(ca, o.get_controls())
    np.testing.assert_allclose(ua, uo, rtol=TOL_U, atol=1e-12)


@pytest.mark.parametrize("wl,K,H", [("C2", 256, 50), ("C4", 256, 80)])
def test_nan_in_the_warm_start_passes_through_the_clamp(wl, K, H):
    """A NaN in u* (e.g. after an all-underflow update) makes every control of its row NaN in the reference -- the clamp's two
    comparisons are false (dd:98-99).  The kernels must notice the NaN when they stage the warm start and must not clamp it to
    a bound with v_max / v_min: the row's controls are NaN, the other rows' controls are the oracle's bits."""
    w = configs.workload(wl)
    p = w.params.with_(num_samples=K, horizon=H)
    path = helpers.oracle_path(w.path)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    g, o = MPPIController(p), helpers.oracle_for(p)
    nom = np.zeros((H - 1, p.udim))
    nom[3, 1] = np.nan
    g.set_nominal(nom)
    o.set_nominal(nom)
    g.iterate(state, p.dt, xr, yr, yaw[0], 9, 0, want_stats=False)
    o.iterate(state, p.dt, xr, yr, yaw[0], seed=9, rng="philox", iteration=0)
    cg, co = g.read_controls(), o.get_controls()
    assert np.all(np.isnan(cg[:, 3, 1])) and np.all(np.isnan(co[:, 3, 1]))
    keep = np.ones(cg.shape[1:], dtype=bool)
    keep[3, 1] = False
    np.testing.assert_array_equal(cg[:, keep], co[:, keep])
    assert np.all(np.isfinite(cg[:, keep]))


def test_distance_gate_100m():
    """calc_MinDistance starts from min_distance = 100 (dd:185): farther points all cost path_weight*100^2."""
    p = configs.diff_drive_defaults(64, 10)
    path = helpers.oracle_path("straight")
    state = np.array([5000.0, 5000.0, 0.3])
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    o, g = helpers.oracle_for(p), MPPIController(p)
    o.iterate(state, p.dt, xr, yr, yaw[0], seed=1, rng="philox", iteration=0)
    g.iterate(state, p.dt, xr, yr, yaw[0], 1, 0)
    assert np.all(o.costs() >= p.horizon * p.path_weight * 1e4)
    np.testing.assert_allclose(g.read_costs(), o.costs(), rtol=1e-12)


def test_large_world_coordinates():
    """Window coefficients are taken relative to the current pose, so an odom origin far away costs no accuracy."""
    p = configs.workload("C2").params.with_(num_samples=128, horizon=30)
    px, py = helpers.oracle_path("sinusoid")
    off = np.array([12345.678, -9876.543])
    path = (px + off[0], py + off[1])
    state = np.array([off[0] + 0.2, off[1] - 0.1, 0.1])
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    o, g = helpers.oracle_for(p), MPPIController(p)
    u_o = o.iterate(state, p.dt, xr, yr, yaw[0], seed=4, rng="philox", iteration=0)
    u_g = g.iterate(state, p.dt, xr, yr, yaw[0], 4, 0, want_stats=False)
    assert np.max(np.abs(g.read_costs() - o.costs()) / o.costs()) < TOL_COST
    assert helpers.rel_err(u_g, u_o) < 1e-8


@pytest.mark.parametrize("model", ["diff_drive", "steering_diff_drive", "full_body"])
def test_huge_heading_takes_the_libm_path(model):
    """The production kernel's branch-free sin/cos is valid for |angle| <= 1e5; beyond that the host routes the call to
    the plain kernel with OCML's sincos (ccv_mppi_capi.hip: fast_trig_safe).  Results must still match the oracle."""
    mk = {"diff_drive": configs.diff_drive_defaults, "steering_diff_drive": configs.steering_defaults,
          "full_body": configs.full_body_defaults}[model]
    p = mk(192, 20).with_(yaw_weight=0.0)        # (the fb yaw term would be 2*(3e5)^2 and underflow every weight)
    path = helpers.oracle_path("sinusoid")
    state = np.zeros(p.nstate)
    state[:3] = 0.3, -0.2, 3.0e5 + 0.4          # yaw far outside [-pi, pi]
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    o, g = helpers.oracle_for(p), MPPIController(p)
    u_o = o.iterate(state, p.dt, xr, yr, yaw[0], seed=6, rng="philox", iteration=2)
    u_g, st = g.iterate(state, p.dt, xr, yr, yaw[0], 6, 2)
    assert np.max(np.abs(g.read_costs() - o.costs()) / o.costs()) < TOL_COST
    assert helpers.rel_err(u_g, u_o) < 1e-8
    # the stage-wise path with injected, unbounded controls makes the same decision from the injected magnitudes
    ctrl = o.get_controls().copy()
    ctrl[5, 3, 1] = 4.0e6                       # one absurd yaw rate
    o.set_controls(ctrl)
    o.predict_States(state, p.dt)
    o.calc_Weights(xr, yr, yaw[0])
    g.inject_controls(ctrl)
    g.predict_States(state, p.dt)
    g.calc_Weights(xr, yr, yaw[0])
    c_o, c_g = o.costs(), g.read_costs()
    assert np.max(np.abs(c_g - c_o) / c_o) < TOL_COST


@pytest.mark.parametrize("dt", [0.1, 0.39, 0.4, 1.0])
def test_diff_drive_turn_per_step_gate(dt):
    """Diff drive advances (sin, cos) of the heading by the step's turn angle w*dt with short polynomials valid for
    |w| dt <= pi/4 (csrc/fast_trig.h: kernel_sincos_n); the host checks the bound per call and otherwise launches the same
    kernel's wide-turn instantiation, which evaluates sin / cos of every heading in full (w_max = 2 rad/s: the switch is at
    dt = 0.3927; round 2 fell back to the plain kernel there, at twice the time).  Either way the oracle's sin(yaw),
    cos(yaw) are matched."""
    p = configs.diff_drive_defaults(320, 50).with_(dt=dt)
    path = helpers.oracle_path("sinusoid")
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    o, g = helpers.oracle_for(p), MPPIController(p)
    for it in range(2):
        u_o = o.iterate(state, p.dt, xr, yr, yaw[0], seed=8, rng="philox", iteration=it)
        u_g, st = g.iterate(state, p.dt, xr, yr, yaw[0], 8, it)
    assert np.max(np.abs(g.read_costs() - o.costs()) / o.costs()) < TOL_COST
    assert helpers.rel_err(u_g, u_o) < 1e-8
    xs_o = np.stack([o.states("x"), o.states("y")], axis=-1)
    np.testing.assert_allclose(g.read_candidates(), xs_o, rtol=1e-11, atol=1e-11)


def test_measured_dt_across_the_small_turn_gate(monkeypatch):
    """The node takes dt from its clock (dd:346-348), so one slow tick can cross |w|max * dt = pi/4 (0.3927 s at w_max = 2)
    and change instantiations: the rotation-based production kernel below, its wide-turn form (full-range sin / cos of every
    heading) above (fast_trig_safe, ccv_mppi_capi.hip).  Loop periods drawn around the gate: whatever the host picks must
    agree with the plain kernel forced (CCV_MPPI_KERNEL=v1, OCML's sincos) and with the oracle, with no jump at the switch;
    the warm start carries over from one period to the next as it would in the node."""
    p0 = configs.diff_drive_defaults(640, 50)
    path = helpers.oracle_path("sinusoid")
    state = start_state(p0, path)
    gate = np.pi / 4 / 2.0
    dts = [gate * f for f in (0.97, 0.999, 0.9999999, 1.0, 1.0000001, 1.001, 1.03)] + [0.39, 0.1, 0.41, 0.1]
    g = MPPIController(p0)
    monkeypatch.setenv("CCV_MPPI_KERNEL", "v1")
    ref = MPPIController(p0)
    o = helpers.oracle_for(p0)
    for it, dt in enumerate(dts):
        p = p0.with_(dt=dt)
        xr, yr, yaw = helpers.oracle_window(p, path, state)
        u_g = g.iterate(state, dt, xr, yr, yaw[0], 8, it, want_stats=False)
        u_r = ref.iterate(state, dt, xr, yr, yaw[0], 8, it, want_stats=False)
        u_o = o.iterate(state, dt, xr, yr, yaw[0], seed=8, rng="philox", iteration=it)
        np.testing.assert_array_equal(g.read_controls(), ref.read_controls())
        np.testing.assert_allclose(g.read_costs(), ref.read_costs(), rtol=1e-11)
        assert np.max(np.abs(g.read_costs() - o.costs()) / o.costs()) < TOL_COST
        assert helpers.rel_err(u_g, u_r) < 1e-9 and helpers.rel_err(u_g, u_o) < 1e-8
        ref.set_nominal(u_g)
        o.set_nominal(u_g)


@pytest.mark.parametrize("case", ["launch", "dt_0.7", "dt_1.0", "wide_direction", "fast_roll"])
def test_full_body_small_angle_gates(case):
    """Full body evaluates sin / cos of yaw, roll and pitch once per block of 8 steps and advances them by rotations with
    short polynomials, and takes sin / cos of the direction angle without range reduction (pc_produce_batched): valid for
    |rate| dt <= pi/4 and |direction| <= pi/4, which the host checks per call (fast_trig_safe) -- anything else runs the plain
    kernel.  Either way the oracle's direct evaluations are matched within the usual tolerances."""
    p = configs.workload("C4", num_samples=320, horizon=40).params
    if case == "dt_0.7":        # w_max = 1 rad/s: 0.7 rad per step stays inside the gate, 1.0 below does not
        p = p.with_(dt=0.7)
    elif case == "dt_1.0":
        p = p.with_(dt=1.0)
    elif case == "wide_direction":   # direction up to 60 degrees
        lo, hi = list(p.u_min), list(p.u_max)
        lo[2], hi[2] = -np.pi / 3, np.pi / 3
        p = p.with_(u_min=tuple(lo), u_max=tuple(hi))
    elif case == "fast_roll":        # roll rate up to 10 rad/s: 1 rad per step
        lo, hi = list(p.u_min), list(p.u_max)
        lo[3], hi[3] = -10.0, 10.0
        p = p.with_(u_min=tuple(lo), u_max=tuple(hi))
    path = helpers.oracle_path("dkan")
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    o, g = helpers.oracle_for(p), MPPIController(p)
    for it in range(2):
        u_o = o.iterate(state, p.dt, xr, yr, yaw[0], seed=8, rng="philox", iteration=it)
        u_g, st = g.iterate(state, p.dt, xr, yr, yaw[0], 8, it)
    assert np.max(np.abs(g.read_costs() - o.costs()) / o.costs()) < TOL_COST
    assert helpers.rel_err(u_g, u_o) < 1e-8
    xs_o = np.stack([o.states("x"), o.states("y")], axis=-1)
    np.testing.assert_allclose(g.read_candidates(), xs_o, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("kind", ["random", "circle", "duplicates", "far", "line_behind"])
def test_window_pruning_is_exact_on_awkward_windows(kind):
    """Diff drive skips window points that cannot be the nearest one for any sample of a wave (pc_consume: bounds of the
    linear form over the wave's bounding box, then the hull of the surviving indices).  The argument does not assume a
    path-like window: windows that are unordered, closed, degenerate or far away must give the oracle's costs too."""
    p = configs.diff_drive_defaults(512, 50)
    rng = np.random.default_rng(3)
    H = p.horizon
    state = np.array([0.4, -0.3, 0.7])
    if kind == "random":
        xr, yr = rng.uniform(-3, 3, H), rng.uniform(-3, 3, H)
    elif kind == "circle":
        a = np.linspace(0, 2 * np.pi, H, endpoint=False)
        xr, yr = state[0] + 1.5 * np.cos(a), state[1] + 1.5 * np.sin(a)
    elif kind == "duplicates":
        xr, yr = np.full(H, 1.0), np.full(H, 0.5)
        xr[::7] += 0.8
    elif kind == "far":
        xr, yr = np.linspace(150.0, 160.0, H), np.linspace(-90.0, -80.0, H)     # beyond the 100 m gate for most samples
    else:
        xr, yr = state[0] - np.linspace(0.0, 4.0, H), np.full(H, state[1] + 0.2)  # a line behind the robot
    o, g = helpers.oracle_for(p), MPPIController(p)
    for it in range(1 if kind == "far" else 3):   # (far: every weight underflows, the next warm start is NaN -- SURVEY Q4)
        u_o = o.iterate(state, p.dt, xr, yr, 0.0, seed=4, rng="philox", iteration=it)
        u_g, st = g.iterate(state, p.dt, xr, yr, 0.0, 4, it)
        assert np.max(np.abs(g.read_costs() - o.costs()) / o.costs()) < TOL_COST
    if kind != "far":
        assert helpers.rel_err(u_g, u_o) < 1e-8


def test_nan_pose_propagates_like_the_reference():
    """No NaN guard anywhere in the reference: a NaN pose makes every cost, weight and control NaN."""
    p = configs.diff_drive_defaults(128, 12)
    path = helpers.oracle_path("straight")
    state = np.array([0.0, 0.0, np.nan])
    xr, yr, yaw = helpers.oracle_window(p, path, np.zeros(3))
    o, g = helpers.oracle_for(p), MPPIController(p)
    u_o = o.iterate(state, p.dt, xr, yr, yaw[0], seed=1, rng="philox", iteration=0)
    u_g, st = g.iterate(state, p.dt, xr, yr, yaw[0], 1, 0)
    assert np.all(np.isnan(u_o)) and np.all(np.isnan(u_g)) and st.nonfinite == 1


def test_nan_warm_start_stays_nan_like_the_reference():
    """After an all-underflow iteration optimal_solution is NaN; the reference then draws N(NaN, sigma) forever."""
    p = configs.workload("C2").params.with_(num_samples=256, horizon=20)
    path = helpers.oracle_path("sinusoid")
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    o, g = helpers.oracle_for(p), MPPIController(p)
    nom = np.zeros((p.horizon - 1, p.udim))
    nom[3, 0] = np.nan
    o.set_nominal(nom)
    g.set_nominal(nom)
    u_o = o.iterate(state, p.dt, xr, yr, yaw[0], seed=1, rng="philox", iteration=0)
    u_g, st = g.iterate(state, p.dt, xr, yr, yaw[0], 1, 0)
    assert np.all(np.isnan(u_o)) and np.all(np.isnan(u_g)) and st.nonfinite == 1


def test_read_back_ranges_and_errors():
    p = configs.workload("C2").params.with_(num_samples=300, horizon=12)
    path = helpers.oracle_path("sinusoid")
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    g = MPPIController(p)
    with pytest.raises(MPPIError) as e:
        g.predict_States(state, p.dt)          # rollout before sampling
    assert e.value.code == capi.ERR_STATE
    with pytest.raises(MPPIError):
        g.read_costs()
    g.iterate(state, p.dt, xr, yr, yaw[0], 1, 0)
    full = g.read_candidates()
    assert full.shape == (300, 12, 2)
    np.testing.assert_array_equal(g.read_candidates(5, 30, 7), full[5:5 + 30 * 7:7])
    np.testing.assert_array_equal(full[:, 0, 0], np.full(300, state[0]))
    np.testing.assert_array_equal(g.read_costs(10, 20), g.read_costs()[10:30])
    np.testing.assert_array_equal(g.read_controls(290, 10), g.read_controls()[290:])
    assert abs(g.read_weights().sum() - 1.0) < 1e-12
    for bad in (lambda: g.read_candidates(0, 301, 1), lambda: g.read_candidates(299, 2, 1), lambda: g.read_costs(290, 11),
                lambda: g.read_candidates(0, 1, 0), lambda: g.read_controls(-1, 1)):
        with pytest.raises(MPPIError) as e:
            bad()
        assert e.value.code == capi.ERR_INVALID_ARG
    lean = MPPIController(p, no_state_store=True)
    lean.iterate(state, p.dt, xr, yr, yaw[0], 1, 0)
    np.testing.assert_array_equal(lean.read_costs(), g.read_costs())
    with pytest.raises(MPPIError):
        lean.read_candidates()


@pytest.mark.parametrize("wl,K,H", [("C2", 1000, 50), ("C3", 640, 50), ("C2", 300, 128), ("C4", 256, 80)])
def test_iteration_without_the_state_store(wl, K, H):
    """CCV_MPPI_FLAG_NO_STATE_STORE drops the K x H (x, y) buffer (bench.py --no-state-store); everything else of an
    iteration -- costs, weights, controls, the update -- must be the same bits as with it, over two iterations (the four-wave
    kernel's store wave counts its outstanding stores for the epilogue's early re-read: a different count without the states)."""
    w = configs.workload(wl)
    p = w.params.with_(num_samples=K, horizon=H)
    path = helpers.oracle_path(w.path)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    full, lean = MPPIController(p), MPPIController(p, no_state_store=True)
    for it in range(2):
        uf = full.iterate(state, p.dt, xr, yr, yaw[0], 13, it, want_stats=False)
        ul = lean.iterate(state, p.dt, xr, yr, yaw[0], 13, it, want_stats=False)
        np.testing.assert_array_equal(uf, ul)
    np.testing.assert_array_equal(full.read_costs(), lean.read_costs())
    np.testing.assert_array_equal(full.read_weights(), lean.read_weights())
    np.testing.assert_array_equal(full.read_controls(), lean.read_controls())


# --------------------------------------------------------------------------------------------------------------
# full BASELINE sizes: size-independent properties
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("wl,K,N", [("C2", 65536, 32), ("C2", 1000, 1000), ("C3", 4096, 1), ("C4", 2048, 100)])
def test_top_candidates(wl, K, N):
    """publish_CandidatePath() feed: the N highest-weight samples, selected on the device, against a host
    sort of all weights; their rollouts against the strided read-back of the same samples."""
    w = configs.workload(wl, num_samples=K)
    p = w.params
    path = helpers.oracle_path(w.path)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    g = MPPIController(p)
    g.iterate(state, p.dt, xr, yr, yaw[0], 21, 2, want_stats=False)
    wts = g.read_weights() * 1.0
    idx, top_w, xy = g.read_top_candidates(N)
    ref = np.lexsort((np.arange(K), -wts))[:N]          # descending weight, ties by index
    np.testing.assert_array_equal(idx, ref)
    # read_weights returns normalised weights, the top list unnormalised ones: same order, constant ratio
    ratio = top_w / wts[idx]
    assert np.allclose(ratio, ratio[0], rtol=1e-12)
    for j in (0, N // 2, N - 1):
        np.testing.assert_array_equal(xy[j], g.read_candidates(first=int(idx[j]), count=1)[0])
    idx2, _, none = g.read_top_candidates(N, with_paths=False)
    assert none is None
    np.testing.assert_array_equal(idx2, idx)


@pytest.mark.parametrize("wl", ["C2", "C3", "C4"])
def test_full_size_properties(wl):
    w = configs.workload(wl)
    p = w.params
    path = helpers.oracle_path(w.path)
    state = start_state(p, path, lateral=0.05)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    g = MPPIController(p)
    nominal = np.random.default_rng(1).normal(0, 0.2, size=(p.horizon - 1, p.udim))
    g.set_nominal(nominal)
    u1, st = g.iterate(state, p.dt, xr, yr, yaw[0], 42, 7)
    K = p.num_samples
    # (a) contiguous blocks of samples re-scored by the oracle (global sample ids via k_offset)
    for first in (0, K // 2 - 100, K - 192):
        o = helpers.oracle_for(p, 192)
        o.set_nominal(nominal)
        o.iterate(state, p.dt, xr, yr, yaw[0], seed=42, rng="philox", iteration=7, k_offset=first)
        np.testing.assert_array_equal(g.read_controls(first, 192), o.get_controls())
        c_o = o.costs()
        assert np.max(np.abs(g.read_costs(first, 192) - c_o) / c_o) < TOL_COST
        xy = g.read_candidates(first, 192, 1)
        np.testing.assert_allclose(xy[..., 0], o.states("x"), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(xy[..., 1], o.states("y"), rtol=1e-12, atol=1e-12)
    # (b) the reduction at full size: weighted mean recomputed on the host from the device's own costs/controls
    c = g.read_costs()
    wts = np.exp(-c / p.lam)
    assert abs(st.sum_w - wts.sum()) <= 1e-10 * wts.sum()
    assert st.min_cost == c.min() and st.max_cost == c.max() and st.n_zero_weight == int((wts == 0).sum())
    top = np.argsort(c)[:64]              # MPPI weights are extremely peaked (effective sample size O(1..10))
    ctrl_top = np.stack([g.read_controls(int(i), 1)[0] for i in top])
    u_host = np.einsum("i,itd->td", wts[top] / wts.sum(), ctrl_top)
    resid = 1.0 - wts[top].sum() / wts.sum()        # weight mass not included above
    umax = max(np.max(np.abs(p.u_min)), np.max(np.abs(p.u_max)))
    assert np.max(np.abs(u_host - u1)) <= resid * umax + 1e-9
    assert abs(g.read_weights().sum() - 1.0) < 1e-10
    # (c) determinism: same inputs, same bits
    g.set_nominal(nominal)
    u2, st2 = g.iterate(state, p.dt, xr, yr, yaw[0], 42, 7)
    np.testing.assert_array_equal(u1, u2)
    assert st.sum_w == st2.sum_w
    # (d) a different seed or iteration changes the draw
    g.set_nominal(nominal)
    u3 = g.iterate(state, p.dt, xr, yr, yaw[0], 42, 8, want_stats=False)
    assert not np.array_equal(u1, u3)


@pytest.mark.parametrize("wl,K", [("C2", 65536), ("C4", 8192)])
def test_shard_invariance_via_partials(wl, K):
    """K split over two handles with global sample ids: summed partials == the single-handle result."""
    import torch
    w = configs.workload(wl, num_samples=K)
    p = w.params
    path = helpers.oracle_path(w.path)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    whole = MPPIController(p)
    u_whole, st = whole.iterate(state, p.dt, xr, yr, yaw[0], 9, 3)
    cut = K // 2 + 64
    a, b = MPPIController(p, num_samples=cut), MPPIController(p, num_samples=K - cut, sample_offset=cut)
    n = a.partials_size()
    assert n == 1 + (p.horizon - 1) * p.udim
    pa = torch.zeros(n, dtype=torch.float64, device="cuda")
    pb = torch.zeros(n, dtype=torch.float64, device="cuda")
    a.iterate_partials_enqueue(state, p.dt, xr, yr, yaw[0], 9, 3, pa.data_ptr())
    b.iterate_partials_enqueue(state, p.dt, xr, yr, yaw[0], 9, 3, pb.data_ptr())
    a.synchronize()
    b.synchronize()
    np.testing.assert_array_equal(np.concatenate([a.read_costs(), b.read_costs()]), whole.read_costs())
    tot = pa + pb
    assert abs(tot[0].item() - st.sum_w) <= 1e-12 * st.sum_w
    torch.cuda.synchronize()
    a.apply_partials_enqueue(tot.data_ptr())
    b.apply_partials_enqueue(tot.data_ptr())
    np.testing.assert_allclose(a.get_nominal(), u_whole, rtol=1e-11, atol=1e-14)
    np.testing.assert_array_equal(a.get_nominal(), b.get_nominal())


@pytest.mark.parametrize("wl,K", [("C2", 4096), ("C3", 2048), ("C4", 1024)])
def test_sharded_loop_with_deferred_apply(wl, K):
    """Several iterations of the K-sharded loop on two handles (sum of the partial vectors standing in for the
    all-reduce).  ccv_mppi_apply_partials_enqueue is deferred: the next rollout launch divides V by S while it stages the
    warm start.  The loop must follow the single-handle loop, and the warm start read back afterwards must be the one the
    kernel wrote."""
    import torch
    w = configs.workload(wl, num_samples=K)
    p = w.params
    path = helpers.oracle_path(w.path)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    whole = MPPIController(p)
    cut = K // 2 + 64
    a, b = MPPIController(p, num_samples=cut), MPPIController(p, num_samples=K - cut, sample_offset=cut)
    n = a.partials_size()
    pa = torch.zeros(n, dtype=torch.float64, device="cuda")
    pb = torch.zeros(n, dtype=torch.float64, device="cuda")
    tot = torch.zeros(n, dtype=torch.float64, device="cuda")
    for it in range(4):
        u_whole = whole.iterate(state, p.dt, xr, yr, yaw[0], 5, it, want_stats=False)
        a.iterate_partials_enqueue(state, p.dt, xr, yr, yaw[0], 5, it, pa.data_ptr())
        b.iterate_partials_enqueue(state, p.dt, xr, yr, yaw[0], 5, it, pb.data_ptr())
        a.synchronize()
        b.synchronize()
        tot.copy_(pa + pb)
        torch.cuda.synchronize()
        a.apply_partials_enqueue(tot.data_ptr())   # deferred into the next iterate_partials_enqueue
        b.apply_partials_enqueue(tot.data_ptr())
    # the last apply is still pending here: get_nominal performs it
    np.testing.assert_allclose(a.get_nominal(), u_whole, rtol=1e-9, atol=1e-13)
    np.testing.assert_array_equal(a.get_nominal(), b.get_nominal())
    # and a warm start written by the kernel itself (deferred apply consumed by a launch) is what get_nominal returns
    a.apply_partials_enqueue(tot.data_ptr())
    a.iterate_partials_enqueue(state, p.dt, xr, yr, yaw[0], 5, 9, pa.data_ptr())
    a.synchronize()
    np.testing.assert_array_equal(a.get_nominal().ravel(), (tot[1:] / tot[0]).cpu().numpy())


def test_closed_loop_matches_oracle_and_tracks():
    """Closed loop in the spirit of record_state.py / calc_e_rmse.py:30-49: the GPU controller and the oracle
    (philox mode) drive the same kinematic plant along the launch sinusoid and must stay together."""
    w = configs.workload("C2", num_samples=2048)
    p = w.params
    px, py = amd.make_path(w.path)
    s_g = np.array([px[0], py[0], 0.0])
    s_o = s_g.copy()
    g, o = MPPIController(p), helpers.oracle_for(p)
    errs = []
    for it in range(60):
        _, xr, yr, yaw = amd.calc_ref_path(px, py, s_g[0], s_g[1], p.v_ref, p.dt, p.resolution, p.horizon)
        u_g = g.iterate(s_g, p.dt, xr, yr, yaw[0], 123, it, want_stats=False)
        xr_o, yr_o, yaw_o = helpers.oracle_window(p, (px, py), s_o)
        u_o = o.iterate(s_o, p.dt, xr_o, yr_o, yaw_o[0], seed=123, rng="philox", iteration=it)
        assert np.all(np.isfinite(u_g))
        s_g = amd.plant_step(p.model, s_g, u_g[0], p.dt)
        s_o = helpers.plant(p.model, s_o, u_o[0], p.dt)
        errs.append(np.min(np.hypot(px - s_g[0], py - s_g[1])))
    assert np.max(np.abs(s_g - s_o)) < 1e-6     # 60 feedback steps apart by rounding only
    assert s_g[0] > 1.5                        # made progress along the path
    assert np.sqrt(np.mean(np.square(errs[10:]))) < 0.5


# --------------------------------------------------------------------------------------------------------------
# direct exchange of the partial vectors (ccv_mppi_exchange_*): one process, and two processes sharing this device
# --------------------------------------------------------------------------------------------------------------
def test_exchange_single_rank_equals_plain_iteration():
    """world = 1: the box is this device's own; the loop must follow ccv_mppi_iterate_enqueue (V/S from the same sums)."""
    from ccv_mppi_path_tracker_amd import sharded
    w = configs.workload("C2", num_samples=4096)
    p = w.params
    path = helpers.oracle_path(w.path)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    a, b = MPPIController(p), MPPIController(p)
    xb = sharded.ExchangeBackend(b)
    assert xb.ok
    drv = sharded.ShardedMPPI(xb)
    for it in range(5):
        a.iterate_enqueue(state, p.dt, xr, yr, yaw[0], 5, it)
        drv.iterate(state, p.dt, xr, yr, yaw[0], 5, it)
    np.testing.assert_array_equal(a.get_nominal(), b.get_nominal())
    with pytest.raises(MPPIError):
        MPPIController(p).iterate_exchange_enqueue(state, p.dt, xr, yr, yaw[0], 5, 0)   # not connected


def _exchange_worker(rank, world, port, K, iters, out_dir):
    import torch
    import torch.distributed as dist
    from ccv_mppi_path_tracker_amd import sharded
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)   # both ranks on the one device of the box: the boxes are mapped across processes
    w = configs.workload("C2", num_samples=K)
    p = w.params
    off, k_local = sharded.shard_bounds(K, world, rank)
    ctl = MPPIController(p, num_samples=k_local, sample_offset=off)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        xb = sharded.ExchangeBackend(ctl)
        assert xb.ok, getattr(xb, "error", "")
        drv = sharded.ShardedMPPI(xb)
        path = helpers.oracle_path(w.path)
        state = start_state(p, path)
        xr, yr, yaw = helpers.oracle_window(p, path, state)
        for it in range(iters):
            drv.iterate(state, p.dt, xr, yr, yaw[0], 5, it)
        u = ctl.get_nominal()
    np.save(os.path.join(out_dir, "u%d.npy" % rank), u)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_exchange_processes_on_one_device(tmp_path, world):
    """2 / 4 ranks (processes) own an equal share of K each and exchange their partial vectors through IPC-mapped boxes,
    40 iterations back to back (both parities, many reuses of the slots): all must end with the same bits, equal to the
    single-handle loop up to summation order."""
    import torch.multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    K, iters = 4096, 40
    mp.spawn(_exchange_worker, args=(world, port, K, iters, str(tmp_path)), nprocs=world, join=True)
    u0 = np.load(tmp_path / "u0.npy")
    for r in range(1, world):
        np.testing.assert_array_equal(u0, np.load(tmp_path / ("u%d.npy" % r)))
    w = configs.workload("C2", num_samples=K)
    p = w.params
    path = helpers.oracle_path(w.path)
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    whole = MPPIController(p)
    for it in range(iters):
        whole.iterate_enqueue(state, p.dt, xr, yr, yaw[0], 5, it)
    np.testing.assert_allclose(u0, whole.get_nominal(), rtol=1e-7, atol=1e-10)


# --------------------------------------------------------------------------------------------------------------
# exact window pruning (pc_prune_window): the loop over the pruned hull must give the same bits as the full loop
# --------------------------------------------------------------------------------------------------------------
def _awkward_window(kind, H, state, rng):
    if kind == "path":
        return None
    if kind == "random":
        return rng.uniform(-3, 3, H), rng.uniform(-3, 3, H)
    if kind == "circle":
        a = np.linspace(0, 2 * np.pi, H, endpoint=False)
        return state[0] + 1.5 * np.cos(a), state[1] + 1.5 * np.sin(a)
    if kind == "duplicates":
        xr, yr = np.full(H, 1.0), np.full(H, 0.5)
        xr[::7] += 0.8
        return xr, yr
    if kind == "far":
        return np.linspace(150.0, 160.0, H), np.linspace(-90.0, -80.0, H)
    if kind == "line_behind":
        return state[0] - np.linspace(0.0, 4.0, H), np.full(H, state[1] + 0.2)
    if kind == "nan_point":
        xr, yr = rng.uniform(-2, 2, H), rng.uniform(-2, 2, H)
        xr[H // 3] = np.nan
        yr[2 * H // 3] = np.nan
        return xr, yr
    if kind == "inf_point":
        xr, yr = np.linspace(0.0, 5.0, H), np.zeros(H)
        xr[H // 2] = np.inf
        yr[H - 2] = -np.inf
        return xr, yr
    if kind == "huge":
        return np.linspace(0.0, 5.0, H) * 1e150, np.linspace(-1.0, 1.0, H) * 1e150   # overflows fp32 (and d^2 in fp64)
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["path", "random", "circle", "duplicates", "far", "line_behind", "nan_point", "inf_point", "huge"])
@pytest.mark.parametrize("wl,K,H,kernel", [("C2", 1000, 50, None), ("C2", 1000, 50, "solo"), ("C3", 640, 50, None),
                                           ("C4", 512, 80, "solo"), ("C4", 256, 128, None), ("C2", 320, 67, "pc")])
def test_pruned_loop_equals_full_loop_bitwise(monkeypatch, kind, wl, K, H, kernel):
    """CCV_MPPI_PRUNE=1 (dominance test + loop over the hull of the survivors) against CCV_MPPI_PRUNE=0 (every window point
    for every state), same kernel: per-sample costs, weights and u* must be the same bits -- the test only ever removes
    points that cannot be the minimum -- for path-like windows and for windows that are unordered, closed, degenerate, far
    away or contain NaN / infinite / huge coordinates.  Three iterations (the warm start moves the samples)."""
    w = configs.workload(wl, num_samples=K, horizon=H)
    p = w.params
    rng = np.random.default_rng(3)
    path = helpers.oracle_path(w.path)
    state = np.zeros(p.nstate)
    state[:3] = path[0][0] + 0.4, path[1][0] - 0.3, 0.3
    win = _awkward_window(kind, H, state, rng)
    if win is None:
        xr, yr, yaw = helpers.oracle_window(p, path, state)
        yaw0 = yaw[0]
    else:
        (xr, yr), yaw0 = win, 0.1
    if kernel:
        monkeypatch.setenv("CCV_MPPI_KERNEL", kernel)
    monkeypatch.setenv("CCV_MPPI_PRUNE", "1")
    a = MPPIController(p)
    monkeypatch.setenv("CCV_MPPI_PRUNE", "0")
    b = MPPIController(p)
    for it in range(3):
        ua, sa = a.iterate(state, p.dt, xr, yr, yaw0, 4, it)
        ub, sb = b.iterate(state, p.dt, xr, yr, yaw0, 4, it)
        np.testing.assert_array_equal(a.read_costs(), b.read_costs())
        np.testing.assert_array_equal(ua, ub)
        assert sa.sum_w == sb.sum_w or (np.isnan(sa.sum_w) and np.isnan(sb.sum_w))
        if not np.all(np.isfinite(ua)):   # (all weights underflowed / NaN: the next warm start is NaN for both -- SURVEY Q4)
            break
    if kind in ("path", "random", "circle", "line_behind") and wl != "C4":   # (full body far from its path: every weight underflows)
        assert np.all(np.isfinite(ua))


def test_pruned_loop_with_nonfinite_positions(monkeypatch):
    """Samples whose positions are NaN or infinite (here: through an infinite speed bound and a NaN in the warm start) end at
    the 100 m gate value whatever the loop covers; the finite lanes of the same wave must be unaffected by them."""
    p = configs.diff_drive_defaults(512, 40).with_(u_min=(-1e308, -2.0), u_max=(1e308, 2.0), control_noise=1e306)
    path = helpers.oracle_path("sinusoid")
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    monkeypatch.setenv("CCV_MPPI_PRUNE", "1")
    a = MPPIController(p)
    monkeypatch.setenv("CCV_MPPI_PRUNE", "0")
    b = MPPIController(p)
    a.iterate(state, p.dt, xr, yr, yaw[0], 9, 0)
    b.iterate(state, p.dt, xr, yr, yaw[0], 9, 0)
    np.testing.assert_array_equal(a.read_costs(), b.read_costs())


# --------------------------------------------------------------------------------------------------------------
# lifetime: ccv_mppi_destroy gives everything back
# --------------------------------------------------------------------------------------------------------------
def test_create_destroy_returns_all_device_memory():
    """create -> device-resident loop + direct exchange set up -> a few iterations -> destroy, 200 times: the free device
    memory (hipMemGetInfo) must come back to where it started.  (Round 1 leaked the resident frame, path, trace and the
    exchange box: ~0.7 MB per handle.)"""
    import torch
    from ccv_mppi_path_tracker_amd import sharded
    w = configs.workload("C4", num_samples=2048, horizon=40)
    p = w.params
    px, py = amd.make_path(w.path)
    state = start_state(p, (px, py))

    def cycle():
        g = MPPIController(p)
        g.resident_set_path(px, py)
        g.resident_set_pose(state)
        xb = sharded.ExchangeBackend(g)
        assert xb.ok
        g.resident_step_enqueue(p.dt, 1, 0, advance=False)
        g.resident_step_exchange_enqueue(p.dt, 1, 1, advance=True)
        g.read_top_candidates(4)        # (allocates the read-back scratch)
        assert np.all(np.isfinite(g.get_nominal()))
        g.close()

    for _ in range(3):   # runtime pools settle
        cycle()
    torch.cuda.synchronize()
    free0, _total = torch.cuda.mem_get_info()
    for _ in range(200):
        cycle()
    torch.cuda.synchronize()
    free1, _total = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 * 2**20, "device memory shrank by %.1f MiB over 200 create/destroy cycles" % ((free0 - free1) / 2**20)


def test_exchange_peer_timeout_is_reported(monkeypatch):
    """Two ranks (two handles of this process), only rank 0 ever iterates: its update kernel gives up after the timeout
    (shortened here), the controls are NaN, and the next synchronisation says why instead of returning OK."""
    monkeypatch.setenv("CCV_MPPI_EXCHANGE_TIMEOUT_MS", "200")
    p = configs.workload("C2", num_samples=1024).params
    path = helpers.oracle_path("sinusoid")
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    a, b = MPPIController(p, num_samples=512), MPPIController(p, num_samples=512, sample_offset=512)
    blobs = [a.exchange_create(2, 0), b.exchange_create(2, 1)]
    a.exchange_connect(blobs)
    b.exchange_connect(blobs)
    assert a.exchange_info() == {"world": 2, "rank": 0, "fine_grained": a.exchange_info()["fine_grained"], "connected": True}
    a.iterate_exchange_enqueue(state, p.dt, xr, yr, yaw[0], 5, 0)   # rank 1 never arrives
    with pytest.raises(MPPIError) as e:
        a.synchronize()
    assert e.value.code == capi.ERR_TIMEOUT
    with pytest.raises(MPPIError) as e:
        a.get_nominal()
    assert e.value.code == capi.ERR_TIMEOUT
    # two ranks that do both arrive are fine (fresh handles; the flag of the old ones is sticky)
    c, d = MPPIController(p, num_samples=512), MPPIController(p, num_samples=512, sample_offset=512)
    blobs = [c.exchange_create(2, 0), d.exchange_create(2, 1)]
    c.exchange_connect(blobs)
    d.exchange_connect(blobs)
    with pytest.raises(MPPIError):
        c.exchange_connect(blobs)   # already connected
    for it in range(6):
        c.iterate_exchange_enqueue(state, p.dt, xr, yr, yaw[0], 5, it)
        d.iterate_exchange_enqueue(state, p.dt, xr, yr, yaw[0], 5, it)
    uc, ud = c.get_nominal(), d.get_nominal()
    np.testing.assert_array_equal(uc, ud)
    whole = MPPIController(p)
    for it in range(6):
        whole.iterate_enqueue(state, p.dt, xr, yr, yaw[0], 5, it)
    np.testing.assert_allclose(uc, whole.get_nominal(), rtol=1e-8, atol=1e-12)
