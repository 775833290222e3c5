"""CPU tests of the oracle itself (no GPU): golden vectors, independent numpy cross-checks of the restatement,
quirk semantics, noise-spec known answers."""
import glob
import os

import numpy as np
import pytest

import helpers
from ccv_mppi_path_tracker_amd import configs
from oracle import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {name: (p, kind) for name, p, kind in helpers.small_cases()}


def test_golden_files_cover_all_cases():
    have = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLD, "*.npz")))
    assert have == sorted(CASES)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    p, kind = CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    px, py = helpers.oracle_path(kind)
    np.testing.assert_array_equal(px, g["path_x"])
    np.testing.assert_array_equal(py, g["path_y"])
    o = helpers.oracle_for(p)
    for it in range(3):
        pre = "it%d_" % it
        state = g[pre + "x0"]
        xr, yr, yaw = helpers.oracle_window(p, (px, py), state)
        np.testing.assert_array_equal(xr, g[pre + "x_ref"])
        np.testing.assert_array_equal(yr, g[pre + "y_ref"])
        assert yaw[0] == g[pre + "yaw_ref0"]
        o.set_nominal(g[pre + "u_in"])
        u = o.iterate(state, p.dt, xr, yr, yaw[0], seed=int(g[pre + "seed"]), rng="mt19937")
        # same libstdc++/glibc => identical bits; the tolerance only absorbs a different libm build
        np.testing.assert_allclose(u, g[pre + "u_out"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(o.costs(), g[pre + "costs"], rtol=1e-13)
        np.testing.assert_allclose(o.sum_w(), g[pre + "sum_w"], rtol=1e-11)
        sel = [0, 1, p.num_samples - 1]
        np.testing.assert_allclose(o.get_controls()[sel], g[pre + "controls_sel"], rtol=1e-14, atol=1e-16)
        np.testing.assert_allclose(o.states("x")[sel], g[pre + "x_sel"], rtol=1e-13, atol=1e-15)


# ---- independent numpy restatement of libstdc++'s mt19937 -> normal_distribution draw order (Q8) ----

def _std_normal_stream(raw):
    """libstdc++ normal_distribution<double> (polar method) fed by generate_canonical<double,53> over 32-bit words."""
    pos = [0]

    def canon():
        w0, w1 = float(raw[pos[0]]), float(raw[pos[0] + 1])
        pos[0] += 2
        r = (w0 + w1 * 4294967296.0) / 18446744073709551616.0
        return np.nextafter(1.0, 0.0) if r >= 1.0 else r

    class Dist:
        def __init__(self, mean, sd):
            self.mean, self.sd, self.saved, self.has = mean, sd, 0.0, False

        def __call__(self):
            if self.has:
                self.has = False
                ret = self.saved
            else:
                while True:
                    x = 2.0 * canon() - 1.0
                    y = 2.0 * canon() - 1.0
                    r2 = x * x + y * y
                    if not (r2 > 1.0 or r2 == 0.0):
                        break
                mult = np.sqrt(-2.0 * np.log(r2) / r2)
                self.saved, self.has = x * mult, True
                ret = y * mult
            return ret * self.sd + self.mean
    return Dist


@pytest.mark.parametrize("model,udim", [("diff_drive", 2), ("steering_diff_drive", 3), ("full_body", 5)])
def test_mt19937_draw_order_matches_numpy_restatement(model, udim):
    K, H, seed = 7, 5, 1234   # odd K: the cached second variate is dropped at the end of every t (fresh objects per t)
    p = {"diff_drive": configs.diff_drive_defaults, "steering_diff_drive": configs.steering_defaults,
         "full_body": configs.full_body_defaults}[model](K, H)
    p = p.with_(u_min=tuple([-1e9] * udim), u_max=tuple([1e9] * udim))
    o = helpers.oracle_for(p)
    nominal = np.arange((H - 1) * udim, dtype=np.float64).reshape(H - 1, udim) * 0.01
    o.set_nominal(nominal)
    o.sampling(seed, rng="mt19937")
    got = o.get_controls()
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)           # init_genrand(seed) == std::mt19937(seed)
    raw = bg.random_raw(20000)
    Dist = _std_normal_stream(raw)
    want = np.zeros((K, H - 1, udim))
    for t in range(H - 1):
        dists = [Dist(nominal[t, d], p.control_noise) for d in range(udim)]
        for i in range(K):
            for d in range(udim):
                want[i, t, d] = dists[d]()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-15)


# ---- independent numpy restatement of the cost, per model ----

def _min_dist(x, y, xr, yr):
    return min(100.0, float(np.min(np.sqrt((x - xr) ** 2 + (y - yr) ** 2))))


@pytest.mark.parametrize("name", sorted(CASES))
def test_cost_matches_numpy_formula(name):
    p, kind = CASES[name]
    o = helpers.oracle_for(p)
    path = helpers.oracle_path(kind)
    state = np.zeros(p.nstate)
    state[:2] = path[0][3], path[1][3] + 0.05
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    o.iterate(state, p.dt, xr, yr, yaw[0], seed=7, rng="mt19937")
    u, X, Y = o.get_controls(), o.states("x"), o.states("y")
    costs = o.costs()
    H = p.horizon
    for i in (0, 3, p.num_samples - 1):
        if p.model != "full_body":
            v = np.append(u[i, :, 0], 0.0)   # Q1: phantom control at index H-1 reads 0.0
            c = sum(p.path_weight * _min_dist(X[i, t], Y[i, t], xr, yr) ** 2 + p.v_weight * (v[t] - p.v_ref) ** 2
                    for t in range(H))
        else:
            zy = o.states("zmp_y")
            zw = 0.0 if p.roll_off else p.zmp_weight
            rw = 0.0 if p.roll_off else p.roll_v_weight
            c = p.yaw_weight * (state[2] - yaw[0]) ** 2
            for t in range(H - 2):
                c += p.path_weight * _min_dist(X[i, t], Y[i, t], xr, yr) ** 2
                c += p.v_weight * (u[i, t, 0] - p.v_ref) ** 2
                c += zw * zy[i, t] ** 2
                c += rw * (u[i, t + 1, 3] - u[i, t, 3]) ** 2
                if u[i, t, 0] < 0.0:
                    c += p.back_weight * u[i, t, 0] ** 2
        assert abs(c - costs[i]) <= 1e-11 * abs(c)
    w = np.exp(-costs / p.lam)
    np.testing.assert_allclose(o.sum_w(), w.sum(), rtol=1e-12)
    np.testing.assert_allclose(o.weights(), w / w.sum(), rtol=1e-12)
    np.testing.assert_allclose(o.get_nominal(), np.einsum("i,itd->td", w / w.sum(), u), rtol=1e-10, atol=1e-14)


def test_full_body_zmp_closed_form():
    """zmp_y closed form of SURVEY.md a6 vs the vector restatement (fb:468-486, 597-603)."""
    p = configs.workload("C4").params.with_(num_samples=32, horizon=20)
    o = helpers.oracle_for(p)
    path = helpers.oracle_path("dkan")
    state = np.array([0.0, 0.0, 0.0, 0.05, -0.02])
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    o.iterate(state, p.dt, xr, yr, yaw[0], seed=3, rng="mt19937")
    u, roll, pitch, zy = o.get_controls(), o.states("roll"), o.states("pitch"), o.states("zmp_y")
    m, L, dt = 60.0, 0.8075 / 2, p.dt
    Ixx = (m * (0.208 ** 2 + 0.8075 ** 2)) / 12 + m * L * L
    for i in (0, 31):
        for t in range(p.horizon - 2):
            v, w, d = u[i, t, 0], u[i, t, 1], u[i, t, 2]
            da = (u[i, t + 1, 0] - v) / dt
            ay = da * np.sin(d) + v * w * np.cos(d)
            want = (m * (9.8 * L * np.sin(roll[i, t]) + L * np.cos(pitch[i, t]) * np.cos(roll[i, t]) * ay)
                    - Ixx * (u[i, t + 1, 3] - u[i, t, 3]) / dt) / (-588.0)
            assert abs(want - zy[i, t]) <= 1e-12 * max(1.0, abs(want))


def test_rollout_states_euler():
    p = configs.workload("C3").params.with_(num_samples=8, horizon=12)
    o = helpers.oracle_for(p)
    path = helpers.oracle_path("sinusoid")
    state = np.array([0.3, -0.1, 0.2])
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    o.iterate(state, p.dt, xr, yr, yaw[0], seed=11, rng="mt19937")
    u, X, Y, Yaw = o.get_controls(), o.states("x"), o.states("y"), o.states("yaw")
    for i in range(8):
        s = state.copy()
        for t in range(p.horizon - 1):
            assert (X[i, t], Y[i, t], Yaw[i, t]) == (s[0], s[1], s[2])
            s = helpers.plant(p.model, s, u[i, t], p.dt)


# ---- host prologue quirks ----

def test_ref_window_truncation_and_tail_repeat():
    px, py = helpers.oracle_path("sinusoid")
    assert len(px) == 101
    idx, xr, yr, yaw = O.calc_ref_path(px, py, 0.0, 0.0, 0.8, 0.1, 0.1, 15)   # stride 0.8 -> 0,0,1,2,3,4,4,5,...
    want = [int(0 + i * (0.8 * 0.1 / 0.1)) for i in range(15)]
    assert want[:3] == [0, 0, 1]
    np.testing.assert_array_equal(xr, px[want])
    assert yaw[0] == 0.0                      # Q11: first two window points coincide -> atan2(0,0) = 0
    idx, xr, yr, yaw = O.calc_ref_path(px, py, px[95], py[95], 1.2, 0.1, 0.1, 50)
    assert idx == 95
    assert np.all(xr[6:] == px[-1]) and np.all(yr[6:] == py[-1])   # Q12: past the end repeats the last pose
    idx, _, _, _ = O.calc_ref_path(px, py, 500.0, 500.0, 1.2, 0.1, 0.1, 50)
    assert idx == 0                           # nothing within the 100 m gate


def test_path_generators():
    assert len(helpers.oracle_path("straight")[0]) == 101
    x, y = O.path_cosine(A=(1.5, 0, 0), omega=(0.127, 0, 0), delta=(0, 0, 0), course_length=20.0)
    assert len(x) == 200 and y[0] == 0.0
    dx, dy = helpers.oracle_path("dkan")
    assert len(dx) == 437 and (dx[0], dy[0]) == (0.0, 0.0) and dy[-1] == 8.0
    assert np.all(np.diff(dx[:178]) > 0) and np.all(dy[:178] == 0.0)


# ---- noise spec ----

def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    assert O.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_normal_pair_against_libm():
    rng = np.random.default_rng(5)
    words = rng.integers(0, 2 ** 32, size=(4000, 2), dtype=np.uint64)
    edge = np.array([[0, 0], [1, 0], [0xFFFFFFFF, 0xFFFFFFFF], [0x80000000, 0x40000000], [0xB504F334, 0xC0000000]],
                    dtype=np.uint64)
    for a, b in np.vstack([words, edge]):
        a, b = int(a), int(b)
        z0, z1 = O.normal_pair(a, b)
        u1 = max(a, 1) / 2.0 ** 32
        r = np.sqrt(-2.0 * np.log(u1))
        frac = ((b & 0x3FFFFFFF) - 2 ** 29) / 2.0 ** 30
        th = (np.pi / 2) * ((b >> 30) + frac)
        assert abs(float(z0) - r * np.cos(th)) < 4e-6 * max(1.0, r)
        assert abs(float(z1) - r * np.sin(th)) < 4e-6 * max(1.0, r)


def test_normals_are_standard_normal():
    from scipy import stats
    z = O.normals(2024, 3, 100, 2048, 196).astype(np.float64).ravel()
    assert abs(z.mean()) < 5e-3 and abs(z.var() - 1.0) < 1e-2 and abs((z ** 4).mean() - 3.0) < 0.05
    assert stats.kstest(z, "norm").pvalue > 1e-3
    # independent across samples / indices: lag correlations vanish
    zz = O.normals(2024, 3, 0, 512, 128).astype(np.float64)
    assert abs(np.corrcoef(zz[:-1].ravel(), zz[1:].ravel())[0, 1]) < 0.01
    assert abs(np.corrcoef(zz[:, :-1].ravel(), zz[:, 1:].ravel())[0, 1]) < 0.01


def test_philox_sampling_is_shard_invariant():
    p = configs.workload("C2").params.with_(num_samples=64, horizon=10)
    full, lo, hi = helpers.oracle_for(p), helpers.oracle_for(p, 40), helpers.oracle_for(p, 24)
    full.sampling(9, rng="philox", iteration=2)
    lo.sampling(9, rng="philox", iteration=2, k_offset=0)
    hi.sampling(9, rng="philox", iteration=2, k_offset=40)
    np.testing.assert_array_equal(full.get_controls(), np.concatenate([lo.get_controls(), hi.get_controls()]))
