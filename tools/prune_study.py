#!/usr/bin/env python3
"""How much of the reference window can an exact, wave-level pruning test discard?  (CPU study, numpy only.)

pc_consume (csrc/mppi_rollout_pc.h) skips window points that cannot be the nearest one for any sample of a wave: bounds
of the linear form f_j(p) = a_j px + b_j py + c_j over a bounding box of positions, then the hull of the surviving
indices.  This script replays a diff-drive MPPI loop on the C2 workload in numpy and reports, per time block of 8 steps,
the fraction of (state, window point) pairs that remain when the box is taken over 8 / 4 / 2 / 1 consecutive states of
the 64 samples of a wave -- and the per-sample ideal (only the true nearest point's group of 4).
"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402


def rollout(p, state, u):
    K, T = u.shape[0], u.shape[1] + 1
    x = np.empty((K, T))
    y = np.empty((K, T))
    yaw = np.full(K, state[2])
    x[:, 0], y[:, 0] = state[0], state[1]
    for t in range(T - 1):
        x[:, t + 1] = x[:, t] + u[:, t, 0] * np.cos(yaw) * p.dt
        y[:, t + 1] = y[:, t] + u[:, t, 0] * np.sin(yaw) * p.dt
        yaw = yaw + u[:, t, 1] * p.dt
    return x, y


def hull_fraction(px, py, a, b, c, group):
    """px, py: [64][n] positions (relative to the pose) of one wave's block; hull of survivors of the box test taken over
    `group` consecutive states; returns the number of (state, point) pairs the loop would evaluate (groups of 4 points)."""
    H = len(a)
    H4 = (H + 3) & ~3
    n = px.shape[1]
    pairs = 0
    for s0 in range(0, n, group):
        sx, sy = px[:, s0:s0 + group], py[:, s0:s0 + group]
        xlo, xhi, ylo, yhi = sx.min(), sx.max(), sy.min(), sy.max()
        ax0, ax1, by0, by1 = a * xlo, a * xhi, b * ylo, b * yhi
        lb = c + (np.minimum(ax0, ax1) + np.minimum(by0, by1))
        ub = c + (np.maximum(ax0, ax1) + np.maximum(by0, by1))
        keep = np.nonzero(~(lb > ub.min()))[0]
        lo, hi = keep[0] & ~3, (keep[-1] | 3) + 1
        pairs += (min(hi, H4) - lo) * sx.shape[1]
    return pairs


def main():
    w = configs.workload("C2", num_samples=4096)
    p = w.params
    px_, py_ = amd.make_path(w.path)
    rng = np.random.default_rng(0)
    K, H = p.num_samples, p.horizon
    nominal = np.zeros((H - 1, 2))
    span = max(1, len(px_) - 2 * H // 3)
    tot = {g: 0 for g in (8, 4, 2, 1)}
    ideal = full = 0
    per_block = {g: np.zeros((H + 7) // 8) for g in (8, 4, 2, 1)}
    nit = 24
    for it in range(nit):
        j = int(it * p.v_ref * p.dt / p.resolution) % span
        yaw = np.arctan2(py_[j + 1] - py_[j], px_[j + 1] - px_[j])
        state = np.array([px_[j] + rng.normal(0, 0.03), py_[j] + rng.normal(0, 0.03), yaw + rng.normal(0, 0.05)])
        _, xr, yr, _ = amd.calc_ref_path(px_, py_, state[0], state[1], p.v_ref, p.dt, p.resolution, H)
        u = np.clip(rng.normal(0, p.control_noise, size=(K, H - 1, 2)) + nominal, p.u_min[:2], p.u_max[:2])
        x, y = rollout(p, state, u)
        d2 = ((x[:, :, None] - xr) ** 2 + (y[:, :, None] - yr) ** 2)
        nearest = d2.argmin(axis=2)
        cost = p.path_weight * np.minimum(d2.min(axis=2), 1e4).sum(axis=1) + p.v_weight * ((u[:, :, 0] - p.v_ref) ** 2).sum(axis=1)
        wgt = np.exp(-(cost - cost.min()) / p.lam)
        nominal = np.einsum("k,ktd->td", wgt / wgt.sum(), u)
        if it < 8:
            continue   # let the warm start settle
        xl, yl = xr - state[0], yr - state[1]
        a, b, c = -2 * xl, -2 * yl, xl * xl + yl * yl
        for wv in range(K // 64):
            sl = slice(wv * 64, wv * 64 + 64)
            for blk in range((H + 7) // 8):
                t0, t1 = blk * 8, min(blk * 8 + 8, H)
                bx, by = x[sl, t0:t1] - state[0], y[sl, t0:t1] - state[1]
                for g in tot:
                    n = hull_fraction(bx, by, a, b, c, g)
                    tot[g] += n
                    per_block[g][blk] += n
                full += ((H + 3) & ~3) * (t1 - t0)
                # per (state) ideal at wave granularity: hull of the 64 samples' true nearest points, in groups of 4
                nn = nearest[sl, t0:t1]
                ideal += (((nn.max(axis=0) | 3) + 1) - (nn.min(axis=0) & ~3)).sum()
    print("pairs evaluated / all pairs (window of %d points, padded to %d):" % (H, (H + 3) & ~3))
    for g in tot:
        print("  box over %d state(s) x 64 samples: %.3f   per block: %s" % (
            g, tot[g] / full, np.array2string(per_block[g] / per_block[8].sum() * tot[8] / full * len(per_block[g]), precision=2)))
    print("  hull of the true nearest points per state (lower bound for any wave-level test): %.3f" % (ideal / full))


if __name__ == "__main__":
    main()
