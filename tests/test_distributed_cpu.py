"""N > 1 path on CPU: two gloo ranks shard K, all-reduce the [sum w, sum w*u] partials and must reproduce the
single-rank result.  The compute backend here is the CPU oracle (the GPU backend is exercised by
tests/test_gpu_parity.py::test_shard_invariance_via_partials and by bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ccv_mppi_path_tracker_amd import configs, sharded  # noqa: E402


class OraclePartials:
    """local_partials/apply over the oracle: S = sum exp(-cost/lambda), V = sum w*u over this rank's samples."""

    def __init__(self, p, k_offset, k_local):
        import helpers
        self.p, self.k_offset = p, k_offset
        self.o = helpers.oracle_for(p, k_local)

    def local_partials(self, x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration):
        o = self.o
        o.sampling(seed, rng="philox", iteration=iteration, k_offset=self.k_offset)
        o.predict_States(x0, dt)
        o.calc_Weights(x_ref, y_ref, yaw_ref0)
        w = np.exp(-o.costs() / self.p.lam)
        v = np.einsum("i,itd->td", w, o.get_controls()).ravel()
        return torch.from_numpy(np.concatenate([[w.sum()], v]))

    def apply(self, reduced):
        r = reduced.numpy()
        self.o.set_nominal((r[1:] / r[0]).reshape(self.p.horizon - 1, self.p.udim))

    # the resident closed loop (ShardedMPPI.iterate_resident): path and pose held by the backend, prologue per tick
    def resident_setup(self, path, state):
        self.path, self.state = path, np.array(state, dtype=np.float64)

    def local_partials_resident(self, dt, seed, iteration, advance):
        import helpers
        if advance:
            self.state = helpers.plant(self.p.model, self.state, self.o.get_nominal()[0], dt)
        xr, yr, yaw = helpers.oracle_window(self.p, self.path, self.state)
        return self.local_partials(self.state, dt, xr, yr, yaw[0], seed, iteration)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, model, out_path, resident=False):
    import helpers
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = configs.workload(model, num_samples=256, horizon=20)
    p = w.params
    off, k_local = sharded.shard_bounds(p.num_samples, world, rank)
    drv = sharded.ShardedMPPI(OraclePartials(p, off, k_local))
    path = helpers.oracle_path(w.path)
    state = np.zeros(p.nstate)
    state[:2] = path[0][0], path[1][0] + 0.05
    outs = []
    if resident:
        drv.backend.resident_setup(path, state)
    for it in range(3):
        if resident:   # pose and window are the backend's business
            drv.iterate_resident(p.dt, 11, it, advance=it > 0)
        else:
            xr, yr, yaw = helpers.oracle_window(p, path, state)
            drv.iterate(state, p.dt, xr, yr, yaw[0], 11, it)
        u = drv.backend.o.get_nominal()
        outs.append(u)
        state = helpers.plant(p.model, state, u[0], p.dt)
    # every rank must hold the same u*
    t = torch.from_numpy(np.stack(outs))
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    for g in gathered:
        assert torch.equal(g, gathered[0])
    if rank == 0:
        np.save(out_path, np.stack(outs))
    dist.destroy_process_group()


@pytest.mark.parametrize("model,resident", [("C2", False), ("C4", False), ("C2", True)])
def test_two_gloo_ranks_reproduce_single_rank(tmp_path, model, resident):
    import helpers
    out = str(tmp_path / "u.npy")
    mp.spawn(_worker, args=(2, _free_port(), model, out, resident), nprocs=2, join=True)
    got = np.load(out)
    # single rank, same global sample ids
    w = configs.workload(model, num_samples=256, horizon=20)
    p = w.params
    o = helpers.oracle_for(p)
    path = helpers.oracle_path(w.path)
    state = np.zeros(p.nstate)
    state[:2] = path[0][0], path[1][0] + 0.05
    for it in range(3):
        xr, yr, yaw = helpers.oracle_window(p, path, state)
        u = o.iterate(state, p.dt, xr, yr, yaw[0], seed=11, rng="philox", iteration=it)
        np.testing.assert_allclose(got[it], u, rtol=1e-11, atol=1e-14)
        o.set_nominal(got[it])
        state = helpers.plant(p.model, state, got[it][0], p.dt)


def test_shard_bounds():
    assert sharded.shard_bounds(524288, 8, 3) == (3 * 65536, 65536)
    with pytest.raises(ValueError):
        sharded.shard_bounds(1000, 3, 0)
    parts = [np.array([2.0, 4.0, 6.0]), np.array([1.0, 1.0, 3.0])]
    u, tot = sharded.combine_partials(parts)
    np.testing.assert_array_equal(tot, [3.0, 5.0, 9.0])
    np.testing.assert_allclose(u, [5.0 / 3.0, 3.0])
