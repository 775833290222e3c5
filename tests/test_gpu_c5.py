"""GPU tests (-m gpu) of BASELINE config C5 -- diff_drive K = 524 288 sharded over 8 devices -- on the ONE device a test box
has: the eight shards are eight handles (global sample ids g * 65 536) whose partial vectors are summed here, against one
K = 524 288 handle; the direct exchange runs with world = 8 (four processes x two handles on the one device); and the RCCL
branch of the sharded driver runs once for real on a one-rank `nccl` process group.  What these cannot show is the cost of
the exchange over xGMI links (the driver's 8-GPU scaling run does).
"""
import os
import socket

import numpy as np
import pytest

import helpers
from ccv_mppi_path_tracker_amd import capi, configs, sharded
from ccv_mppi_path_tracker_amd.controller import MPPIController

pytestmark = pytest.mark.gpu

TOL_COST = 1e-9


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(gpu_required):
    capi.load()


def _inputs(p, kind, lateral=0.05):
    path = helpers.oracle_path(kind)
    s = np.zeros(p.nstate)
    s[0], s[1] = path[0][0], path[1][0] + lateral
    xr, yr, yaw = helpers.oracle_window(p, path, s)
    return s, xr, yr, yaw


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("forced", [None, "solo"])
def test_c5_eight_shards_equal_one_handle(monkeypatch, forced):
    """C5 as the 8-GPU run shards it (sample_offset = g * 65 536, C2 parameters) against ONE handle with K = 524 288.
    Default kernels: the shards run the four-wave kernel, the whole runs the one-wave kernel -- same samples and states,
    costs equal up to the order of a sample's cost terms.  With the one-wave kernel forced for the shards as well, every
    per-sample cost is the same bits.  Either way the sum of the eight partial vectors gives the whole's u*."""
    import torch
    w = configs.workload("C2")
    p = w.params
    G, kl = 8, 65536
    K = G * kl
    state, xr, yr, yaw = _inputs(p, w.path)
    nominal = np.random.default_rng(5).normal(0, 0.2, size=(p.horizon - 1, p.udim))
    whole = MPPIController(p, num_samples=K)
    whole.set_nominal(nominal)
    u_whole, st = whole.iterate(state, p.dt, xr, yr, yaw[0], 42, 3)
    c_whole = whole.read_costs()
    if forced:
        monkeypatch.setenv("CCV_MPPI_KERNEL", forced)
    n = whole.partials_size()
    tot = torch.zeros(n, dtype=torch.float64, device="cuda")
    part = torch.zeros(n, dtype=torch.float64, device="cuda")
    for g in range(G):
        sh = MPPIController(p, num_samples=kl, sample_offset=g * kl)
        sh.set_nominal(nominal)
        sh.iterate_partials_enqueue(state, p.dt, xr, yr, yaw[0], 42, 3, part.data_ptr())
        sh.synchronize()
        tot += part
        c = sh.read_costs()
        if forced:
            np.testing.assert_array_equal(c, c_whole[g * kl:(g + 1) * kl])
        else:
            np.testing.assert_allclose(c, c_whole[g * kl:(g + 1) * kl], rtol=1e-12)
        if g in (0, 7):   # same noise for the same global sample id
            np.testing.assert_array_equal(sh.read_controls(kl - 64, 64), whole.read_controls((g + 1) * kl - 64, 64))
        sh.close()
    t = tot.cpu().numpy()
    assert abs(t[0] - st.sum_w) <= 1e-11 * st.sum_w
    u_sharded = (t[1:] / t[0]).reshape(p.horizon - 1, p.udim)
    assert helpers.rel_err(u_sharded, u_whole) < 1e-9


def test_c5_full_size_blocks_rescored_by_the_oracle():
    """K = 524 288 on one handle (the one-wave kernel): three blocks of 192 samples -- first, middle, last -- re-scored by
    the oracle at their global offsets; the reduction recomputed on the host; determinism."""
    w = configs.workload("C2")
    p = w.params
    K = 524288
    state, xr, yr, yaw = _inputs(p, w.path)
    nominal = np.random.default_rng(1).normal(0, 0.2, size=(p.horizon - 1, p.udim))
    g = MPPIController(p, num_samples=K)
    g.set_nominal(nominal)
    u1, st = g.iterate(state, p.dt, xr, yr, yaw[0], 42, 7)
    for first in (0, K // 2 - 96, K - 192):
        o = helpers.oracle_for(p, 192)
        o.set_nominal(nominal)
        o.iterate(state, p.dt, xr, yr, yaw[0], seed=42, rng="philox", iteration=7, k_offset=first)
        np.testing.assert_array_equal(g.read_controls(first, 192), o.get_controls())
        c_o = o.costs()
        assert np.max(np.abs(g.read_costs(first, 192) - c_o) / c_o) < TOL_COST
        xy = g.read_candidates(first, 192, 1)
        np.testing.assert_allclose(xy[..., 0], o.states("x"), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(xy[..., 1], o.states("y"), rtol=1e-12, atol=1e-12)
    c = g.read_costs()
    wts = np.exp(-c / p.lam)
    assert abs(st.sum_w - wts.sum()) <= 1e-10 * wts.sum()
    assert st.min_cost == c.min() and st.max_cost == c.max() and st.n_zero_weight == int((wts == 0).sum())
    top = np.argsort(c)[:64]
    ctrl_top = np.stack([g.read_controls(int(i), 1)[0] for i in top])
    u_host = np.einsum("i,itd->td", wts[top] / wts.sum(), ctrl_top)
    resid = 1.0 - wts[top].sum() / wts.sum()
    umax = max(np.max(np.abs(p.u_min)), np.max(np.abs(p.u_max)))
    assert np.max(np.abs(u_host - u1)) <= resid * umax + 1e-9
    g.set_nominal(nominal)
    u2, st2 = g.iterate(state, p.dt, xr, yr, yaw[0], 42, 7)
    np.testing.assert_array_equal(u1, u2)
    assert st.sum_w == st2.sum_w


# ---- direct exchange, world = 8: four processes x two handles, all on the one device -------------------------------------
def _exchange8_worker(proc, nproc, per, port, K, iters, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=proc, world_size=nproc)
    torch.cuda.set_device(0)
    world = nproc * per
    w = configs.workload("C2", num_samples=K)
    p = w.params
    ctls, blobs = [], []
    for j in range(per):
        rank = proc * per + j
        off, k_local = sharded.shard_bounds(K, world, rank)
        c = MPPIController(p, num_samples=k_local, sample_offset=off)
        ctls.append(c)
        blobs.append(c.exchange_create(world, rank))
    gathered = [None] * nproc
    dist.all_gather_object(gathered, blobs)
    handles = [b for sub in gathered for b in sub]   # rank order
    for c in ctls:
        c.exchange_connect(handles)
        info = c.exchange_info()
        assert info["world"] == world and info["connected"]
    state, xr, yr, yaw = _inputs(p, w.path, lateral=0.0)
    dist.barrier()
    for it in range(iters):
        for c in ctls:   # each handle on its own stream; nothing blocks on the host
            c.iterate_exchange_enqueue(state, p.dt, xr, yr, yaw[0], 5, it)
    for j, c in enumerate(ctls):
        np.save(os.path.join(out_dir, "u%d.npy" % (proc * per + j)), c.get_nominal())
        np.save(os.path.join(out_dir, "fine%d.npy" % (proc * per + j)), np.array([int(c.exchange_info()["fine_grained"])]))
    dist.barrier()
    for c in ctls:
        c.close()
    dist.destroy_process_group()


def test_exchange_world_8_on_one_device(tmp_path):
    """kMaxRanks = 8, the C5 world size: eight ranks exchange their partial vectors through boxes mapped across processes
    (hipIpc) and inside a process (direct), 40 iterations back to back.  All eight must end with the same bits, equal to the
    single-handle loop up to summation order."""
    import torch.multiprocessing as mp
    nproc, per, K, iters = 4, 2, 8192, 40
    mp.spawn(_exchange8_worker, args=(nproc, per, _free_port(), K, iters, str(tmp_path)), nprocs=nproc, join=True)
    u0 = np.load(tmp_path / "u0.npy")
    assert np.all(np.isfinite(u0))
    for r in range(1, nproc * per):
        np.testing.assert_array_equal(u0, np.load(tmp_path / ("u%d.npy" % r)))
    w = configs.workload("C2", num_samples=K)
    p = w.params
    state, xr, yr, yaw = _inputs(p, w.path, lateral=0.0)
    whole = MPPIController(p)
    for it in range(iters):
        whole.iterate_enqueue(state, p.dt, xr, yr, yaw[0], 5, it)
    np.testing.assert_allclose(u0, whole.get_nominal(), rtol=1e-7, atol=1e-10)
    print("fine-grained boxes:", [int(np.load(tmp_path / ("fine%d.npy" % r))[0]) for r in range(nproc * per)])


# ---- the RCCL branch of the sharded driver, for real, on a one-rank nccl process group -----------------------------------
def _nccl1_worker(rank, port, K, iters, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    w = configs.workload("C2", num_samples=K)
    p = w.params
    state, xr, yr, yaw = _inputs(p, w.path, lateral=0.0)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        a = MPPIController(p)
        a.set_stream(stream.cuda_stream)
        b = MPPIController(p)
        drv = sharded.ShardedMPPI(sharded.DevicePartials(b))   # rollout -> k_finalize -> ncclAllReduce -> deferred division
        for it in range(iters):
            a.iterate_enqueue(state, p.dt, xr, yr, yaw[0], 5, it)
            drv.iterate(state, p.dt, xr, yr, yaw[0], 5, it)
        ua, ub = a.get_nominal(), b.get_nominal()
    np.save(os.path.join(out_dir, "ua.npy"), ua)
    np.save(os.path.join(out_dir, "ub.npy"), ub)
    dist.destroy_process_group()


def test_sharded_driver_over_a_one_rank_nccl_group(tmp_path):
    """ShardedMPPI(DevicePartials) with backend `nccl` (= RCCL), world size 1: 40 iterations, every one with a real
    ncclAllReduce(sum, double) of the partial vector between the update kernel and the deferred division, on the stream the
    kernels run on, no host synchronisation.  One rank's sum is the vector itself, so the loop must equal
    ccv_mppi_iterate_enqueue bit for bit."""
    import torch.multiprocessing as mp
    mp.spawn(_nccl1_worker, args=(_free_port(), 4096, 40, str(tmp_path)), nprocs=1, join=True)
    ua, ub = np.load(tmp_path / "ua.npy"), np.load(tmp_path / "ub.npy")
    assert np.all(np.isfinite(ua))
    np.testing.assert_array_equal(ua, ub)
