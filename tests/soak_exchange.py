#!/usr/bin/env python3
"""Soak of the direct exchange (not collected by pytest; run on a GPU box):  python tests/soak_exchange.py [world] [iters]
`world` processes share device 0, each owns K/world samples, `iters` iterations back to back with a read-back every 997
iterations; all ranks must hold identical, finite controls at every read-back."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port, iters, out_dir):
    import torch
    import torch.distributed as dist
    import helpers
    from ccv_mppi_path_tracker_amd import configs, sharded
    from ccv_mppi_path_tracker_amd.controller import MPPIController
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    K = 8192
    w = configs.workload("C2", num_samples=K)
    p = w.params
    off, k_local = sharded.shard_bounds(K, world, rank)
    ctl = MPPIController(p, num_samples=k_local, sample_offset=off)
    stream = torch.cuda.Stream()
    snaps = []
    with torch.cuda.stream(stream):
        xb = sharded.ExchangeBackend(ctl)
        assert xb.ok, getattr(xb, "error", "")
        drv = sharded.ShardedMPPI(xb)
        path = helpers.oracle_path(w.path)
        state = np.zeros(p.nstate)
        state[:2] = path[0][0], path[1][0] + 0.05
        xr, yr, yaw = helpers.oracle_window(p, path, state)
        for it in range(iters):
            drv.iterate(state, p.dt, xr, yr, yaw[0], 5, it)
            if it % 997 == 996 or it == iters - 1:
                snaps.append(ctl.get_nominal().copy())
    np.save(os.path.join(out_dir, "snap%d.npy" % rank), np.stack(snaps))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(world, port, iters, d), nprocs=world, join=True)
        snaps = [np.load(os.path.join(d, "snap%d.npy" % r)) for r in range(world)]
    same = all(np.array_equal(snaps[0], s_) for s_ in snaps[1:])
    print("world=%d iters=%d read-backs=%d: identical on all ranks %s, finite %s" % (world, iters, len(snaps[0]), same, bool(np.all(np.isfinite(snaps[0])))))
    sys.exit(0 if same and np.all(np.isfinite(snaps[0])) else 1)
