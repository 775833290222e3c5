"""K sharded over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).

Rank g owns the samples [g*K_local, (g+1)*K_local) and draws their noise with GLOBAL sample ids, so the union of the
shards is exactly the single-device sample set.  Every rank receives the same (pose, dt, window) and produces the
unnormalised partial vector [sum_i w_i, sum_i w_i*u_i[t][d]] of its shard; ONE all-reduce(sum) of 1 + (H-1)*u_dim
doubles (<= 3.2 KB) per iteration follows, then every rank divides -- which reproduces the reference's normalised
weighted mean (src/diff_drive_mppi.cpp:216-237) without a min-cost shift.  The message is latency-bound; nothing else
crosses the links.
"""
import numpy as np


def shard_bounds(num_samples_global, world_size, rank):
    """Contiguous, equal shards (the driver sizes K as a multiple of the world size)."""
    if num_samples_global % world_size:
        raise ValueError("num_samples (%d) must be a multiple of the world size (%d)" % (num_samples_global, world_size))
    k_local = num_samples_global // world_size
    return rank * k_local, k_local


class ShardedMPPI:
    """Drives one shard.  `backend` computes the local partials and applies the reduced ones:
         backend.local_partials(x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration) -> torch tensor [1 + (H-1)*u_dim] (float64)
         backend.apply(reduced tensor)                                         -> None (u* <- V / S)
       The production backend is DevicePartials below (GPU, no host sync); the CPU tests plug the oracle in."""

    def __init__(self, backend, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.backend = backend
        self.group = group

    def iterate(self, x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration):
        if hasattr(self.backend, "iterate"):   # the backend exchanges the partials itself (ExchangeBackend)
            return self.backend.iterate(x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration)
        part = self.backend.local_partials(x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration)
        if self.dist.is_initialized():   # (also with one rank: the same call sequence whatever the world size)
            self.dist.all_reduce(part, op=self.dist.ReduceOp.SUM, group=self.group)
        self.backend.apply(part)
        return part

    def iterate_resident(self, dt, seed, iteration, advance=True):
        """The device-resident closed loop, K sharded: every rank holds the same path and pose (backend.resident_setup),
        advances the pose with the same u* and builds the same window; only the partial vector crosses the links.
        backend.local_partials_resident(dt, seed, iteration, advance) -> tensor as local_partials."""
        if hasattr(self.backend, "iterate_resident"):   # ExchangeBackend
            return self.backend.iterate_resident(dt, seed, iteration, advance)
        part = self.backend.local_partials_resident(dt, seed, iteration, advance)
        if self.dist.is_initialized():
            self.dist.all_reduce(part, op=self.dist.ReduceOp.SUM, group=self.group)
        self.backend.apply(part)
        return part


class DevicePartials:
    """GPU backend: the partials live in a torch tensor on the controller's device; everything is enqueued on the
    current torch stream (no host synchronisation between the rollout, the all-reduce and the division)."""

    def __init__(self, controller):
        import torch
        self.torch = torch
        self.ctl = controller
        self.buf = torch.zeros(controller.partials_size(), dtype=torch.float64, device="cuda")
        controller.set_stream(torch.cuda.current_stream().cuda_stream)

    def local_partials(self, x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration):
        self.ctl.iterate_partials_enqueue(x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration, self.buf.data_ptr())
        return self.buf

    def resident_setup(self, path_x, path_y, state, resolution=None):
        self.ctl.resident_set_path(path_x, path_y, resolution)
        self.ctl.resident_set_pose(state)

    def local_partials_resident(self, dt, seed, iteration, advance):
        self.ctl.resident_step_partials_enqueue(dt, seed, iteration, advance, self.buf.data_ptr())
        return self.buf

    def apply(self, reduced):
        self.ctl.apply_partials_enqueue(reduced.data_ptr())


class ExchangeBackend:
    """GPU backend for the devices of ONE node that needs no collective call per iteration: every device writes its
    partial vector straight into a box in each peer's HBM (mapped once through hipIpc, xGMI peer access), the update kernel
    waits for the peers' vectors and adds them in rank order (ccv_mppi_exchange_* in include/ccv_mppi.h).  The process
    group is used once, to hand the IPC handles round.  `ok` is False when the boxes could not be created or mapped on
    some rank -- every rank then sees False and the caller falls back to DevicePartials + all-reduce."""

    def __init__(self, controller, group=None):
        import torch
        import torch.distributed as dist
        self.ctl = controller
        controller.set_stream(torch.cuda.current_stream().cuda_stream)
        multi = dist.is_initialized() and dist.get_world_size(group) > 1
        world = dist.get_world_size(group) if multi else 1
        rank = dist.get_rank(group) if multi else 0
        try:
            mine = controller.exchange_create(world, rank)
        except Exception as e:   # noqa: BLE001 -- any failure means "not available here"
            mine, self.error = None, str(e)
        handles = [mine]
        if multi:
            handles = [None] * world
            dist.all_gather_object(handles, mine, group=group)
        good = all(h is not None for h in handles)
        if good:
            try:
                controller.exchange_connect(handles)
            except Exception as e:   # noqa: BLE001
                good, self.error = False, str(e)
        if multi:
            flags = [None] * world
            dist.all_gather_object(flags, bool(good), group=group)
            good = all(flags)
        self.ok = bool(good)
        self.info = controller.exchange_info() if mine is not None else {}

    def iterate(self, x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration):
        self.ctl.iterate_exchange_enqueue(x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration)

    def resident_setup(self, path_x, path_y, state, resolution=None):
        self.ctl.resident_set_path(path_x, path_y, resolution)
        self.ctl.resident_set_pose(state)

    def iterate_resident(self, dt, seed, iteration, advance=True):
        self.ctl.resident_step_exchange_enqueue(dt, seed, iteration, advance)


def combine_partials(parts):
    """Reference combination on the host (tests): sum of the per-shard vectors, then u* = V / S."""
    tot = np.sum(np.asarray(parts, dtype=np.float64), axis=0)
    return tot[1:] / tot[0], tot
