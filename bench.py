#!/usr/bin/env python3
"""bench.py -- trajectory rollouts/s of the MPPI hot path on MI355X (BASELINE.json metric).

A "step" is one whole MPPI iteration (sample + rollout + cost + weights + update) over one batch of K samples
of synthetic input; the workload at N=1 is BASELINE.json configs[1] (C2: diff_drive, K=65 536, T=50, sinusoid
reference path, launch parameters).  With N>1 ranks (one process per GPU, torch.distributed/RCCL) K is sharded:
every rank rolls out K=65 536 samples with global sample ids (weak scaling; N=8 is BASELINE configs[4], K=524 288)
and the per-rank partials [sum w, sum w*u] are all-reduced once per iteration.

Inputs (pose + reference window) are precomputed on the host and passed as kernel arguments; the warm start u*
stays resident in HBM and evolves from step to step.  Nothing under oracle/ is touched except for the
`cpu_baseline` leg (rank 0, N=1), which times the CPU restatement of the reference loop on the host.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python bench.py --gpus N ...          (no launcher: the parent starts the N ranks itself, before anything touches a GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 GB/s achievable


def algorithmic_bytes(T, udim):
    """B = 8*[2*(T-1)*u_dim + 2*T + 2] per rollout: controls written once and read once for the weighted
    update, x,y written once, weight written and read once.  The rollout kernel does all of it (the update's partial sums
    are fused into its epilogue), so B is also the dominant kernel's algorithmic traffic."""
    whole = 8 * (2 * (T - 1) * udim + 2 * T + 2)
    return whole, whole


def rollout_kernel_name(model, k_local, device):
    """The kernel ccv_mppi_create selects (csrc/ccv_mppi_capi.hip): more blocks of 64 samples than four (full body) / five per CU ->
    one wave per block (k_rollout_solo); else the four-wave kernel -- full body: up to one block per CU, the two-wave kernel
    (k_rollout_pc) beyond."""
    forced = os.environ.get("CCV_MPPI_KERNEL")
    if forced:
        return {"v1": "k_rollout_cost", "pc": "k_rollout_pc", "r3": "k_rollout_pc" if model == "full_body" else "k_rollout_r3",
                "r4": "k_rollout_r4", "solo": "k_rollout_solo"}.get(forced, forced)
    import torch
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    blocks = (k_local + 63) // 64
    if blocks > (4 if model == "full_body" else 5) * cus:
        return "k_rollout_solo"
    if model == "full_body" and blocks > cus:
        return "k_rollout_pc"
    return "k_rollout_r4"


def device_copy_gbs(torch, nbytes=1 << 30, reps=10):
    """Achievable HBM ceiling next to the nominal 8 TB/s: a device-to-device copy of 1 GiB, bytes read +
    bytes written per second, best of `reps` (torch's copy kernel; measured after the timed region)."""
    a = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda")
    b = torch.ones(nbytes // 8, dtype=torch.float64, device="cuda")
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        a.copy_(b)
        e1.record()
        e1.synchronize()
        best = max(best, 2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return best


def script_inputs(amd, w, n):
    """n (pose, window) pairs along the reference path: the robot advances ~v_ref*dt per step with a small lateral
    and heading offset, as a tracking controller would see them."""
    p = w.params
    px, py = amd.make_path(w.path)
    rng = np.random.default_rng(0)
    span = max(1, len(px) - 2 * p.horizon // 3)
    if os.environ.get("CCV_BENCH_SCRIPT_SPAN"):   # diagnostic: only poses whose window does not run past the end of the path
        span = max(1, min(span, int(os.environ["CCV_BENCH_SCRIPT_SPAN"])))
    out = []
    for i in range(n):
        j = int(i * p.v_ref * p.dt / p.resolution) % span
        yaw = np.arctan2(py[j + 1] - py[j], px[j + 1] - px[j])
        s = np.zeros(p.nstate)
        s[0], s[1], s[2] = px[j] + rng.normal(0, 0.03), py[j] + rng.normal(0, 0.03), yaw + rng.normal(0, 0.05)
        _, xr, yr, yawr = amd.calc_ref_path(px, py, s[0], s[1], p.v_ref, p.dt, p.resolution, p.horizon)
        out.append((s, xr, yr, float(yawr[0])))
    return out


def host_cores():
    """CPUs this process may run on (the GPU boxes give a one-GPU job a share of a 256-CPU host)."""
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(w, budget_s=12.0, by_value=False):
    """The oracle (CPU restatement of the reference's single-threaded loop, mt19937 mode) on a bounded sample of the
    same workload: whole iterations of the full K until ~budget_s of CPU time is spent.  by_value: calc_Cost /
    calc_MinDistance take their arguments by value as the reference's signatures do (dd.h:130,140, dd:183,194; SURVEY.md
    Q16) -- the reference-shaped baseline of SURVEY.md 8(d)(i); otherwise the same loop without those copies (a fast port)."""
    import ccv_mppi_path_tracker_amd as amd
    from oracle import oracle_lib as O
    p = w.params
    o = O.Oracle(p.model, p.num_samples, p.horizon, p.control_noise, p.lam, p.v_ref, p.u_min, p.u_max,
                 path_weight=p.path_weight, v_weight=p.v_weight, zmp_weight=p.zmp_weight, roll_v_weight=p.roll_v_weight,
                 back_weight=p.back_weight, yaw_weight=p.yaw_weight, roll_off=p.roll_off, steer_off=p.steer_off)
    o.set_by_value(by_value)
    inputs = script_inputs(amd, w, 16)
    n, t0 = 0, time.perf_counter()
    while True:
        s, xr, yr, yaw0 = inputs[n % len(inputs)]
        o.iterate(s, p.dt, xr, yr, yaw0, seed=42 + n, rng="mt19937")
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    shape = ("AoS of std::vector, calc_Cost / calc_MinDistance arguments BY VALUE as dd.h:130,140" if by_value
             else "the same loop without the reference's by-value copies: a fast port")
    return {"value": p.num_samples * n / el, "unit": "rollouts/s", "cores": 1, "kind": "port",
            "sample": "%d whole iterations of %s (oracle/mppi_oracle.cpp, serial mt19937 like the reference; %s; "
                      "%.1f s, %.0f ms/iteration)" % (n, w.description, shape, el, 1e3 * el / n),
            "host_cpus": os.cpu_count(), "cores_available": host_cores()}


def cpu_baseline_threads(w, threads, budget_s=6.0):
    """The strong CPU baseline of SURVEY.md 8(d): the same oracle in its counter-based (Philox) mode, K sharded over
    `threads` host threads exactly as it is over GPUs (global sample ids, [sum w, sum w*u] combined, then divided); the
    C++ calls release the GIL.  Whole iterations until ~budget_s."""
    import concurrent.futures as cf
    import ccv_mppi_path_tracker_amd as amd
    from oracle import oracle_lib as O
    p = w.params
    k_t = p.num_samples // threads
    shards = [O.Oracle(p.model, k_t, p.horizon, p.control_noise, p.lam, p.v_ref, p.u_min, p.u_max,
                       path_weight=p.path_weight, v_weight=p.v_weight, zmp_weight=p.zmp_weight,
                       roll_v_weight=p.roll_v_weight, back_weight=p.back_weight, yaw_weight=p.yaw_weight,
                       roll_off=p.roll_off, steer_off=p.steer_off) for _ in range(threads)]
    inputs = script_inputs(amd, w, 16)
    n, t0 = 0, time.perf_counter()
    with cf.ThreadPoolExecutor(threads) as pool:
        while True:
            s, xr, yr, yaw0 = inputs[n % len(inputs)]
            us = list(pool.map(lambda a: a[1].iterate(s, p.dt, xr, yr, yaw0, seed=42, rng="philox", iteration=n, k_offset=a[0] * k_t),
                               enumerate(shards)))
            sw = np.array([o.sum_w() for o in shards])
            u = np.tensordot(sw, np.stack(us), axes=1) / sw.sum()
            for o in shards:
                o.set_nominal(u)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n >= 256:
                break
    return {"value": k_t * threads * n / el, "unit": "rollouts/s", "cores": threads, "kind": "port",
            "sample": "%d whole iterations of %s, K sharded over %d threads (oracle/mppi_oracle.cpp, Philox mode, "
                      "%.1f s, %.1f ms/iteration)" % (n, w.description, threads, el, 1e3 * el / n),
            "host_cpus": os.cpu_count(), "cores_available": host_cores()}


def closed_loop_leg(amd, torch, ctl, w, seed, warm=64, ticks=512):
    """SURVEY.md 8(d) asks for the closed loop: the device-resident tick (plant + get_CurrentIndex + calc_RefPath +
    iteration, two launches per tick -- the rollout kernel, and the update fused with the next tick's prologue -- no host data:
    ccv_mppi_resident_step_enqueue), `ticks` of them back to back after `warm`,
    pose fed back every tick, on the workload's own path with the course extended so that it does not end."""
    p = w.params
    need = (warm + ticks + 8) * max(abs(p.u_max[0]), abs(p.u_min[0])) * p.dt + 2.0 * p.horizon * p.v_ref * p.dt
    px, py = amd.make_path(w.path, p.resolution, length=max(need, 10.0))
    s0 = np.zeros(p.nstate)
    s0[0], s0[1] = px[0], py[0]
    ctl.set_nominal(np.zeros((p.horizon - 1, p.udim)))
    ctl.resident_set_path(px, py)
    ctl.resident_set_pose(s0)
    for i in range(warm):
        ctl.resident_step_enqueue(p.dt, seed, i, advance=i > 0)
    ctl.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(ticks):
        ctl.resident_step_enqueue(p.dt, seed, warm + i, advance=True)
    ctl.synchronize()
    el = time.perf_counter() - t0
    tr = ctl.resident_read_trace(max_rows=ticks)
    d = np.hypot(px[None, :] - tr[:, 0:1], py[None, :] - tr[:, 1:2]).min(axis=1)
    return {"us_per_tick": 1e6 * el / ticks, "rollouts_per_s": p.num_samples * ticks / el, "ticks": ticks, "warmup_ticks": warm,
            "what": "device-resident closed loop: pose advanced by u*[0], window rebuilt on the device, %d path poses" % len(px),
            "path_error_rms_m": float(np.sqrt(np.mean(d * d))), "path_error_max_m": float(d.max()),
            "distance_travelled_m": float(np.hypot(np.diff(tr[:, 0]), np.diff(tr[:, 1])).sum())}


def live_pmc_traffic(workload_name, timeout_s=60):
    """HBM bytes per rollout-kernel launch, MEASURED for this run: two short child runs of this script under
    `rocprofv3 --pmc` -- WRITE_SIZE and FETCH_SIZE, each in a pass of its own with nothing else (MI355X_MICROARCH.md, HBM:
    FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2) -- averaged over the launches of the rollout kernel and corrected as that
    guide prescribes for gfx950 (WRITE_SIZE exact, FETCH_SIZE counts the 128-byte requests of a coalesced stream as 64 bytes:
    doubled; calibrated for this code's access pattern with tools/microbench/hbm_calib.hip).  Called before this process
    touches the GPU (the children have it to themselves, and nothing is started from a process that holds a GPU context).
    Returns (bytes, detail) or (None, reason)."""
    import csv
    import glob as _glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    got = {}
    for counter in ("WRITE_SIZE", "FETCH_SIZE"):
        d = tempfile.mkdtemp(prefix="ccv_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--workload", workload_name, "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-kernel-events",
               "--no-closed-loop-leg", "--no-other-workloads", "--no-defaults-leg", "--no-live-traffic"]
        try:
            res = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
        except Exception as e:   # noqa: BLE001 -- a profiler that cannot run is not a reason to fail the bench
            shutil.rmtree(d, ignore_errors=True)
            return None, "%s pass: %s" % (counter, e)
        vals = []
        for f in _glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "rollout" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    vals.append(float(r["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
        if res.returncode != 0 or not vals:
            return None, "%s pass: rc %d, %d kernel rows" % (counter, res.returncode, len(vals))
        got[counter] = (sum(vals) / len(vals), len(vals))
    total = (got["WRITE_SIZE"][0] + 2.0 * got["FETCH_SIZE"][0]) * 1024.0
    return total, {"WRITE_SIZE_KB": round(got["WRITE_SIZE"][0]), "FETCH_SIZE_KB_raw": round(got["FETCH_SIZE"][0]),
                   "launches_averaged": [got["WRITE_SIZE"][1], got["FETCH_SIZE"][1]],
                   "correction": "gfx950: WRITE_SIZE exact, FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM)"}


def latest_pmc_traffic(workload_name):
    """HBM bytes per rollout-kernel launch from the committed rocprofv3 --pmc summary (profiles/*pmc*.json)."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json"))):
        try:
            d = json.load(open(f))
            if d.get("workload") == workload_name and d.get("hbm_bytes_per_launch"):
                best = float(d["hbm_bytes_per_launch"])
        except Exception:
            pass
    return best


def measure_workload(amd, torch, configs, name, steps, warmup, stream, device, prime_s=0.3):
    """One more single-GPU BASELINE workload (C3, C4) measured the way the headline is: untimed priming, `warmup` steps,
    `steps` timed steps between two device synchronisations, then a 256-launch pass with hipEvents on the rollout kernel's
    dispatch.  Returns the entry of `other_workloads`."""
    w = configs.workload(name)
    p = w.params
    ctl = amd.MPPIController(p, device=device)
    ctl.set_stream(stream.cuda_stream)
    inputs = script_inputs(amd, w, 64)

    def step(i):
        s, xr, yr, yaw0 = inputs[i % len(inputs)]
        ctl.iterate_enqueue(s, p.dt, xr, yr, yaw0, 42, i)

    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < prime_s or n < 256:
        for _ in range(64):
            step(n)
            n += 1
    torch.cuda.synchronize()
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ctl.timing_enable(True, every=1)
    ctl.timing_read(reset=True)
    for i in range(256):
        step(warmup + steps + i)
    torch.cuda.synchronize()
    roll_us, iter_us, n_ev = ctl.timing_read(reset=True)
    ctl.timing_enable(False)
    finite = bool(np.all(np.isfinite(ctl.get_nominal())))
    ctl.close()
    B, _ = algorithmic_bytes(p.horizon, p.udim)
    k_us = roll_us / max(n_ev, 1)
    traffic = latest_pmc_traffic(name)
    return {"workload": "%s: %s, u_dim=%d, launch parameters" % (w.name, w.description, p.udim),
            "value": p.num_samples * steps / el, "unit": "rollouts/s", "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * el / steps, "primed_iterations": n, "finite": finite,
            "kernel": rollout_kernel_name(p.model, p.num_samples, device), "kernel_avg_us": k_us,
            "kernel_launches_averaged": int(n_ev), "iteration_avg_us": iter_us / max(n_ev, 1),
            "algorithmic_bytes_per_launch": B * p.num_samples,
            "frac": B * p.num_samples / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS if k_us > 0 else None,
            "traffic": traffic,
            "measured_traffic_frac": (traffic / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if (traffic and k_us > 0) else None}


def defaults_leg(amd, device, with_cpu=True):
    """The operating points the reference itself runs at: dd K = 1 000 (dd:18-19), sd K = 1 000
    (launch/steering_diff_drive_mppi.launch:11), fb K = 10 000 (fb:9-10), H = 15, inside a 100 ms tick (dd:334) -- what a node
    that drops the library in gets per tick.  Timed around ONE C call per tick, ccv_mppi_node_run_once() of the ROS-free
    mirror (csrc/host/mppi_node.cpp: run()'s body, dd:346-361), closed loop on the host plant:
      blocking_iterate    calc_RefPath() on the host + blocking ccv_mppi_iterate              (use_fused_)
      stage_wise_4_calls  the four stage-wise calls INTEGRATION.md section 2 pastes into the method bodies
      device_prologue     window built on the device + iteration + u* back
    and, through ccv_mppi_iterate_enqueue, the back-to-back rate without a host wait.  The CPU figure is the oracle
    (mt19937 mode) at the same size on one host core."""
    from ccv_mppi_path_tracker_amd import configs
    from ccv_mppi_path_tracker_amd.node import ControllerNode
    out = {}
    cases = (("dd_K1000_H15", "diff_drive", {}, "sinusoid", configs.diff_drive_defaults(1000, 15)),
             ("sd_K1000_H15", "steering_diff_drive", {"num_samples": 1000}, "sinusoid", configs.steering_defaults(1000, 15)),
             ("fb_K10000_H15", "full_body", {}, "dkan", configs.full_body_defaults(10000, 15)))
    for key, model, params, path, p in cases:
        px, py = amd.make_path(path)
        res = {"what": "%s defaults, K=%d H=%d, %s path" % (model, p.num_samples, p.horizon, path)}
        for mode, kw in (("blocking_iterate", dict(fused=True)), ("stage_wise_4_calls", dict(fused=False)),
                         ("device_prologue", dict(fused=True, device_prologue=True))):
            node = ControllerNode(model, params, device=device, **kw)
            node.set_path(px, py)
            s = np.zeros(5)
            s[0], s[1] = px[0], py[0]
            lat = []
            for i in range(260):
                node.set_state(s)
                t0 = time.perf_counter()
                cmd = node.run_once(p.dt)
                lat.append(time.perf_counter() - t0)
                u0 = node.optimal_solution()[0]
                s[:p.nstate] = amd.plant_step(model, s[:p.nstate], u0, p.dt)
                if np.hypot(px[-1] - s[0], py[-1] - s[1]) < 1.0 or cmd is None:   # end of the course: start over
                    s[:] = 0.0
                    s[0], s[1] = px[0], py[0]
            node.close()
            res[mode + "_us_median"] = 1e6 * float(np.median(lat[60:]))
        # back to back, no host wait: n enqueues, one synchronisation
        ctl = amd.MPPIController(p, device=device)
        inputs = script_inputs(amd, configs.Workload(key, p, path), 16)
        for rep in range(2):
            n = 512
            t0 = time.perf_counter()
            for i in range(n):
                s0, xr, yr, yaw0 = inputs[i % len(inputs)]
                ctl.iterate_enqueue(s0, p.dt, xr, yr, yaw0, 42, i)
            ctl.synchronize()
            res["iterate_enqueue_us_per_iteration"] = 1e6 * (time.perf_counter() - t0) / n
        ctl.timing_enable(True, every=1)
        ctl.timing_read(reset=True)
        for i in range(64):
            s0, xr, yr, yaw0 = inputs[i % len(inputs)]
            ctl.iterate_enqueue(s0, p.dt, xr, yr, yaw0, 42, i)
        ctl.synchronize()
        r_us, i_us, n_ev = ctl.timing_read(reset=True)
        ctl.close()
        res["rollout_kernel_us"] = r_us / max(n_ev, 1)
        res["device_iteration_us"] = i_us / max(n_ev, 1)
        res["kernel"] = rollout_kernel_name(p.model, p.num_samples, device)
        if not with_cpu:
            out[key] = res
            continue
        # the oracle at the same size (one core): a cpu_baseline leg, the only kind of use bench.py makes of oracle/
        from oracle import oracle_lib as O
        o = O.Oracle(p.model, p.num_samples, p.horizon, p.control_noise, p.lam, p.v_ref, p.u_min, p.u_max,
                     path_weight=p.path_weight, v_weight=p.v_weight, zmp_weight=p.zmp_weight, roll_v_weight=p.roll_v_weight,
                     back_weight=p.back_weight, yaw_weight=p.yaw_weight, roll_off=p.roll_off, steer_off=p.steer_off)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 1.0 or n < 3:
            s0, xr, yr, yaw0 = inputs[n % len(inputs)]
            o.iterate(s0, p.dt, xr, yr, yaw0, seed=42 + n, rng="mt19937")
            n += 1
        res["cpu_oracle_ms_per_iteration_1_core"] = 1e3 * (time.perf_counter() - t0) / n
        out[key] = res
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, 127.0.0.1 rendezvous), let rank 0 print the JSON line on the inherited
    stdout, and return the first non-zero exit code (the others are then stopped by their exact PIDs).  Called before torch
    or the HIP library is imported: the parent never touches a GPU and nothing is ever exec'ed over a process that has."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                              env=dict(base, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
    rc, alive = 0, set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print("bench.py: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
                t_stop = time.time()
                while any(procs[q].poll() is None for q in alive) and time.time() - t_stop < 15.0:
                    time.sleep(0.05)
                for q in alive:
                    if procs[q].poll() is None:
                        procs[q].kill()
        time.sleep(0.02)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C2", help="C2 (headline) | C3 | C4 | dd_default | sd_default | fb_default (the reference's own "
                                                     "operating points, K = 1 000 / 1 000 / 10 000 at H = 15: profiling, not a bench line)")
    ap.add_argument("--samples-per-gpu", type=int, default=None)
    ap.add_argument("--path", default=None, help="reference path instead of the workload's: straight | sinusoid | dkan")
    ap.add_argument("--dt", type=float, default=None,
                    help="loop period instead of the workload's 0.1 s (the node measures it, dd:346-348): beyond |w|max*dt = pi/4 "
                         "(0.3927 s at the C2 limits) the host launches the kernel's full-range sin/cos instantiation (not the headline)")
    ap.add_argument("--closed-loop", action="store_true",
                    help="not the headline: the device-resident closed loop (pose advanced by u*[0] and the window rebuilt "
                         "on the device every step; the workload's path generator with the course extended so that it never ends)")
    ap.add_argument("--exchange", default=os.environ.get("CCV_MPPI_EXCHANGE", "auto"), choices=["auto", "p2p", "rccl"],
                    help="N > 1: how the partial vectors meet -- p2p: written straight into the peers' HBM by the update "
                         "kernel (one node); rccl: one all-reduce per iteration; auto: p2p when it can be set up and "
                         "reproduces the all-reduce result, else rccl")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-closed-loop-leg", action="store_true", help="skip the device-resident closed-loop figure of the default line")
    ap.add_argument("--no-state-store", action="store_true", help="skip the KxH x,y buffer (not the headline)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not record per-kernel hipEvents in the timed region")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed PMC profile instead of two rocprofv3 --pmc child runs of this launch")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the C3 / C4 legs of the default line")
    ap.add_argument("--no-defaults-leg", action="store_true", help="skip the reference-default operating points (K = 1 000 / 10 000, H = 15)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    # roofline.traffic, measured for THIS run (the headline launch only; before anything here touches the GPU)
    live_traffic, live_detail = None, "not the headline launch"
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_live_traffic and not args.closed_loop
            and args.workload in ("C2", "C3", "C4") and args.samples_per_gpu is None and args.dt is None and args.path is None
            and not args.no_state_store and not os.environ.get("CCV_MPPI_KERNEL")):
        live_traffic, live_detail = live_pmc_traffic(args.workload)

    import torch
    import torch.distributed as dist
    import ccv_mppi_path_tracker_amd as amd
    from ccv_mppi_path_tracker_amd import configs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # (rehearsal of the N > 1 path on a one-GPU box: CCV_BENCH_DEVICE=0 puts every rank on device 0 and
    #  CCV_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device; never the reported configuration)
    if "CCV_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["CCV_BENCH_DEVICE"])
    backend = os.environ.get("CCV_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if args.workload.endswith("_default"):
        w = {"dd_default": configs.Workload("dd_default", configs.diff_drive_defaults(1000, 15), "sinusoid", "diff_drive K=1000 T=15 sinusoid (code defaults)"),
             "sd_default": configs.Workload("sd_default", configs.steering_defaults(1000, 15), "sinusoid", "steering_diff_drive K=1000 T=15 sinusoid (code defaults, launch K)"),
             "fb_default": configs.Workload("fb_default", configs.full_body_defaults(10000, 15), "dkan", "full_body K=10000 T=15 dkan (code defaults)")}[args.workload]
    else:
        w = configs.workload(args.workload)
    if args.path:
        import dataclasses
        w = dataclasses.replace(w, path=args.path, description=w.description.replace(w.path, args.path))
    if args.dt is not None:
        import dataclasses
        w = dataclasses.replace(w, params=w.params.with_(dt=args.dt), description=w.description + " dt=%g" % args.dt)
    p = w.params
    k_local = args.samples_per_gpu or p.num_samples
    k_total = k_local * world
    ctl = amd.MPPIController(p, device=local_rank, num_samples=k_local, sample_offset=rank * k_local,
                             no_state_store=args.no_state_store)
    from ccv_mppi_path_tracker_amd import sharded
    # a dedicated (non-default) stream: the legacy default stream adds implicit synchronisation to every launch
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctl.set_stream(stream.cuda_stream)
    inputs = script_inputs(amd, w, 64)
    seed = 42
    # N > 1: per-GPU partials [sum w, sum w*u] -> one small RCCL all-reduce over xGMI -> every rank divides (no host sync)
    driver, exchange_used, exchange_info = None, None, {}

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        rccl = sharded.ShardedMPPI(sharded.DevicePartials(ctl))
        driver, exchange_used = rccl, ("rccl all-reduce" if backend == "nccl" else "%s all-reduce (rehearsal, not RCCL)" % backend)
        zeros = np.zeros((p.horizon - 1, p.udim))

        def run(drv, n):
            for i in range(n):
                s0, xr0, yr0, yaw00 = inputs[i % len(inputs)]
                drv.iterate(s0, p.dt, xr0, yr0, yaw00, seed, i)

        def time_256(drv):
            """us per step of 256 steps after 1024 untimed ones (the first several hundred launches of a process are slow
            on the host); the slowest rank decides"""
            run(drv, 1024)
            fence()
            t_sel = time.perf_counter()
            run(drv, 256)
            fence()
            t = torch.tensor([time.perf_counter() - t_sel], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item()) / 256 * 1e6

        direct, good, xb = None, False, None
        if args.exchange != "rccl":
            xb = sharded.ExchangeBackend(ctl)
            good = xb.ok
            direct = sharded.ShardedMPPI(xb) if good else None
            if good:
                # both ways on the same inputs from the same warm start: the direct exchange is considered only if it
                # reproduces the all-reduce result on every rank (rank-order vs ring-order sums: rounding only)
                got = []
                for drv in (direct, rccl):
                    ctl.set_nominal(zeros)
                    run(drv, 3)
                    got.append(ctl.get_nominal())
                good = bool(np.all(np.isfinite(got[0])) and np.allclose(got[0], got[1], rtol=1e-9, atol=1e-12))
            flags = [None] * world
            dist.all_gather_object(flags, good)
            good = all(flags)
            exchange_info.update(direct_available=bool(good), direct_fine_grained=bool(getattr(xb, "info", {}).get("fine_grained", False)))
            if not good and args.exchange == "p2p":
                print("bench.py: --exchange p2p could not be set up or verified: %s" % getattr(xb, "error", flags), file=sys.stderr)
                sys.exit(5)
        # the all-reduce path is always timed (north_star's figure), the direct exchange when it is available; the faster
        # one runs the timed region unless --exchange forces one
        ctl.set_nominal(zeros)
        # (key name: `rccl_us_per_step` only when the process group really is RCCL; the one-device rehearsal uses gloo)
        ar_key = "rccl_us_per_step" if backend == "nccl" else "%s_allreduce_us_per_step" % backend
        # what the process group itself reports (not the launcher's environment): a SCALE line proves with these two keys
        # that RCCL saw N ranks
        exchange_info.update(world_seen=int(dist.get_world_size()), backend_seen=str(dist.get_backend()))
        exchange_info.update(backend=backend, **{ar_key: time_256(rccl)})
        if good:
            ctl.set_nominal(zeros)
            exchange_info.update(direct_us_per_step=time_256(direct))
            if exchange_info["direct_us_per_step"] <= exchange_info[ar_key] or args.exchange == "p2p":
                driver = direct
                exchange_used = "direct stores into the peers' HBM (hipIpc over xGMI), rank-order sum"
            exchange_used += " [256 steps: direct %.1f us/step, %s all-reduce %.1f us/step]" % (
                exchange_info["direct_us_per_step"], "rccl" if backend == "nccl" else backend, exchange_info[ar_key])
        ctl.set_nominal(zeros)
    elif args.exchange == "p2p":   # N = 1: the exchange kernel talking to itself (its overhead over k_finalize)
        xb = sharded.ExchangeBackend(ctl)
        if not xb.ok:
            print("bench.py: --exchange p2p could not be set up: %s" % getattr(xb, "error", ""), file=sys.stderr)
            sys.exit(5)
        driver, exchange_used = sharded.ShardedMPPI(xb), "direct exchange, one rank"
        exchange_info.update(direct_available=True, direct_fine_grained=bool(xb.info.get("fine_grained", False)))

    if args.closed_loop:
        if world > 1:
            print("bench.py: --closed-loop is a single-GPU measurement", file=sys.stderr)
            sys.exit(2)
        # the course must outlast warm-up + timed steps at the largest speed the controller may command
        need = (args.warmup + args.steps + 8) * max(abs(p.u_max[0]), abs(p.u_min[0])) * p.dt + 2.0 * p.horizon * p.v_ref * p.dt
        cl_px, cl_py = amd.make_path(w.path, p.resolution, length=max(need, 10.0))
        cl_start = np.zeros(p.nstate)
        cl_start[0], cl_start[1] = cl_px[0], cl_py[0]
        ctl.resident_set_path(cl_px, cl_py)
        ctl.resident_set_pose(cl_start)

    def step(i):
        if args.closed_loop:
            ctl.resident_step_enqueue(p.dt, seed, i, advance=True)
            return
        s, xr, yr, yaw0 = inputs[i % len(inputs)]
        if driver is None:
            ctl.iterate_enqueue(s, p.dt, xr, yr, yaw0, seed, i)
        else:
            driver.iterate(s, p.dt, xr, yr, yaw0, seed, i)

    # Set-up, not part of the warm-up count: for the first several hundred launches of a process the host side of a
    # launch is several times slower on some boxes (runtime pools growing, host and device clocks ramping; measured with
    # tools/host_cost.py: 120 us/call for the first ~600 calls, 12 us afterwards).  Run until half a second has passed.
    # (N > 1: every step is a collective, so all ranks must run the same number -- fixed count, no clock)
    t_prime, n_prime = time.perf_counter(), 0
    while (n_prime < 4096) if world > 1 else (time.perf_counter() - t_prime < 0.5 or n_prime < 1024):
        for _ in range(128):
            step(n_prime)
            n_prime += 1
        if world > 1:
            fence()
    fence()
    primed_iterations = n_prime
    if args.closed_loop:   # back to the start of the course (the set-up iterations above moved the robot)
        ctl.resident_set_pose(cl_start)
        ctl.set_nominal(np.zeros((p.horizon - 1, p.udim)))
    for i in range(args.warmup):
        step(i)
    fence()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    # Kernel durations: a separate pass right after the timed region (same inputs, same warm start), hipEvents attached
    # to the rollout kernel's dispatch and to the end of the launch sequence of EVERY launch, on `stream` -- at least 256
    # launches whatever --steps is (recording inside the timed region would cost ~8 us/step of launch serialisation).
    n_event_pass = 0 if args.no_kernel_events else max(256, min(args.steps, 2048))
    roll_us = iter_us = 0.0
    n_ev = 0
    if n_event_pass:
        ctl.timing_enable(True, every=1)
        ctl.timing_read(reset=True)
        for i in range(n_event_pass):
            step(args.warmup + args.steps + i)
        fence()
        roll_us, iter_us, n_ev = ctl.timing_read(reset=True)
        ctl.timing_enable(False)
    u_final = ctl.get_nominal()
    if not np.all(np.isfinite(u_final)):
        print("bench.py: non-finite controls after the timed region", file=sys.stderr)
        sys.exit(4)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        B, B_roll = algorithmic_bytes(p.horizon, p.udim)
        roll_avg_s = roll_us / max(n_ev, 1) * 1e-6
        iter_avg_s = iter_us / max(n_ev, 1) * 1e-6
        achieved = B_roll * k_local / roll_avg_s / 1e9 if roll_avg_s > 0 else None
        # the committed PMC pass is of the workload's own launch (its K, its kernel): no figure for any other launch
        pmc_applies = world == 1 and k_local == w.params.num_samples and args.dt is None and not args.no_state_store
        traffic = latest_pmc_traffic(args.workload) if pmc_applies else None
        traffic_source = ("rocprofv3 --pmc pass of the same workload and kernel, committed under profiles/ (not measured by this run%s)"
                          % ("" if args.no_live_traffic else ": " + str(live_detail))) if traffic else None
        if pmc_applies and live_traffic:
            traffic = live_traffic
            traffic_source = {"measured": "by this run: two child runs of this command under rocprofv3 --pmc, before the timed region",
                              **live_detail}
        out = {
            # BASELINE.json's metric string; `value` is its first component, `ms_per_step` the second (ms per MPPI iteration)
            "metric": "trajectory rollouts/s (K\u00d7iters/s) + ms/MPPI-iteration, diff-drive T=50" if args.workload == "C2"
                      else "trajectory rollouts/s (K\u00d7iters/s) + ms/MPPI-iteration",
            "value": k_total * args.steps / elapsed,
            "unit": "rollouts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            # the arithmetic is fp64 throughout (rollout, cost, exp, reductions); the N(0,1) variates are made in specified
            # fp32 arithmetic from one 32-bit Philox word each (|z| <= 6.66, 2^-24 grid), then sigma * z + u* in fp64
            "noise": "philox4x32-10 + fp32 Box-Muller (DESIGN.md section 3); the reference draws fp64 polar-method normals from mt19937",
            "primed_iterations": primed_iterations,
            "config": {"workload": "%s: %s, u_dim=%d, launch parameters" % (w.name, w.description.replace(
                "K=%d" % p.num_samples, "K=%d" % k_total), p.udim),
                       "samples_per_gpu": k_local, "horizon": p.horizon, "sharding": "K over %d GPU(s)" % world,
                       # untimed set-up iterations before the W warm-up steps (the host side of the first several hundred
                       # launches of a process is slow): the steady state `value` is quoted in
                       "primed_iterations": primed_iterations,
                       **({"exchange": exchange_used} if exchange_used else {}),
                       **({"exchange_detail": exchange_info} if exchange_info else {}),
                       "state_store": not args.no_state_store,
                       **({"closed_loop": "device-resident: plant + get_CurrentIndex + calc_RefPath on the device, "
                                          "%d path poses" % len(cl_px)} if args.closed_loop else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         # `frac` divides the CONTRACT's algorithmic bytes (fp64 controls written and re-read)
                         # by the kernel time; the kernel really moves `traffic` bytes (it stores the fp32 normals and
                         # re-derives the controls), i.e. measured_traffic_frac of the 8 TB/s -- what limits it is VALU issue
                         # (profiles/*_pmc_summary.txt: SQ_ACTIVE_INST_VALU), not HBM
                         "measured_traffic_frac": (traffic / roll_avg_s / 1e9 / HBM_PEAK_GBS) if (traffic and roll_avg_s > 0) else None,
                         "traffic_source": traffic_source,
                         "limiter": "valu-issue (contract bound: hbm)",
                         "kernel": rollout_kernel_name(p.model, k_local, local_rank) if args.dt is None else
                                   rollout_kernel_name(p.model, k_local, local_rank) + " (its full-range sin/cos instantiation beyond the small-turn gate)",
                         "kernel_avg_us": 1e6 * roll_avg_s,
                         "kernel_launches_averaged": int(n_ev),
                         "algorithmic_bytes_per_launch": B_roll * k_local,
                         "iteration_avg_us": 1e6 * iter_avg_s,
                         "iteration_algorithmic_bytes": B * k_local,
                         "iteration_achieved": (B * k_local / iter_avg_s / 1e9) if iter_avg_s > 0 else None,
                         "iteration_frac": (B * k_local / iter_avg_s / 1e9 / HBM_PEAK_GBS) if iter_avg_s > 0 else None},
        }
        headline = (world == 1 and not args.closed_loop and args.workload == "C2" and args.samples_per_gpu is None
                    and args.dt is None and args.path is None and not args.no_state_store)
        if world == 1:
            out["roofline"]["device_copy_gbs"] = device_copy_gbs(torch)
            if not args.closed_loop:
                # SURVEY.md 8(d), end to end through the blocking C-ABI call: host window in, u* out (PCIe and the host
                # synchronisation included), closed loop on the host plant.  Never `value`.
                px_e, py_e = amd.make_path(w.path)
                s_e = np.zeros(p.nstate)
                s_e[0], s_e[1] = px_e[0], py_e[0]
                ctl.set_nominal(np.zeros((p.horizon - 1, p.udim)))
                lat = []
                for i in range(220):
                    t_e = time.perf_counter()
                    _, xr_e, yr_e, yaw_e = amd.calc_ref_path(px_e, py_e, s_e[0], s_e[1], p.v_ref, p.dt, p.resolution, p.horizon)
                    u_e = ctl.iterate(s_e, p.dt, xr_e, yr_e, yaw_e[0], seed, i, want_stats=False)
                    lat.append(time.perf_counter() - t_e)
                    s_e = amd.plant_step(p.model, s_e, u_e[0], p.dt)
                    if np.hypot(px_e[-1] - s_e[0], py_e[-1] - s_e[1]) < 1.0:   # end of the course: start over
                        s_e = np.zeros(p.nstate)
                        s_e[0], s_e[1] = px_e[0], py_e[0]
                out["end_to_end"] = {"python_binding_us_median": 1e6 * float(np.median(lat[20:])),
                                     "what": "per tick of a closed loop on the host plant, 200 ticks after 20 -- blocking_iterate: "
                                             "ONE C call, ccv_mppi_node_run_once() of the ROS-free node mirror = calc_RefPath() on the "
                                             "host + blocking ccv_mppi_iterate (window in the kernel arguments, u* back through the "
                                             "pinned result mailbox) + publish_CmdVel/CmdPos: what a node that drops the library in "
                                             "pays; python_binding: the same two calls through the ctypes wrappers (numpy conversions "
                                             "included)"}
                if headline:
                    from ccv_mppi_path_tracker_amd.node import ControllerNode
                    node = ControllerNode(p.model, {"num_samples": p.num_samples, "horizon": p.horizon, "v_ref": p.v_ref,
                                                    "v_max": p.u_max[0], "path_weight": p.path_weight}, device=local_rank)
                    node.set_path(px_e, py_e)
                    s_n, lat_n = np.zeros(5), []
                    s_n[0], s_n[1] = px_e[0], py_e[0]
                    for i in range(220):
                        node.set_state(s_n)
                        t_e = time.perf_counter()
                        node.run_once(p.dt)
                        lat_n.append(time.perf_counter() - t_e)
                        s_n[:p.nstate] = amd.plant_step(p.model, s_n[:p.nstate], node.optimal_solution()[0], p.dt)
                        if np.hypot(px_e[-1] - s_n[0], py_e[-1] - s_n[1]) < 1.0:
                            s_n[:] = 0.0
                            s_n[0], s_n[1] = px_e[0], py_e[0]
                    node.close()
                    out["end_to_end"]["blocking_iterate_us_median"] = 1e6 * float(np.median(lat_n[20:]))
        if args.closed_loop:
            tr = ctl.resident_read_trace()
            d = np.hypot(cl_px[None, :] - tr[:, 0:1], cl_py[None, :] - tr[:, 1:2]).min(axis=1)
            out["closed_loop"] = {"ticks": int(len(tr)), "distance_travelled_m": float(np.hypot(np.diff(tr[:, 0]), np.diff(tr[:, 1])).sum()),
                                  "path_error_rms_m": float(np.sqrt(np.mean(d * d))), "path_error_max_m": float(d.max())}
        if world == 1 and not args.closed_loop and args.workload == "C2" and not args.no_closed_loop_leg:
            out["closed_loop"] = closed_loop_leg(amd, torch, ctl, w, seed)
        if headline and not args.no_other_workloads:
            # BASELINE configs[2] and configs[3], measured like the headline (their own handles; C2's is released first)
            ctl.close()
            out["other_workloads"] = {"C3": measure_workload(amd, torch, configs, "C3", 100, 10, stream, local_rank),
                                      "C4": measure_workload(amd, torch, configs, "C4", 60, 10, stream, local_rank)}
        if headline and not args.no_defaults_leg:
            out["defaults"] = defaults_leg(amd, local_rank, with_cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
            out["cpu_baseline_reference_shaped"] = cpu_baseline(w, budget_s=10.0, by_value=True)
            out["cpu_baseline_all_cores"] = cpu_baseline_threads(w, max(1, host_cores()))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
