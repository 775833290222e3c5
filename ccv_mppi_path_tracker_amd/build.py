"""Build libccv_mppi_hip.so (the C-ABI + gfx950 kernels) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU; the built .so is git-ignored but travels to
the GPU box with the gpurun snapshot.
"""
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIBDIR = os.path.join(_PKG, "lib")
LIB = os.path.join(LIBDIR, "libccv_mppi_hip.so")
# one translation unit per rollout kernel family (csrc/mppi_launch.h) + the C ABI + the host prologue + the node mirror: they
# compile side by side
KERNEL_UNITS = ["k_r4_fb.hip", "k_r4.hip", "k_r3.hip", "k_pc.hip", "k_pc_fb.hip", "k_solo.hip", "k_solo_fb.hip", "k_plain.hip"]
SOURCES = KERNEL_UNITS + ["ccv_mppi_capi.hip", "ccv_mppi_host.cpp", os.path.join("host", "mppi_node.cpp")]
HEADERS = ["mppi_kernels.h", "mppi_update.h", "mppi_launch.h", "mppi_rollout_pc.h", "mppi_rollout_r3.h", "mppi_rollout_r4.h",
           "mppi_rollout_solo.h", "mppi_resident.h", "fast_trig.h", "noise_spec.h",
           os.path.join("..", "..", "include", "ccv_mppi.h"), os.path.join("..", "..", "include", "ccv_mppi_host.h"),
           os.path.join("..", "..", "include", "ccv_mppi_node.hpp")]
DEPS = SOURCES + HEADERS

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # a*b+c is fused only where fma() is written: the rollout mirrors the reference's rounding,
    # and the fp32 noise spec must be bit-reproducible on the CPU
    "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-result",
]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X path cannot be built (there is no CPU fallback)")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    if verbose and res.stderr:
        print(res.stderr)


def build(force=False, verbose=False, extra_flags=(), out=None, jobs=None):
    """Compile every HIP source for gfx950 into LIB (or `out` for experiment builds); returns the path.  The translation
    units are compiled in parallel (`jobs` at a time, default: the CPUs this process may use, at most 8) and linked.
    Several processes may call this at once (the ranks of a launch that finds the library stale): one builds, under a file
    lock and into a directory of its own, the others wait and take what it made."""
    if out is None and not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    target = out or LIB
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if out is None and not force and not stale():   # (another process built it while this one waited)
                return LIB
            objdir = os.path.join(os.path.dirname(os.path.abspath(target)), "obj_%s_%d" % (os.path.basename(target), os.getpid()))
            os.makedirs(objdir, exist_ok=True)
            cc = hipcc()
            objs = [os.path.join(objdir, os.path.basename(s).rsplit(".", 1)[0] + ".o") for s in SOURCES]
            cmds = [[cc] + HIPCC_FLAGS + list(extra_flags) + ["-c", "-o", o, os.path.join(CSRC, s)] for s, o in zip(SOURCES, objs)]
            if jobs is None:
                try:
                    jobs = len(os.sched_getaffinity(0))
                except AttributeError:
                    jobs = os.cpu_count() or 1
                jobs = max(1, min(8, jobs))
            from concurrent.futures import ThreadPoolExecutor
            try:
                with ThreadPoolExecutor(jobs) as pool:
                    list(pool.map(lambda c: _run(c, verbose), cmds))
                tmp = target + ".tmp%d" % os.getpid()
                _run([cc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp] + objs, verbose)
                os.replace(tmp, target)   # (atomic: a process that has the old library open keeps it)
            finally:
                shutil.rmtree(objdir, ignore_errors=True)   # (every build compiles everything: the objects are of no further use)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return target


if __name__ == "__main__":
    print(build(force=True, verbose=True))
