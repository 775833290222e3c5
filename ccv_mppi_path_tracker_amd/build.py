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
SOURCES = ["ccv_mppi_capi.hip", "ccv_mppi_host.cpp", os.path.join("host", "mppi_node.cpp")]
DEPS = ["ccv_mppi_capi.hip", "ccv_mppi_host.cpp", os.path.join("host", "mppi_node.cpp"),
        os.path.join("..", "..", "include", "ccv_mppi_node.hpp"), "mppi_kernels.h", "mppi_rollout_pc.h", "mppi_rollout_r3.h", "mppi_rollout_r4.h", "mppi_rollout_solo.h", "mppi_resident.h", "fast_trig.h", "noise_spec.h",
        os.path.join("..", "..", "include", "ccv_mppi.h"), os.path.join("..", "..", "include", "ccv_mppi_host.h")]

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
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


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every HIP source for gfx950 into LIB (or `out` for experiment builds); returns the path."""
    if out is None and not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    target = out or LIB
    cmd = [hipcc()] + HIPCC_FLAGS + list(extra_flags) + ["-o", target] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose and res.stderr:
        print(res.stderr)
    return target


if __name__ == "__main__":
    print(build(force=True, verbose=True))
