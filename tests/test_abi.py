"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol the public headers declare,
the ctypes mirrors of the structs match the C layout, and there is no CPU fallback."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from ccv_mppi_path_tracker_amd import build, capi, configs
from ccv_mppi_path_tracker_amd.controller import MPPIController, MPPIError, make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("ccv_mppi.h", "ccv_mppi_host.h", "ccv_mppi_node.hpp")]


def _declared():
    names = set()
    for h in HEADERS:
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        if h.endswith(".hpp"):
            src = src[src.index('extern "C"'):]   # only the plain-C part of the C++ header is an ABI
        names |= set(re.findall(r"\b(ccv_mppi_[a-z_0-9]+)\s*\(", src))
    return names


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build()
    assert os.path.exists(path)
    lib = C.CDLL(path)
    declared = _declared()
    assert len(declared) >= 38
    for name in declared:
        assert hasattr(lib, name), "libccv_mppi_hip.so does not export %s" % name
    assert declared == set(capi.SIGNATURES), "ctypes table and headers disagree"
    assert b"gfx950" in capi.load().ccv_mppi_version()


def test_two_processes_that_find_the_library_stale_do_not_race():
    """The ranks of a multi-GPU launch import the package at the same moment; if the library is older than a source, each of
    them used to compile into the same object directory and remove it when done (one rank's link then failed).  Now one builds
    under a file lock and the others wait for it: both calls succeed and the library is fresh afterwards."""
    lib = build.build()
    old = os.path.getmtime(lib) - 10 * 365 * 86400.0
    os.utime(lib, (old, old))                       # stale: older than every source
    assert build.stale()
    code = "from ccv_mppi_path_tracker_amd import build; print(build.build())"
    procs = [subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
        assert out.strip().endswith("libccv_mppi_hip.so")
    assert not build.stale()
    assert not [d for d in os.listdir(build.LIBDIR) if d.startswith("obj_")]   # nobody's object directory is left behind


def test_struct_layout_matches_c(tmp_path):
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ccv_mppi.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ccv_mppi_config),'
                   'offsetof(ccv_mppi_config, control_noise), offsetof(ccv_mppi_config, u_max),'
                   'offsetof(ccv_mppi_config, yaw_weight), sizeof(ccv_mppi_stats),'
                   'offsetof(ccv_mppi_stats, n_zero_weight), offsetof(ccv_mppi_stats, rollout_us));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(capi.Config), capi.Config.control_noise.offset, capi.Config.u_max.offset,
            capi.Config.yaw_weight.offset, C.sizeof(capi.Stats), capi.Stats.n_zero_weight.offset,
            capi.Stats.rollout_us.offset]
    assert got == want


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "inc.c"
    src.write_text('#include "ccv_mppi.h"\n#include "ccv_mppi_host.h"\nint main(void){return CCV_MPPI_ABI_VERSION - 1;}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                    "-o", str(tmp_path / "inc")], check=True)


def test_invalid_config_is_rejected_before_touching_a_device():
    lib = capi.load()
    h = capi._H()
    good = configs.diff_drive_defaults(64, 15)
    for mutate in (lambda c: setattr(c, "abi_version", 99), lambda c: setattr(c, "model", 7),
                   lambda c: setattr(c, "num_samples", 0), lambda c: setattr(c, "horizon", 2),
                   lambda c: setattr(c, "horizon", capi.MAX_HORIZON + 1), lambda c: setattr(c, "sample_offset", -1)):
        cfg = make_config(good)
        mutate(cfg)
        assert lib.ccv_mppi_create(C.byref(cfg), C.byref(h)) == capi.ERR_INVALID_ARG
        assert not h.value
    assert lib.ccv_mppi_create(None, C.byref(h)) == capi.ERR_INVALID_ARG
    assert lib.ccv_mppi_udim(0) == 2 and lib.ccv_mppi_udim(1) == 3 and lib.ccv_mppi_udim(2) == 5
    assert lib.ccv_mppi_udim(3) == capi.ERR_INVALID_ARG
    assert lib.ccv_mppi_destroy(None) == capi.ERR_INVALID_ARG


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(MPPIError) as ei:
        MPPIController(configs.diff_drive_defaults(64, 15))
    assert ei.value.code in (capi.ERR_NO_DEVICE, capi.ERR_HIP)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "ccv_mppi_path_tracker_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "liboracle" not in text and "libmppi_oracle" not in text, f
                assert not re.search(r'#include\s+".*oracle', text), f
    code = "import sys; import ccv_mppi_path_tracker_amd as m; m.capi.load(); " \
           "assert not [k for k in sys.modules if k.split('.')[0] == 'oracle']"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


def test_node_api_rejects_bad_arguments_and_has_no_fallback():
    import ctypes as C
    import torch
    lib = capi.load()
    h = capi._H()
    assert lib.ccv_mppi_node_create(9, None, None, 0, 0, C.byref(h)) == capi.ERR_INVALID_ARG
    assert lib.ccv_mppi_node_create(0, None, None, 0, 0, None) == capi.ERR_INVALID_ARG
    assert lib.ccv_mppi_node_destroy(None) == capi.ERR_INVALID_ARG
    if not torch.cuda.is_available():
        from ccv_mppi_path_tracker_amd import ControllerNode
        with pytest.raises(MPPIError):
            ControllerNode("diff_drive", {"num_samples": 64})
