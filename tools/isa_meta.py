#!/usr/bin/env python3
"""Register / spill / scratch figures of every kernel of some translation units, compiled with the product's own flags
(build.HIPCC_FLAGS) to assembly under /tmp: `python tools/isa_meta.py TAG k_solo_fb k_r4 ...` -> /tmp/isa/<unit>_<TAG>.s
(extra compiler flags: ISA_FLAGS="-DX -DY")"""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ccv_mppi_path_tracker_amd import build

def main():
    tag, units = sys.argv[1], sys.argv[2:]
    os.makedirs("/tmp/isa", exist_ok=True)
    flags = [f for f in build.HIPCC_FLAGS if f != "-fPIC"] + os.environ.get("ISA_FLAGS", "").split()
    def one(u):
        out = f"/tmp/isa/{u}_{tag}.s"
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-I", build.CSRC, "-I", os.path.join(ROOT, "include"),
                        os.path.join(build.CSRC, u + ".hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
        return out
    with ThreadPoolExecutor(8) as ex:
        outs = list(ex.map(one, units))
    for out in outs:
        print("==", out)
        text = open(out).read()
        for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
            body = m.group(2)
            g = lambda k: re.search(r"\." + k + r":\s+(\d+)", body).group(1)
            name = m.group(1).replace("_ZN3ccv", "").replace("NS_11RolloutArgsENS_6WindowE", "")
            print(f"  {name:44s} vgpr {g('vgpr_count'):>4s} (spilled {g('vgpr_spill_count'):>3s})  sgpr spilled {g('sgpr_spill_count'):>3s}  scratch {g('private_segment_fixed_size'):>4s} B")

if __name__ == "__main__":
    main()
