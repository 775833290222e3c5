#!/usr/bin/env python3
"""Library builds that differ in the compiler flags of ONE translation unit: `python tools/flag_sweep.py k_solo_fb name="-mllvm -x" ...`
-> _abl/lib_fs_<name>.so per variant (all other units compiled once with the product's flags), with the unit's register figures."""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ccv_mppi_path_tracker_amd import build

def obj_of(d, s):
    return os.path.join(d, os.path.basename(s).rsplit(".", 1)[0] + ".o")

def main():
    unit = sys.argv[1] + ".hip"
    variants = [a.split("=", 1) for a in sys.argv[2:]]
    cache = os.path.join(ROOT, "_abl", "obj_fs")
    os.makedirs(cache, exist_ok=True)
    cc = build.hipcc()
    def comp(src, obj, extra):
        subprocess.run([cc] + build.HIPCC_FLAGS + extra + ["--save-temps=obj", "-c", "-o", obj, os.path.join(build.CSRC, src)], check=True,
                       stderr=subprocess.DEVNULL, cwd=cache)
    others = [s for s in build.SOURCES if s != unit]
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda s: comp(s, obj_of(cache, s), []), others))
        def one(v):
            name, flags = v
            d = os.path.join(cache, name)
            os.makedirs(d, exist_ok=True)
            o = obj_of(d, unit)
            comp(unit, o, flags.split())
            lib = os.path.join(ROOT, "_abl", f"lib_fs_{name}.so")
            subprocess.run([cc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib, o] + [obj_of(cache, s) for s in others], check=True)
            asm = [f for f in os.listdir(d) if f.endswith(".s") and "gfx950" in f]
            text = open(os.path.join(d, asm[0])).read() if asm else ""
            out = []
            for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
                g = lambda k: re.search(r"\." + k + r":\s+(\d+)", m.group(2)).group(1)
                out.append(f"vgpr {g('vgpr_count')} spilled {g('vgpr_spill_count')} sgpr-spilled {g('sgpr_spill_count')} scratch {g('private_segment_fixed_size')}")
            return name, flags, out
        for name, flags, out in ex.map(one, variants):
            print(f"{name:12s} {flags:70s} {' | '.join(out)}")

if __name__ == "__main__":
    main()
