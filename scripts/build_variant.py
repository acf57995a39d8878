"""Build an A/B variant of the kernel library: scripts/build_variant.py NAME [-DFLAG ...] -> crog_amd/variants/libcrog_NAME.so
(objects in crog_amd/csrc/build_NAME/; select it at run time with CROG_LIB=crog_amd/variants/libcrog_NAME.so).  --packed: build WITH packed-fp32 instructions."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crog_amd import _lib
name, flags = sys.argv[1], sys.argv[2:]
base = list(_lib.HIPCC_FLAGS)
if "--packed" in flags:      # WITH the packed-fp32 VALU instructions (the victim build of scripts/pk_probe.py): drop the target-feature switch
    flags.remove("--packed")
    base = [f for f in base if f not in _lib.NO_PACKED_F32 and f != "-DCROG_NO_PACKED_F32=1"]
bdir = os.path.join(_lib.CSRC, "build_" + name)
os.makedirs(bdir, exist_ok=True)
os.makedirs(os.path.join(ROOT, "crog_amd", "variants"), exist_ok=True)
out = os.path.join(ROOT, "crog_amd", "variants", f"libcrog_{name}.so")
def one(src):
    o = os.path.join(bdir, src.replace(".hip", ".o"))
    r = subprocess.run(["hipcc"] + base + _lib.EXTRA_FLAGS.get(src, []) + flags + ["-c", os.path.join(_lib.CSRC, src), "-o", o], capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr)
    return o
with ThreadPoolExecutor(max_workers=6) as ex:
    objs = list(ex.map(one, _lib.SOURCES))
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)
if r.returncode:
    raise SystemExit(r.stderr)
print(out)
