"""Which kernels of the library contain packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)?  They are what the
compiler's SLP vectoriser makes of adjacent scalar fp32 operations, and round 5 found their results wrong in lanes 48-63 when an MFMA kernel
of another stream shares the SIMD (LAB_NOTES section 10, scripts/pk_probe.py).  usage: python scripts/count_pk.py [extra hipcc flags ...]
Compiles every source of crog_amd/csrc to ISA with the library's flags (+ the extra ones) and counts per kernel."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crog_amd import _lib

extra = sys.argv[1:]
PAT = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")


def one(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "x.s")
        flags = [f for f in _lib.HIPCC_FLAGS if f != "-fPIC"]
        r = subprocess.run(["hipcc"] + flags + extra + ["-S", "--cuda-device-only", os.path.join(_lib.CSRC, src), "-o", out], capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stderr)
        per, cur = {}, None
        for line in open(out):
            m = re.match(r"^(_Z\w+|\w+):\s*;? ?@?", line)
            if m and not line.startswith(".L"):
                cur = m.group(1)
            elif PAT.search(line) and cur:
                per[cur] = per.get(cur, 0) + 1
        return src, per


with ThreadPoolExecutor(max_workers=6) as ex:
    res = list(ex.map(one, _lib.SOURCES))
total = 0
for src, per in res:
    n = sum(per.values())
    total += n
    print(f"{src}: {n} packed-fp32 instructions in {len(per)} kernels")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:8]:
        print(f"    {v:5d}  {k[:120]}")
print(f"total {total}")
