"""Exec-mask branches per kernel (s_and_saveexec / s_cbranch_execz) and how many of them guard a memory load directly (a load followed by
its own wait inside the guarded block: the pattern that serialised the attention backward kernels, LAB_NOTES section 10).
usage: python scripts/count_branches.py [source.hip ...]   (default: every source of the library)"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crog_amd import _lib
srcs = sys.argv[1:] or _lib.SOURCES

def one(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "x.s")
        flags = [f for f in _lib.HIPCC_FLAGS if f != "-fPIC"] + _lib.EXTRA_FLAGS.get(src, [])
        r = subprocess.run(["hipcc"] + flags + ["-S", "--cuda-device-only", os.path.join(_lib.CSRC, src), "-o", out], capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stderr)
        res, cur, lines = [], None, []
        for line in open(out):
            m = re.match(r"^(_Z\w+|crog\w*):", line)
            if m:
                if cur: res.append((cur, lines))
                cur, lines = m.group(1), []
            elif cur:
                lines.append(line)
        if cur: res.append((cur, lines))
        rows = []
        for name, L in res:
            se = sum("s_and_saveexec" in l for l in L)
            if se < 8: continue
            inloop = sum("s_and_saveexec" in l for i, l in enumerate(L) if any("in Loop" in x or "Loop Header" in x for x in L[max(0, i - 40):i + 1]))
            guarded = 0
            for i, l in enumerate(L):
                if "s_cbranch_execz" in l:
                    blk = L[i + 1:i + 14]
                    if any(re.search(r"\b(global_load|ds_read|buffer_load)", x) for x in blk) and any("s_waitcnt" in x for x in blk):
                        guarded += 1
            rows.append((src, name[:70], len(L), se, inloop, guarded))
        return rows

with ThreadPoolExecutor(max_workers=6) as ex:
    allrows = [r for rows in ex.map(one, srcs) for r in rows]
print(f"{'source':14s} {'kernel':70s} {'lines':>6s} {'saveexec':>8s} {'near loop':>9s} {'load+wait guarded':>17s}")
for r in sorted(allrows, key=lambda r: -r[5])[:60]:
    print(f"{r[0]:14s} {r[1]:70s} {r[2]:6d} {r[3]:8d} {r[4]:9d} {r[5]:17d}")
