"""Per-kernel totals from a rocprofv3 results.db (rocpd sqlite): python scripts/prof_db.py <db> [steps]  -> name, ms/step, launches/step, avg us."""
import sqlite3, sys, collections, subprocess, re
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
rows = cur.execute("select name, start, end from kernels").fetchall()
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e in rows:
    a = agg[n]; a[0] += 1; a[1] += e - s
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n[:150]
tot = sum(a[1] for a in agg.values())
print(f"total kernel time {tot/1e6/steps:.2f} ms/step over {steps} steps, {len(rows)/steps:.0f} launches/step")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{a[1]/1e6/steps:8.3f} ms  {a[0]/steps:7.1f}/step  {a[1]/a[0]/1e3:8.1f} us  {short(n)}")
