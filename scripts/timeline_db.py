"""Timeline analysis of a rocprofv3 results.db: busy union, per-stream busy time and gaps.  python scripts/timeline_db.py <db> [steps]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = [dict(name=r[0], s=r[1], e=r[2], q=r[3], scratch=r[4], vgpr=r[5]) for r in
        db.execute("select name, start, end, stream_id, scratch_size, vgpr_count from kernels order by start")]
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["name"]]
groups = []
for i in adam:
    if groups and rows[i]["s"] - rows[groups[-1][-1]]["e"] < 2e6: groups[-1].append(i)
    else: groups.append([i])
lo, hi = rows[groups[-n - 1][-1]]["e"], rows[groups[-1][-1]]["e"]
win = [r for r in rows if r["s"] >= lo and r["e"] <= hi]
print(f"{len(groups)} steps seen; window {n} steps: {(hi-lo)/n/1e6:.3f} ms/step, {len(win)/n:.0f} kernels/step")
ev = sorted([(r["s"], 1) for r in win] + [(r["e"], -1) for r in win])
busy = 0; depth = 0; last = lo; conc = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last
    conc[depth] += t - last; depth += d; last = t
print(f"GPU busy (union) {busy/n/1e6:.3f} ms/step; idle {(hi-lo-busy)/n/1e6:.3f}; by depth:", {k: round(v/n/1e6, 2) for k, v in sorted(conc.items())})
byq = collections.defaultdict(list)
for r in win: byq[r["q"]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -sum(r["e"]-r["s"] for r in kv[1])):
    tot = sum(r["e"]-r["s"] for r in rs)
    gaps = [b["s"]-a["e"] for a, b in zip(rs, rs[1:]) if b["s"] > a["e"]]
    small = [g for g in gaps if g < 30000]
    print(f"stream {q}: {len(rs)/n:.0f} kernels/step busy {tot/n/1e6:.3f} ms/step; gaps<30us: {len(small)/n:.0f}/step sum {sum(small)/n/1e6:.3f} ms median {sorted(small)[len(small)//2]/1e3 if small else 0:.1f} us; larger gaps sum {sum(g for g in gaps if g >= 30000)/n/1e6:.3f} ms")
