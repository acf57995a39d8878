"""Timeline analysis of a rocprofv3 --kernel-trace CSV: per-queue busy time, union of busy intervals, idle gaps, concurrency.
usage: python scripts/timeline.py <kernel_trace.csv> [steps]   (the last `steps` Adam launches delimit the analysed window)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
# steps are delimited by the LAST adam launch of each step (adam launches come in groups)
groups = []
for i in adam:
    if groups and rows[i]["s"] - rows[groups[-1][-1]]["e"] < 2e6:
        groups[-1].append(i)
    else:
        groups.append([i])
print("steps seen:", len(groups))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo = rows[groups[-n - 1][-1]]["e"]
hi = rows[groups[-1][-1]]["e"]
win = [r for r in rows if r["s"] >= lo and r["e"] <= hi]
T = (hi - lo) / n / 1e6
print(f"window: {n} steps, {T:.3f} ms/step, {len(win)/n:.0f} kernels/step")
# union of busy intervals
ev = sorted([(r["s"], 1) for r in win] + [(r["e"], -1) for r in win])
busy = 0; depth = 0; last = lo; conc = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last
    conc[depth] += t - last
    depth += d; last = t
print(f"GPU busy (union): {busy/n/1e6:.3f} ms/step = {busy/(hi-lo)*100:.1f}%   idle {(hi-lo-busy)/n/1e6:.3f} ms/step")
print("time at concurrency depth (ms/step):", {k: round(v / n / 1e6, 2) for k, v in sorted(conc.items())})
byq = collections.defaultdict(list)
for r in win: byq[(r["Queue_Id"], r.get("Stream_Id", "?"))].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -sum(r["e"] - r["s"] for r in kv[1])):
    tot = sum(r["e"] - r["s"] for r in rs)
    gaps = [b["s"] - a["e"] for a, b in zip(rs, rs[1:]) if b["s"] > a["e"]]
    small = [g for g in gaps if g < 50000]
    print(f"queue/stream {q}: {len(rs)/n:.0f} kernels/step, busy {tot/n/1e6:.3f} ms/step; gaps<50us: n={len(small)/n:.0f}/step sum {sum(small)/n/1e6:.3f} ms/step "
          f"median {sorted(small)[len(small)//2]/1e3 if small else 0:.1f} us; gaps>=50us sum {sum(g for g in gaps if g >= 50000)/n/1e6:.3f} ms/step")
# scratch users
sc = collections.Counter()
for r in win:
    if int(r["Scratch_Size"]) > 0: sc[(r["Kernel_Name"][:90], r["Scratch_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"])] += 1
for k, v in sc.most_common(12): print("scratch:", v // n, k)
