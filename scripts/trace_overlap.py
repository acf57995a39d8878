"""Timeline statistics of a rocprofv3 --kernel-trace CSV: per queue busy time, gaps between consecutive kernels, union-busy time,
for the last two steps (split at the adam kernel).  usage: trace_overlap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows))
ends = [i for i, e in enumerate(ev) if "adam_kernel" in e[3] and (i + 1 == len(ev) or "adam_kernel" not in ev[i + 1][3])]
print("kernels", len(ev), "steps found", len(ends))
if len(ends) < 3:
    sys.exit(0)
lo, hi = ends[-3] + 1, ends[-1] + 1          # the last two full steps
seg = ev[lo:hi]
t0, t1 = seg[0][0], max(e[1] for e in seg)
print(f"2 steps: wall {(t1 - t0) / 2e6:.2f} ms/step, launches/step {len(seg) / 2:.0f}, kernel-time sum {sum(e[1] - e[0] for e in seg) / 2e6:.2f} ms/step")
cur_s, cur_e, busy = None, None, 0
for s, e, _, _ in seg:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"GPU busy (union) {busy / 2e6:.2f} ms/step, idle {(t1 - t0 - busy) / 2e6:.2f} ms/step")
byq = collections.defaultdict(list)
for s, e, q, n in seg: byq[q].append((s, e, n))
for q, L in sorted(byq.items()):
    gaps = [L[i + 1][0] - L[i][1] for i in range(len(L) - 1)]
    small = [g for g in gaps if 0 <= g < 50000]
    print(f"queue {q}: {len(L) / 2:.0f} launches/step, busy {sum(e - s for s, e, _ in L) / 2e6:.2f} ms/step, median gap {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us, "
          f"mean gap (<50us) {sum(small) / max(len(small), 1) / 1e3:.1f} us over {len(small)} gaps, gaps >50us: {sum(1 for g in gaps if g >= 50000)}")
