"""Idle gaps on each queue of the last replayed step of a rocprofv3 --kernel-trace CSV: total, histogram, and the largest gaps with the
kernels on either side (what the critical chain waits for).  usage: chain_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
sys.path.insert(0, __file__.rsplit("/", 1)[0])
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows))
starts = [i for i, e in enumerate(ev) if "stem_im2col" in e[3]]      # once per step, the first kernel of the image forward
seg = ev[starts[-2]:starts[-1]]
t0 = seg[0][0]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("_ZN12_GLOBAL__N_1", "")[:46]
for q in sorted({e[2] for e in seg}):
    L = [e for e in seg if e[2] == q]
    gaps = []
    for a, b in zip(L, L[1:]):
        gaps.append((b[0] - a[1], a, b))
    tot = sum(max(0, g[0]) for g in gaps)
    hist = collections.Counter()
    for g, _, _ in gaps:
        hist["<1us" if g < 1000 else "1-3us" if g < 3000 else "3-6us" if g < 6000 else "6-12us" if g < 12000 else "12-50us" if g < 50000 else ">50us"] += 1
    print(f"queue {q}: {len(L)} launches, span {(L[0][0]-t0)/1e6:.2f}..{(L[-1][1]-t0)/1e6:.2f} ms, idle between kernels {tot/1e6:.2f} ms; gaps {dict(hist)}")
    for g, a, b in sorted(gaps, key=lambda x: -x[0])[:12]:
        print(f"    {g/1e3:8.1f} us at {(a[1]-t0)/1e6:6.2f} ms   {short(a[3])}  ->  {short(b[3])}")
    print("    head of the queue:")
    for a in L[:12]:
        print(f"      {(a[0]-t0)/1e6:7.3f} .. {(a[1]-t0)/1e6:7.3f} ms  {(a[1]-a[0])/1e3:7.1f} us  {short(a[3])}")
    print("    tail of the queue:")
    for a in L[-14:]:
        print(f"      {(a[0]-t0)/1e6:7.3f} .. {(a[1]-t0)/1e6:7.3f} ms  {(a[1]-a[0])/1e3:7.1f} us  {short(a[3])}")
