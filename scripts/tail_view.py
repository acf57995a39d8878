"""Last stretch of every queue of the last replayed step in a rocprofv3 kernel trace: what the step's end waits for.
usage: tail_view.py <kernel_trace.csv> [from_ms]"""
import csv, sys
import gzip
rows = list(csv.DictReader(gzip.open(sys.argv[1], "rt") if sys.argv[1].endswith(".gz") else open(sys.argv[1])))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 27.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows))
st = [i for i, e in enumerate(ev) if "stem_im2col" in e[3]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2      # which step from the end (bench.py's last step carries the roofline timers)
seg = ev[st[-1 - k]:st[-k] + 3]; t0 = seg[0][0]
print(f"step {(ev[st[-k]][0] - t0) / 1e6:.3f} ms")
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("_ZN12_GLOBAL__N_1", "")[:56]
for s, e, q, n, g in seg:
    t = (s - t0) / 1e6
    if t > lo:
        print(f"q{q} {t:7.3f}..{(e - t0) / 1e6:7.3f} {(e - s) / 1e3:7.1f} us blocks={g:5d} {short(n)}")
