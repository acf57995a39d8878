"""Group a rocprofv3 kernel trace by (kernel, grid): ms per step, launches per step, mean duration - the table behind profiles/*_by_grid.txt.
usage: python scripts/by_grid.py <kernel_trace.csv> <steps executed> [rows] [--last K]
--last K: only the last K complete steps of the trace (a step starts at its stem_im2col launch), i.e. replayed steps only."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
if "--last" in sys.argv:
    k = int(sys.argv[sys.argv.index("--last") + 1])
    del sys.argv[sys.argv.index("--last"):sys.argv.index("--last") + 2]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    st = [i for i, r in enumerate(rows) if "stem_im2col" in r["Kernel_Name"]]
    rows = rows[st[-1 - k]:st[-1]]
    steps = float(k)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"]
    if "probe" in name: continue
    g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) if "Grid_Size_X" in r else 0
    gy = int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1)))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    a = agg[(name, g, gy)]
    a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
print(f"total {tot / steps:.2f} ms/step, {sum(a[0] for a in agg.values()) / steps:.0f} launches/step")
short = lambda n: re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*", "", n)[:110]
for (name, g, gy), (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print(f"{d / steps:7.3f} ms {n / steps:7.1f}/step {d / n * 1e3:8.1f} us  blocks={g:6d} y={gy:4d}  {short(name)}")
