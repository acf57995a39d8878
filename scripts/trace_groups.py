"""Group a rocprofv3 kernel trace (csv) by (kernel, grid): python scripts/trace_groups.py <kernel_trace.csv> <steps> [top]."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1]))); steps = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
def short(n):
    m = re.search(r"gemm_dma_kernelIDF16bLi(\d)ELi(\d)ENS_5ShapeILi(\d)ELi(\d)ELi(\d)ELi(\d)EEELi(\d)", n)
    if m: return "gemm_dma A%s B%s S%s%s%s%s asum%s" % m.groups()
    m = re.search(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)I", n)
    if m: return m.group(1)
    return re.sub(r"\(anonymous namespace\)::", "", n)[:60]
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
    a = agg[k]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
print(f"total {tot/1e6/steps:.2f} ms/step, {len(rows)/steps:.0f} launches/step")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{a[1]/1e6/steps:7.3f} ms  {a[0]/steps:6.1f}/step  {a[1]/a[0]/1e3:8.1f} us  blocks={k[1]:6d} y={k[2]:4d}  {k[0]}")
