"""Kernel time by family over the last K replayed steps of a rocprofv3 kernel trace (CSV or .csv.gz): GEMM forward / data gradient,
weight gradient, BatchNorm, LayerNorm, attention, optimizer, rest.  usage: family_breakdown.py <kernel_trace.csv[.gz]> [K=3]"""
import csv, gzip, re, sys, collections
f = sys.argv[1]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(gzip.open(f, "rt") if f.endswith(".gz") else open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "stem_im2col" in r["Kernel_Name"]]
rows = rows[st[-1 - k]:st[-1]]
def family(n):
    if "gemm_ppt" in n: return "weight gradient (ping-pong 256x256 / grouped)"
    if "conv_sw" in n: return "3x3 conv forward + data gradient"
    m = re.search(r"gemm_dma_kernelIDF16bLi(\d)ELi(\d)E", n)
    lay = (int(m.group(1)), int(m.group(2))) if m else None
    if lay is None:
        m2 = re.search(r"gemm_dma_kernel<bool _Accum, int, E, (\d), ", n)
        if m2: lay = (1, int(m2.group(1)))
        elif "gemm_dma_kernel<bool _Accum, int, EL, int, E," in n: lay = (2, 1)
    m3 = re.search(r"gemm_(?:dma16|pp)_kernel(?:<|ILi)(\d)", n)
    if m3: lay = (int(m3.group(1)), 0)
    if lay is not None:
        if lay[0] == 2: return "weight gradient (128x128 / 64x64 tiles)"
        return "3x3 conv forward + data gradient" if lay[0] == 1 else "1x1 / linear forward + data gradient, attention products"
    if "gemm" in n: return "other GEMM"
    if "bn_" in n: return "BatchNorm"
    if "ln_" in n: return "LayerNorm"
    if "flash" in n or "softmax" in n: return "attention (flash / softmax)"
    if "adam" in n: return "Adam"
    return "rest"
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    a = agg[family(r["Kernel_Name"])]
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
tot = sum(a[1] for a in agg.values())
print(f"last {k} steps: {tot / k:.2f} ms of kernel time per step, {sum(a[0] for a in agg.values()) / k:.0f} launches per step")
for name, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{d / k:7.3f} ms {100 * d / tot:5.1f} % {n / k:6.0f} launches  {name}")
