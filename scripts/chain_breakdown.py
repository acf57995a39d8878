"""Per-queue kernel-time breakdown of the last full step in a rocprofv3 --kernel-trace CSV of a replayed run, split at the start of backward.
usage: chain_breakdown.py <kernel_trace.csv>   (STEP_BACK=n: the n-th step before the last full one; .csv.gz accepted)"""
import csv, collections, re, sys
import gzip
rows = list(csv.DictReader(gzip.open(sys.argv[1], "rt") if sys.argv[1].endswith(".gz") else open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows))
# one full step = from one stem_im2col launch (the first kernel of the image forward, once per step) to the next; the last complete one
starts = [i for i, e in enumerate(ev) if "stem_im2col" in e[3]]
import os
back = int(os.environ.get("STEP_BACK", "0"))
lo, hi = starts[-2 - back], starts[-1 - back]
seg = ev[lo:hi]
t0 = seg[0][0]
def short(n):
    n = n.replace('(anonymous namespace)::', '')
    mm = re.search(r'gemm_dma_kernelI(DF16b|f)Li(\d)ELi(\d)ENS_5ShapeILi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELi\dELb(\d)ELb(\d)', n)
    if mm: return f"gemm A{mm.group(2)}B{mm.group(3)} {32*int(mm.group(4))*int(mm.group(6))}x{32*int(mm.group(5))*int(mm.group(7))}{' bwdz' if mm.group(9)=='1' else ''}{' atom' if mm.group(10)=='1' else ''}"
    mm = re.search(r'gemm_dma_kernel<.*?(\d), Shape<(\d), (\d), (\d), (\d), (\w+), \d, (\w+), (\w+)>', n)
    if mm: return f"gemm(?) {32*int(mm.group(2))*int(mm.group(4))}x{32*int(mm.group(3))*int(mm.group(5))}{' bwdz' if mm.group(7)=='true' else ''}{' atom' if mm.group(8)=='true' else ''}"
    mm = re.search(r'gemm_pp_kernel(?:<|ILi)(\d)(?:, |ELi)(\d)', n)
    if mm: return f"gemm_pp A{mm.group(1)} {64*int(mm.group(2))}x256"
    mm = re.search(r'gemm_ppt_kernel(?:<|ILi)(\d)', n)
    if mm: return f"gemm_ppt B{mm.group(1)} 256x256"
    m = re.search(r'_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)I', n)
    if m: return m.group(1)
    return n.split('(')[0].replace('void ', '')[:44]
bwd0 = min(e[0] for e in seg if "bn_bwd" in e[3] or "flash_bwd" in e[3] or "stencil_bwd" in e[3])
print(f"step wall {(seg[-1][1]-t0)/1e6:.2f} ms; backward starts at {(bwd0-t0)/1e6:.2f} ms")
for q in sorted({e[2] for e in seg}):
    for name, L in (("fwd", [e for e in seg if e[2] == q and e[0] < bwd0]), ("bwd", [e for e in seg if e[2] == q and e[0] >= bwd0])):
        if not L: continue
        agg, cnt = collections.Counter(), collections.Counter()
        for s, e, _, n in L: agg[short(n)] += e - s; cnt[short(n)] += 1
        print(f"--- queue {q} {name}: {len(L)} launches, busy {sum(agg.values())/1e6:.2f} ms, span {(L[0][0]-t0)/1e6:.2f}..{(max(e[1] for e in L)-t0)/1e6:.2f} ms")
        for k, v in agg.most_common(11): print(f"   {v/1e6:6.2f} ms {cnt[k]:4d}x  {k}")
