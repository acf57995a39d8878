"""Turn rocprofv3 output (kernel stats + PMC passes) into profiles/<tag>_summary.md and copy the raw stats CSV."""
import collections, csv, glob, os, re, shutil, sys
tag, stats_dir, fetch_dir, write_dir, steps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
PMC_STEPS = int(sys.argv[6]) if len(sys.argv) > 6 else 3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
sf = max(glob.glob(os.path.join(stats_dir, "*", "*_kernel_stats.csv")) + glob.glob(os.path.join(stats_dir, "*_kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(sf, os.path.join(out, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(sf)))
probes_ns = sum(float(r["TotalDurationNs"]) for r in rows if "probe_kernel" in r["Name"])
rows = [r for r in rows if "probe_kernel" not in r["Name"]]      # bench.py's peak probes run once per process: not part of a step
# the runtime's blit kernel: ~950 back-to-back host-to-device uploads of the parameters while the model is built (kernel trace: each one's
# neighbours are other copyBuffer launches; 967 launches with 7 executed steps, 987 with 17), 2 per step for the batch
upload_ns = sum(float(r["TotalDurationNs"]) for r in rows if "rocclr_copyBuffer" in r["Name"])
rows = [r for r in rows if "rocclr_copyBuffer" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
def agg(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    fs = glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv"))
    if not fs: return acc
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]][0] += 1; acc[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return acc
f, w = agg(fetch_dir, "FETCH_SIZE"), agg(write_dir, "WRITE_SIZE")
def clean(n): return re.sub(r"\(anonymous namespace\)::", "", n)
NOTE = os.environ.get("PROFILE_NOTE", "")
L = [f"# rocprofv3 summary — {tag}", ""] + ([NOTE, ""] if NOTE else []) + [
     f"Command: `{os.environ.get('PROFILE_ENV', '')}rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline` (CROG-R50 bf16, B=32, 416x416, 1x MI355X; {steps} steps executed: warm-ups, the eager steps before the capture, the replays).",
     f"PMC passes (separate runs): `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction, MI355X_MICROARCH.md §HBM).", "",
     f"Total kernel time: {tot/1e6/steps:.2f} ms per step ({steps} steps profiled; the peak probes of bench.py, {probes_ns/1e6:.1f} ms once per process, are left out, and so are the {upload_ns/1e6:.1f} ms of `__amd_rocclr_copyBuffer` parameter uploads at model construction).", "",
     "| ms/step | % | launches/step | avg us | HBM MB/launch (PMC) | kernel |", "|---|---|---|---|---|---|"]
for r in rows[:40]:
    k = r["Name"]
    hb = ""
    if k in f and f[k][0]:
        hb = f"{(2*f[k][1]/f[k][0] + (w[k][1]/w[k][0] if k in w and w[k][0] else 0))*1024/1e6:.0f}"
    L.append(f"| {float(r['TotalDurationNs'])/1e6/steps:.3f} | {float(r['Percentage']):.2f} | {int(r['Calls'])/steps:.1f} | {float(r['AverageNs'])/1e3:.1f} | {hb} | `{clean(k)[:110]}` |")
probe = lambda k: "probe_kernel" in k          # bench.py's peak probes (1-GiB copies, MFMA loop) run once per process: not part of a step
tb = sum(2*f[k][1] + w.get(k, [0, 0.0])[1] for k in f if not probe(k)) * 1024
tp = sum(2*f[k][1] + w.get(k, [0, 0.0])[1] for k in f if probe(k)) * 1024
if tb: L += ["", f"Measured HBM traffic of the training steps: {tb/1e9:.1f} GB over the PMC run = {tb/1e9/PMC_STEPS:.1f} GB/step ({PMC_STEPS} steps executed in the PMC runs; "
                 f"the peak probes of bench.py moved another {tp/1e9:.1f} GB and are left out); algorithmic 56.2 GB/step (1.58 GB/img x 32 + 5.6 GB)."]
# per-launch HBM traffic of the GEMM kernels whose (A layout, B layout) can be read off the mangled name; bench.py reports it as roofline.traffic
import json
names = {(1, 0): "conv3x3_fwd", (1, 2): "conv3x3_dgrad", (2, 3): "conv3x3_wgrad", (0, 0): "lin_fwd", (2, 1): "lin_wgrad"}
pm = {}
for k in f:   # one roofline key can cover several tile-shape instantiations of the kernel: aggregate launches and bytes
    m = re.search(r"gemm_dma_kernelIDF16bLi(\d)ELi(\d)E", k)
    lay = (int(m.group(1)), int(m.group(2))) if m else None
    if lay is None:   # rocprofv3 mis-demangles some instantiations: "<bool _Accum, int, E, N, Shape" is (A=1, B=N), "int, EL, int, E," is (2, 1)
        m2 = re.search(r"gemm_dma_kernel<bool _Accum, int, E, (\d), ", k)
        if m2:
            lay = (1, int(m2.group(1)))
        elif "gemm_dma_kernel<bool _Accum, int, EL, int, E," in k:
            lay = (2, 1)
    m3 = re.search(r"gemm_dma16_kernel(?:<|ILi)(\d)", k)      # the 16x16x32 kernel: <AL, Shape>, B is K-contiguous
    if m3:
        lay = (int(m3.group(1)), 0)
    m4 = re.search(r"gemm_pp_kernel(?:<|ILi)(\d)", k)        # the ping-pong kernel (gemm_pp.hip): <AL, RBQ, D>, B is K-contiguous
    if m4:
        lay = (int(m4.group(1)), 0)
    m5 = re.search(r"gemm_ppt_kernel(?:<|ILi)(\d)", k)       # its weight-gradient form (gemm_ppt.hip): <BL, SLAB, D>, A is dy^T
    if m5:
        lay = (2, int(m5.group(1)))
    if "conv_sw_kernel" in k:                                # the sliding-window 3x3 forward / data gradient (conv_sw.hip): 10 of the family's 53 launches
        lay = (1, 0)
    if "wgrad_sw_kernel" in k:                               # ... and the stem / layer1 3x3 weight gradients (wgrad_sw.hip)
        lay = (2, 3)
    if lay in names and f[k][0]:
        key = names[lay]
        e = pm.setdefault(key, dict(kernels=[], launches=0, _bytes=0.0,
                                    source=f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, {tag}"))
        e["kernels"].append(k)
        e["launches"] += f[k][0]
        e["_bytes"] += (2 * f[k][1] + (w[k][1] if k in w else 0)) * 1024
for e in pm.values():
    e["bytes_per_launch"] = round(e.pop("_bytes") / e["launches"])
    try:      # which kernel sources these counters were taken on: bench.py marks the figure stale when the library's sources differ
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from crog_amd import _lib
        e["source_digest"] = _lib.source_digest()
    except Exception:
        pass
if pm: json.dump(pm, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
if tb and not os.environ.get("PROFILE_ENV"):
    # where the step's HBM bytes go: every kernel's PMC bytes per step, largest first (profiles/<tag>_traffic.md; the text around the table is
    # written by hand after reading it)
    T = [f"# HBM traffic per kernel — {tag} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024)", "",
         f"Step total {tb/1e9/PMC_STEPS:.1f} GB (algorithmic 56.2 GB: 1.58 GB/img x 32 + 5.6 GB of weight-side traffic).", "",
         "| GB/step | % | launches/step | MB/launch | read MB | written MB | kernel |", "|---|---|---|---|---|---|---|"]
    per = []
    for k in f:
        if probe(k) or not f[k][0]: continue
        rd, wr = 2 * f[k][1] * 1024, (w[k][1] if k in w else 0.0) * 1024
        per.append((rd + wr, f[k][0], rd, wr, k))
    for b, n, rd, wr, k in sorted(per, reverse=True)[:45]:
        T.append(f"| {b/1e9/PMC_STEPS:.2f} | {100*b/tb:.1f} | {n/PMC_STEPS:.1f} | {b/n/1e6:.1f} | {rd/n/1e6:.1f} | {wr/n/1e6:.1f} | `{clean(k)[:100]}` |")
    open(os.path.join(out, f"{tag}_traffic_table.md"), "w").write("\n".join(T) + "\n")
open(os.path.join(out, f"{tag}_summary.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L[:16]))
