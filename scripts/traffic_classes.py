"""HBM bytes of a profiled run by kernel CLASS (every kernel, not the top 45 of <tag>_traffic_table.md).
usage: python scripts/traffic_classes.py <fetch_dir> <write_dir> <pmc_steps>     (the two rocprofv3 --pmc output directories of
scripts/profile_round.sh; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, MI355X_MICROARCH.md section HBM).  The classes follow kernel names:
which GEMM instantiation serves which layer class is approximate for the generic gemm_dma kernels (they serve 1x1 and a few 3x3 launches)."""
import collections
import csv
import glob
import os
import re
import sys


def agg(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    fs = glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv"))
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]][0] += 1
            acc[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return acc


def gemm_a_layout(n):
    """A-operand layout of a GEMM instantiation from its (sometimes mis-demangled) name, as scripts/summarize_profile.py reads it:
    0 = rows of K-contiguous activations (1x1 / linear forward, data gradients), 1 = implicit 3x3 im2col, 2 = dy^T (weight gradients)."""
    m = re.search(r"gemm_dma_kernelIDF16bLi(\d)ELi(\d)E", n)
    if m:
        return int(m.group(1))
    if re.search(r"gemm_dma_kernel<bool _Accum, int, E, (\d), ", n):
        return 1
    if "gemm_dma_kernel<bool _Accum, int, EL, int, E," in n:
        return 2
    for pat in (r"gemm_dma16_kernel(?:<|ILi)(\d)", r"gemm_pp_kernel(?:<|ILi)(\d)"):
        m = re.search(pat, n)
        if m:
            return int(m.group(1))
    return None


def cls(n):
    if "probe_kernel" in n or "rocclr_copyBuffer" in n:
        return None
    if "bn_apply" in n or "bn_fwd" in n:
        return "BatchNorm forward apply"
    if "bn_bwd_apply" in n:
        return "BatchNorm backward apply"
    if "bn_bwd_partial" in n or "bn_" in n:
        return "BatchNorm backward first pass / other BN"
    if "ln_" in n:
        return "LayerNorm"
    if "adam" in n:
        return "Adam"
    if "zero_f32" in n or "FillFunctor" in n or "fill" in n.lower():
        return "zeroing (gradient buffer, statistics rows)"
    if "ppt" in n or "wgrad" in n or "splitk_reduce" in n or "reduce_split" in n or "colsum" in n:
        return "weight gradients (+ their reductions)"
    if "flash" in n or "softmax" in n:
        return "attention"
    al = gemm_a_layout(n)
    if al == 2:
        return "weight gradients (+ their reductions)"
    if al == 1 or "conv_sw" in n:
        return "3x3 conv forward + data gradient"
    if "conv3_dgrad_weights" in n or "weights_t" in n or "transpose" in n:
        return "weight layout refresh"
    if "gemm" in n:
        return "1x1 / linear forward + data gradient"
    if "pool" in n or "upsample" in n:
        return "pooling / upsampling"
    return "rest (head, losses, embedding, elementwise)"


def main():
    fd, wd, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    f, w = agg(fd, "FETCH_SIZE"), agg(wd, "WRITE_SIZE")
    t = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for k in f:
        c = cls(k)
        if c is None:
            continue
        t[c][0] += 2 * f[k][1] * 1024
        t[c][1] += (w[k][1] if k in w else 0.0) * 1024
        t[c][2] += f[k][0]
    tot = sum(v[0] + v[1] for v in t.values())
    print(f"{tot / 1e9 / steps:6.2f} GB per step over {steps} steps")
    print("| class | GB/step | % | read | written | launches/step |\n|---|---|---|---|---|---|")
    for c, v in sorted(t.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
        print(f"| {c} | {(v[0] + v[1]) / 1e9 / steps:.2f} | {100 * (v[0] + v[1]) / tot:.1f} | {v[0] / 1e9 / steps:.2f} | {v[1] / 1e9 / steps:.2f} | {v[2] / steps:.0f} |")


if __name__ == "__main__":
    main()
