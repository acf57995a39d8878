"""Code-object resource table of every kernel in the library: python scripts/kernel_resources.py [out.md]
Compiles each csrc/*.hip for gfx950 with -S (device only) and reads the .amdhsa metadata: VGPRs, AGPRs, SGPRs, LDS, scratch, spills."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crog_amd._lib import SOURCES, CSRC, HIPCC_FLAGS
rows = []
for src in SOURCES:
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["hipcc", *[x for x in HIPCC_FLAGS if x != "-fPIC"], "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", f.name],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(f.name).read()
    for blk in re.split(r"\n  - \.agpr_count:", text)[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = g("name")
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        m = re.match(r"\s*(\d+)", blk)
        rows.append((src, name, g("vgpr_count"), m.group(1) if m else "?", g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"),
                     g("vgpr_spill_count"), g("sgpr_spill_count")))
def short(n):
    m = re.search(r"gemm_(dma_)?kernelI(DF16b|f)Li(\d)ELi(\d)E(Lb(\d)E)?NS_5ShapeILi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)E(?:Li(\d+)E)?EE(Li(\d)E)?", n)
    if m:
        return "gemm_%skernel<%s, A%s, B%s, %sShape<%s,%s,%s,%s%s>%s>" % (m.group(1) or "", "bf16" if m.group(2) == "DF16b" else "f32", m.group(3), m.group(4),
            ("hwtr%s, " % m.group(6)) if m.group(6) else "", m.group(7), m.group(8), m.group(9), m.group(10), ",lean" if m.group(11) == "1" else "", (", asum%s" % m.group(14)) if m.group(14) else "")
    m = re.search(r"_ZN12_GLOBAL__N_1\d+([a-zA-Z0-9_]+?)I", n)
    return m.group(1) if m else n[:80]
out = ["| source | kernel | VGPR | AGPR | SGPR | static LDS B | scratch B | VGPR spills | SGPR spills |", "|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    out.append("| %s | `%s` | %s | %s | %s | %s | %s | %s | %s |" % (r[0], short(r[1]), *r[2:]))
bad = [r for r in rows if r[6] not in ("0", "?") or r[7] not in ("0", "?")]
head = ["# Kernel resources (gfx950 code-object metadata)", "",
        "`python scripts/kernel_resources.py` — hipcc -O3 --offload-arch=gfx950 -S of every source in `crog_amd/csrc/`, fields from the `.amdhsa` kernel metadata.",
        f"{len(rows)} kernels; **{len(bad)} use scratch memory or spill** (listed first).", ""]
if bad:
    head += ["Kernels with scratch / spills:"] + ["- `%s` (%s): scratch %s B, %s VGPR spills" % (short(r[1]), r[0], r[6], r[7]) for r in bad] + [""]
txt = "\n".join(head + out) + "\n"
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt)
print("\n".join(head))
