"""Which FOREIGN gfx950 kernels that run beside this library's MFMA kernels carry packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 /
v_pk_add_f32)?  (the library itself is built without them - crog_amd/_lib.py NO_PACKED_F32 -, RCCL's reduction kernels and
ATen's element-wise kernels are not, and the gradient all-reduce is overlapped with backward by design.)
usage (build container, no GPU): python scripts/count_pk_foreign.py /path/to/lib.so [substring of the kernel names to report ...]
The .hip_fatbin section is dumped (llvm-objcopy), the gfx950 code object unbundled (clang-offload-bundler), disassembled as a stream
(llvm-objdump -d) and counted per kernel symbol; nothing of the disassembly is kept."""
import os, re, subprocess, sys, tempfile, collections
LLVM = "/opt/rocm/lib/llvm/bin"
lib, pats = sys.argv[1], sys.argv[2:]
PK = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")
MFMA = re.compile(r"\bv_mfma_")
pk, ins = collections.Counter(), collections.Counter()
with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
    fat = os.path.join(td, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, "/dev/null"], check=True, capture_output=True)
    data = open(fat, "rb").read()
    os.remove(fat)
    # one bundle per translation unit, compressed ("CCOB") or plain; a library like libtorch_hip.so holds hundreds back to back
    starts = sorted(m.start() for m in re.finditer(rb"CCOB|__CLANG_OFFLOAD_BUNDLE__", data) if m.start() % 8 == 0)
    print(f"{lib}: {len(starts)} offload bundles in .hip_fatbin ({len(data) >> 20} MiB)", flush=True)
    done = 0
    for i, o in enumerate(starts):
        piece, co = os.path.join(td, "piece.bin"), os.path.join(td, "gfx950.co")
        end = starts[i + 1] if i + 1 < len(starts) else len(data)
        if data[o:o + 4] == b"CCOB":      # compressed bundle: its header holds the exact size (the section pads each bundle; the decompressor checks)
            ver = int.from_bytes(data[o + 4:o + 6], "little")
            size = int.from_bytes(data[o + 8:o + 12], "little") if ver == 2 else int.from_bytes(data[o + 8:o + 16], "little")
            end = min(end, o + size) if size > 0 else end
        open(piece, "wb").write(data[o:end])
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={piece}", f"--output={co}", "--unbundle"],
                           capture_output=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        done += 1
        p = subprocess.Popen([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "-C", co], stdout=subprocess.PIPE, text=True, errors="replace")
        cur = None
        for line in p.stdout:
            if line and line[0] == "0" and line.rstrip().endswith(">:"):
                cur = line.split("<", 1)[1].rsplit(">:", 1)[0]
                continue
            if cur is None:
                continue
            ins[cur] += 1
            if PK.search(line):
                pk[cur] += 1
        p.wait()
        os.remove(co)
    print(f"{done} gfx950 code objects disassembled", flush=True)
names = [k for k in ins if not pats or any(s in k for s in pats)]
with_pk = [k for k in names if pk[k]]
print(f"{len(ins)} kernels / functions disassembled; {len(names)} match {pats or 'everything'}; {len(with_pk)} of those hold packed-fp32 instructions "
      f"({sum(pk[k] for k in with_pk)} instructions)")
for k in sorted(with_pk, key=lambda k: -pk[k])[:40]:
    print(f"  {pk[k]:6d} of {ins[k]:7d} instructions  {k[:160]}")
