R=$PWD; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for mode in torch direct; do
rm -rf $R/gpurun_out/prof_ddp
if [ $mode = direct ]; then export CROG_SYNCBN_DIRECT=1; fi
CROG_FORCE_DDP=1 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_ddp -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --roofline-kernel none > $R/gpurun_out/p_ddp.log 2>&1
f=$(find $R/gpurun_out/prof_ddp -name "*.db" | head -1)
echo "== $mode"; grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/p_ddp.log
python3 - "$f" <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
# main stream = the one with most kernels
cnt = collections.Counter(r[3] for r in rows); main = cnt.most_common(1)[0][0]
ms = [r for r in rows if r[3] == main]
# gap after bn_bwd_partial / gemm-with-stats before bn apply on the main stream
def gaps(prev_pat, next_pat):
    g = []
    for a, b in zip(ms, ms[1:]):
        if prev_pat in a[0] and next_pat in b[0]: g.append((b[1] - a[2]) / 1e3)
    return g
for pp, nn in (("bn_bwd_partial", "bn_bwd_apply"), ("gemm_dma", "bn_apply_stats")):
    g = sorted(gaps(pp, nn))
    if g: print(f"{pp} -> {nn}: n={len(g)} median {g[len(g)//2]:.1f} us, mean {sum(g)/len(g):.1f} us, p90 {g[int(len(g)*0.9)]:.1f}")
names = collections.Counter(r[0][:60] for r in rows if "nccl" in r[0].lower() or "rccl" in r[0].lower() or "AllReduce" in r[0])
print(names.most_common(5), "streams", cnt.most_common(6))
PY
rm -f $f
done
