R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_now
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_now -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --roofline-kernel none > $R/gpurun_out/p_now.log 2>&1
cd $R
f=$(find gpurun_out/prof_now -name "*.db" | head -1)
python - "$f" <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
prev = collections.Counter(); nxt = collections.Counter(); streams = collections.Counter()
last = {}
for i, (n, s, e, q, gx, wx) in enumerate(rows):
    if "copyBuffer" in n:
        streams[q] += 1
        p = last.get(q)
        prev[(p or "?")[:70]] += 1
        for j in range(i + 1, min(i + 30, len(rows))):
            if rows[j][3] == q: nxt[rows[j][0][:70]] += 1; break
    last[q] = n
print("streams", streams); print("prev on same stream", prev.most_common(8)); print("next on same stream", nxt.most_common(8))
try:
    print(db.execute("select name from sqlite_master where type in ('table','view') and name like '%memory%'").fetchall())
except Exception as e: print(e)
PY
rm -f $f
