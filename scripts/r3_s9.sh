#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s9; mkdir -p $O
CROG_WGRAD256=128 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "production_conv3x3" > $O/t_k.log 2>&1; grep -E "wgrad splitk|passed|failed" $O/t_k.log
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$O/$name.json") if l.startswith("{")][-1]; print("$name", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("achieved"), (d.get("roofline") or {}).get("avg_launch_us"))
except Exception as e: print("$name failed", e, open("$O/$name.err").read()[-1200:])
PY
}
A="--steps 40 --warmup 6 --no-cpu-baseline"
b base_1 python bench.py $A
for t in 64 128 256; do b w$t env CROG_WGRAD256=$t python bench.py $A; done
b base_2 python bench.py $A
for t in 96 192; do b w$t env CROG_WGRAD256=$t python bench.py $A; done
b w128wg env CROG_WGRAD256=128 python bench.py $A --roofline-kernel conv3x3_wgrad
b basewg python bench.py $A --roofline-kernel conv3x3_wgrad
