#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s11; mkdir -p $O
CROG_WGRAD256=144 CROG_WGRAD256_LIN=18 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q > $O/t_k.log 2>&1; tail -3 $O/t_k.log
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$O/$name.json") if l.startswith("{")][-1]; print("$name", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("achieved"), (d.get("roofline") or {}).get("avg_launch_us"))
except Exception as e: print("$name failed", e, open("$O/$name.err").read()[-1200:])
PY
}
A="--steps 60 --warmup 6 --no-cpu-baseline"
for i in 1 2; do
b base_$i python bench.py $A
b w144_$i env CROG_WGRAD256=144 python bench.py $A
b w144l19_$i env CROG_WGRAD256=144 CROG_WGRAD256_LIN=19 python bench.py $A
b w144l18_$i env CROG_WGRAD256=144 CROG_WGRAD256_LIN=18 python bench.py $A
b w176_$i env CROG_WGRAD256=176 python bench.py $A
b w176l18_$i env CROG_WGRAD256=176 CROG_WGRAD256_LIN=18 python bench.py $A
done
