#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s3; mkdir -p $O
python -m pytest tests/test_graph_step_gpu.py -x -q > $O/t_graph.log 2>&1; tail -15 $O/t_graph.log
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["ms_per_step"], d.get("step_issue"), (d.get("roofline") or {}).get("achieved"), (d.get("roofline") or {}).get("avg_launch_us"))
except Exception as e: print("$name failed", e, open("$O/$name.err").read()[-1200:])
PY
}
A="--steps 30 --warmup 6 --no-cpu-baseline"
b r32 python bench.py $A
b e32 python bench.py $A --eager
b r32b python bench.py $A
b h32 env CROG_STEP_GRAPH=hipgraph python bench.py $A
b r8 python bench.py $A --batch 8 --roofline-kernel none
b e8 python bench.py $A --batch 8 --roofline-kernel none --eager
b r16 python bench.py $A --batch 16 --roofline-kernel none
b e16 python bench.py $A --batch 16 --roofline-kernel none --eager
b rddp env CROG_FORCE_DDP=1 CROG_STEP_GRAPH=1 python bench.py $A --roofline-kernel none
b eddp env CROG_FORCE_DDP=1 python bench.py $A --roofline-kernel none --eager
python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; tail -5 $O/t_all.log
