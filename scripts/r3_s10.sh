#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s10; mkdir -p $O
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$O/$name.json") if l.startswith("{")][-1]; print("$name", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("achieved"), (d.get("roofline") or {}).get("avg_launch_us"))
except Exception as e: print("$name failed", e, open("$O/$name.err").read()[-1200:])
PY
}
A="--steps 60 --warmup 6 --no-cpu-baseline"
for i in 1 2 3; do
b base_$i python bench.py $A
b w80_$i env CROG_WGRAD256=80 python bench.py $A
b w96_$i env CROG_WGRAD256=96 python bench.py $A
b w112_$i env CROG_WGRAD256=112 python bench.py $A
b w144_$i env CROG_WGRAD256=144 python bench.py $A
done
