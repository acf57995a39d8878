#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s12; mkdir -p $O
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
b l18_$i env CROG_WGRAD256=144 CROG_WGRAD256_LIN=18 python bench.py $A
b l17_$i env CROG_WGRAD256=144 CROG_WGRAD256_LIN=17 python bench.py $A
b l16_$i env CROG_WGRAD256=144 CROG_WGRAD256_LIN=16 python bench.py $A
b l18c18_$i env CROG_WGRAD256=144 CROG_WGRAD256_LIN=18 CROG_WGRAD256_CONV=18 python bench.py $A
b l17c18_$i env CROG_WGRAD256=144 CROG_WGRAD256_LIN=17 CROG_WGRAD256_CONV=18 python bench.py $A
b t112l18_$i env CROG_WGRAD256=112 CROG_WGRAD256_LIN=18 python bench.py $A
b t208l18_$i env CROG_WGRAD256=208 CROG_WGRAD256_LIN=18 python bench.py $A
done
