#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s14; mkdir -p $O
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$O/$name.json") if l.startswith("{")][-1]; print("$name", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("achieved"), d["last_step"]["loss"])
except Exception as e: print("$name failed", e, open("$O/$name.err").read()[-600:])
PY
}
A="--steps 60 --warmup 6 --no-cpu-baseline"
for i in 1 2 3; do
b conv_only_$i env CROG_WGRAD256_LIN=0 python bench.py $A
b lin18_$i env CROG_WGRAD256_LIN=18 python bench.py $A
b lin20_$i env CROG_WGRAD256_LIN=20 python bench.py $A
done
