#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s8; mkdir -p $O
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$O/$name.json") if l.startswith("{")][-1]; print("$name", d["value"], d["ms_per_step"], d.get("step_issue",{}).get("mode"), (d.get("roofline") or {}).get("achieved"), (d.get("roofline") or {}).get("avg_launch_us"))
except Exception as e: print("$name failed", e, open("$O/$name.err").read()[-1200:])
PY
}
A="--steps 40 --warmup 6 --no-cpu-baseline"
for i in 1 2; do
b base_$i python bench.py $A
b big_$i env CROG_DEFER_WGRAD=big python bench.py $A
done
