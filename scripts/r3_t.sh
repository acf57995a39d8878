#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s16; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; grep -E "passed|failed|FAILED" $O/t_all.log | tail -5
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$O/$name.json") if l.startswith("{")][-1]; print("$name", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("achieved"), d["last_step"]["loss"])
except Exception as e: print("$name failed", e, open("$O/$name.err").read()[-800:])
PY
}
A="--steps 60 --warmup 6 --no-cpu-baseline"
for i in 1 2 3; do b t_$i python bench.py $A; done
b t_b8 python bench.py $A --batch 8 --roofline-kernel none
