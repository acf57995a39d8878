#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s13; mkdir -p $O
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$O/$name.json") if l.startswith("{")][-1]; print("$name", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("achieved"), d["last_step"])
except Exception as e: print("$name failed", e, open("$O/$name.err").read()[-1200:])
PY
}
A="--steps 60 --warmup 6 --no-cpu-baseline"
b new_1 python bench.py $A
b new_2 python bench.py $A
b new_e python bench.py $A --eager
b new_b8 python bench.py $A --batch 8 --roofline-kernel none
timeout 1500 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; tail -5 $O/t_all.log
