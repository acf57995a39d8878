#!/bin/bash
# round 3, GPU session 1: new tests first, then graph vs eager, then probes
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s1; mkdir -p $O
python -m pytest tests/test_graph_step_gpu.py -x -q > $O/t_graph.log 2>&1; tail -3 $O/t_graph.log
python scripts/probe_graph_events.py > $O/events.log 2>&1; tail -8 $O/events.log
for mode in "" "--eager"; do
  python bench.py --steps 30 --warmup 6 --no-cpu-baseline $mode > $O/b32$mode.json 2> $O/b32$mode.err; python - <<PY
import json
try:
    d=json.loads(open("$O/b32$mode.json").read().strip().splitlines()[-1]); print("B32 [$mode]", d["value"], d["ms_per_step"], d.get("step_issue"), d["roofline"]["achieved"] if d["roofline"] else None, d.get("measured_peaks",{}).get("hbm_copy_modes_GBps"))
except Exception as e: print("B32 [$mode] failed", e, open("$O/b32$mode.err").read()[-1500:])
PY
  python bench.py --steps 30 --warmup 6 --batch 8 --no-cpu-baseline --roofline-kernel none $mode > $O/b8$mode.json 2> $O/b8$mode.err; python - <<PY
import json
try:
    d=json.loads(open("$O/b8$mode.json").read().strip().splitlines()[-1]); print("B8 [$mode]", d["value"], d["ms_per_step"], d.get("step_issue"))
except Exception as e: print("B8 [$mode] failed", e, open("$O/b8$mode.err").read()[-1500:])
PY
done
for mode in "" "--eager"; do
  CROG_FORCE_DDP=1 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --roofline-kernel none $mode > $O/ddp$mode.json 2> $O/ddp$mode.err
  tail -c 700 $O/ddp$mode.json; tail -3 $O/ddp$mode.err
done
for cus in 64 96 128 192; do
  CROG_WGRAD_CUS=$cus python bench.py --steps 20 --warmup 5 --no-cpu-baseline --eager > $O/cu$cus.json 2> $O/cu$cus.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/cu$cus.json").read().strip().splitlines()[-1]); print("CU$cus", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_launch_us"])
except Exception as e: print("CU$cus failed", e, open("$O/cu$cus.err").read()[-800:])
PY
done
python scripts/find_copies.py > $O/copies.log 2>&1; tail -45 $O/copies.log
python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; tail -5 $O/t_all.log
