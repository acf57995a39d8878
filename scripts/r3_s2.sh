#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s2; mkdir -p $O
python -m pytest tests/test_graph_step_gpu.py -x -q > $O/t_graph.log 2>&1; tail -3 $O/t_graph.log
b() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["ms_per_step"], d.get("step_issue",{}).get("mode"))
except Exception as e: print("$name failed", e, open("$O/$name.err").read()[-600:])
PY
}
A="--steps 20 --warmup 6 --no-cpu-baseline --roofline-kernel none"
b g_def python bench.py $A
b g_q1 env DEBUG_HIP_FORCE_GRAPH_QUEUES=1 python bench.py $A
b g_q2 env DEBUG_HIP_FORCE_GRAPH_QUEUES=2 python bench.py $A
b g_q8 env DEBUG_HIP_FORCE_GRAPH_QUEUES=8 python bench.py $A
b g_pc0 env DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python bench.py $A
b g_pc1 env DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python bench.py $A
b g_1s env CROG_SINGLE_STREAM=1 python bench.py $A
b e_1s env CROG_SINGLE_STREAM=1 python bench.py $A --eager
b g_b8_1s env CROG_SINGLE_STREAM=1 python bench.py $A --batch 8
b e_b8_1s env CROG_SINGLE_STREAM=1 python bench.py $A --batch 8 --eager
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $O/tr_g -- python3 bench.py --steps 4 --warmup 5 --no-cpu-baseline --roofline-kernel none > $O/tr_g.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr_e -- python3 bench.py --steps 4 --warmup 5 --no-cpu-baseline --roofline-kernel none --eager > $O/tr_e.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr_g8 -- python3 bench.py --steps 4 --warmup 5 --no-cpu-baseline --roofline-kernel none --batch 8 > $O/tr_g8.log 2>&1
for t in tr_g tr_e tr_g8; do f=$(find $O/$t -name "*kernel_trace.csv" | head -1); echo "== $t $f"; python scripts/trace_overlap.py $f; done
find $O -name "*kernel_trace.csv" -size +20M -delete
