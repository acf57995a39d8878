#!/bin/bash
# Round 6: the 1x1 / linear legs with their per-shape tables, then the XCD-masked weight-gradient stream A/B (item 4).
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r6; mkdir -p $o gpurun_out/t6
CROG_BENCH_LAUNCH_TABLE=$o/linfwd_table_insitu.txt python bench.py --roofline-kernel lin_fwd --no-cpu-baseline > $o/bench_linfwd.json 2> $o/bench_linfwd.err
CROG_SINGLE_STREAM=1 CROG_BENCH_LAUNCH_TABLE=$o/linfwd_table_serial.txt python bench.py --roofline-kernel lin_fwd --no-cpu-baseline > $o/bench_linfwd_serial.json 2> $o/bench_linfwd_serial.err
CROG_BENCH_LAUNCH_TABLE=$o/linwgrad_table_insitu.txt python bench.py --roofline-kernel lin_wgrad --no-cpu-baseline > $o/bench_linwgrad.json 2> $o/bench_linwgrad.err
tail -c 400 $o/bench_linfwd.json; echo; head -12 $o/linfwd_table_serial.txt
BENCH_ARGS="--steps 30 --warmup 5" AB_PASSES=2 bash scripts/ab_env.sh $o/ab_xcd.txt "-" "CROG_WGRAD_XCDS=3" "CROG_WGRAD_XCDS=4" "CROG_WGRAD_XCDS=5" "CROG_WGRAD_XCDS=6" "-"
