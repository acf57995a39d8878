#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/s4; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; tail -6 $O/t_all.log
python -m pytest tests/test_fulldepth_gpu.py -x -q -m gpu -s -k "config1 or bf16_training" > $O/t_full.log 2>&1; grep -E "distance to|HIP bf16|max \|dlogit|passed|failed" $O/t_full.log | head -60
CROG_SINGLE_STREAM=1 python scripts/profile_gemms.py 32 > $O/gemms_serial.log 2>&1; head -64 $O/gemms_serial.log
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_r -- python3 bench.py --steps 6 --warmup 5 --no-cpu-baseline --roofline-kernel none > $O/tr_r.log 2>&1
f=$(find $O/tr_r -name "*kernel_trace.csv" | head -1); python scripts/trace_overlap.py $f
f=$(find $O/tr_r -name "*kernel_stats.csv" | head -1); head -45 $f | cut -c1-180
find $O -name "*kernel_trace.csv" -size +20M -delete
