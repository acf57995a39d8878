#!/bin/bash
# In-situ durations of the attention kernels in the default replayed step (GPU box, from the repo root): one rocprofv3 kernel trace,
# grouped by (kernel, grid) over the last 3 replayed steps.  usage: bash scripts/flash_insitu.sh [tag]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
tag=${1:-now}; out=gpurun_out/flash_insitu; mkdir -p $out; rm -rf $out/t
rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/$tag.log 2>&1 || exit 1
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python3 scripts/by_grid.py $f 7 400 --last 3 | grep -i "flash\|softmax\|^total" > $out/$tag.txt
rm -rf $out/t; cat $out/$tag.txt
