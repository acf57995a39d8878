#!/bin/bash
# In-situ durations of the fused attention kernels with and without the forward's dropout bit map (GPU box, from the repo root):
# one rocprofv3 kernel trace per setting, grouped by (kernel, grid) over the last 3 replayed steps; then an ABBA bench.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/keep_prof; rm -rf $out; mkdir -p $out
for k in 0 1; do
  CROG_FLASH_KEEP=$k rocprofv3 --kernel-trace --output-format csv -d $out/k$k -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/k$k.log 2>&1 || exit 1
  f=$(find $out/k$k -name "*kernel_trace.csv" | head -1)
  echo "== CROG_FLASH_KEEP=$k" >> $out/flash.txt
  python3 scripts/by_grid.py $f 7 400 --last 3 | grep -i "flash\|^total" >> $out/flash.txt
  rm -rf $out/k$k
done
cat $out/flash.txt
AB_PASSES=2 BENCH_ARGS="--steps 40 --warmup 8" bash scripts/ab_env.sh $out/abba.txt "-" "CROG_FLASH_KEEP=0" "CROG_FLASH_KEEP=0" "-"
