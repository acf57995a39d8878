#!/bin/bash
# In-situ cost of the decoder's cross-attention, unfused (S GEMM, softmax, PV GEMM + five backward launches) against the fused kernels with
# the key padding mask (GPU box, from the repo root): one rocprofv3 kernel trace per setting, the step's launches and kernel time, the
# attention kernels by grid; then an ABBA bench.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/cross_prof; rm -rf $out; mkdir -p $out
for k in 0 1; do
  CROG_FLASH_CROSS=$k rocprofv3 --kernel-trace --output-format csv -d $out/k$k -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/k$k.log 2>&1 || exit 1
  f=$(find $out/k$k -name "*kernel_trace.csv" | head -1)
  echo "== CROG_FLASH_CROSS=$k" >> $out/cross.txt
  python3 scripts/by_grid.py $f 7 400 --last 3 | grep -i "flash\|softmax\|^total\|y= 256" >> $out/cross.txt
  python3 scripts/chain_breakdown.py $f | head -3 >> $out/cross.txt
  rm -rf $out/k$k
done
cat $out/cross.txt
AB_PASSES=2 BENCH_ARGS="--steps 40 --warmup 8" bash scripts/ab_env.sh $out/abba.txt "-" "CROG_FLASH_CROSS=0" "CROG_FLASH_CROSS=0" "-"
