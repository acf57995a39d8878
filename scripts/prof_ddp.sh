#!/bin/bash
# Where does the forced-DDP step (CROG_FORCE_DDP=1: DistributedDataParallel + SyncBatchNorm at world size 1, every exchange executed) spend its
# extra 2.4 ms?  One kernel trace of the default step and one of the forced-DDP step, same box, per-queue / per-family / gap tables of both.
# usage (GPU box, repo root): bash scripts/prof_ddp.sh
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_ddp; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/plain -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/plain.log 2>&1
export CROG_FORCE_DDP=1
rocprofv3 --kernel-trace --output-format csv -d $out/ddp -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/ddp.log 2>&1
unset CROG_FORCE_DDP
for m in plain ddp; do
  f=$(find $out/$m -name "*kernel_trace.csv" | head -1)
  python3 scripts/chain_breakdown.py $f > $out/${m}_chains.txt 2>&1
  python3 scripts/chain_gaps.py $f > $out/${m}_gaps.txt 2>&1
  python3 scripts/family_breakdown.py $f 3 > $out/${m}_families.txt 2>&1
  python3 scripts/by_grid.py $f 7 200 --last 3 > $out/${m}_by_grid.txt 2>&1
  gzip -c $f > $out/${m}_trace.csv.gz
done
find $out -name "*kernel_trace.csv" -delete; find $out -name "*.csv" -size +5M -delete
tail -c 600 $out/plain.log; echo; tail -c 600 $out/ddp.log; echo; head -3 $out/plain_chains.txt $out/ddp_chains.txt
