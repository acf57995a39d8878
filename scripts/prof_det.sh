#!/bin/bash
# kernel-time families of the deterministic step (GPU box): bash scripts/prof_det.sh
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_det; rm -rf $out; mkdir -p $out
export CROG_DETERMINISTIC=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/stats.log 2>&1
f=$(find $out/stats -name "*kernel_trace.csv" | head -1)
python3 scripts/family_breakdown.py $f 3 > $out/det_families.txt 2>&1
python3 scripts/chain_breakdown.py $f > $out/det_chains.txt 2>&1
python3 scripts/by_grid.py $f 7 60 --last 3 > $out/det_by_grid.txt 2>&1
cat $out/det_families.txt; head -60 $out/det_chains.txt
find $out -name "*kernel_trace.csv" -delete
