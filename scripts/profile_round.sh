#!/bin/bash
# Round profile on the GPU box (from the repo root): overlapped (default replay) and single-stream kernel stats, the two PMC passes,
# then the default bench line.  usage: bash scripts/profile_round.sh <tag>; afterwards copy gpurun_out/prof_<tag>/<tag>_* (and the two
# *_trace.csv.gz as profiles/<tag>_[serial_]kernel_trace.csv.gz) into profiles/: only gpurun_out/ travels back from the GPU box
# bench.py --steps 3 --warmup 2 executes 7 steps (2 warm-ups, 1 more eager step, capture + first replay, 3 timed replays);
# --steps 2 --warmup 1 executes 6.
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
PROFILE_NOTE="Default step (captured once, re-issued on three streams by crog_replay_launch): weight gradients and the text tower overlap the main chain, so per-kernel durations include what the neighbours cost." python3 scripts/summarize_profile.py $tag $out/stats $out/fetch $out/write 7 6 > $out/summarize.log 2>&1
export CROG_SINGLE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/stats1.log 2>&1
cp profiles/pmc_traffic.json $out/pmc_keep.json
PROFILE_ENV="CROG_SINGLE_STREAM=1 " PROFILE_NOTE="Single-stream run (no weight-gradient / text-tower side streams): per-kernel durations are the kernels' own." python3 scripts/summarize_profile.py ${tag}_serial $out/stats1 $out/fetch $out/write 7 6 > $out/summarize1.log 2>&1
cp $out/pmc_keep.json profiles/pmc_traffic.json
unset CROG_SINGLE_STREAM
f=$(find $out/stats -name "*kernel_trace.csv" | head -1); python3 scripts/chain_breakdown.py $f > profiles/${tag}_chains.txt 2>&1; python3 scripts/by_grid.py $f 7 140 --last 3 > profiles/${tag}_by_grid.txt 2>&1; python3 scripts/chain_gaps.py $f > profiles/${tag}_chain_gaps.txt 2>&1
f1=$(find $out/stats1 -name "*kernel_trace.csv" | head -1); python3 scripts/by_grid.py $f1 7 140 --last 3 > profiles/${tag}_serial_by_grid.txt 2>&1
python3 scripts/family_breakdown.py $f 3 > profiles/${tag}_families.txt 2>&1; python3 scripts/family_breakdown.py $f1 3 > profiles/${tag}_serial_families.txt 2>&1
cp profiles/${tag}_families.txt profiles/${tag}_serial_families.txt profiles/${tag}_traffic_table.md profiles/${tag}_chain_gaps.txt profiles/${tag}_by_grid.txt profiles/${tag}_serial_by_grid.txt profiles/${tag}_summary.md profiles/${tag}_serial_summary.md profiles/${tag}_kernel_stats.csv profiles/${tag}_serial_kernel_stats.csv profiles/pmc_traffic.json profiles/${tag}_chains.txt $out/ 2>/dev/null
python3 bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
tail -c 1500 $out/bench.json
tail -3 $out/summarize.log; tail -3 $out/summarize1.log; head -3 profiles/${tag}_chains.txt
for t in $(find $out -name "*kernel_trace.csv"); do gzip -c $t > $out/$(basename $(dirname $(dirname $t)))_trace.csv.gz; done; find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -size +15M -delete
