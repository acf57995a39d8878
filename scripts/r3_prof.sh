#!/bin/bash
# round-3 profile: overlapped (default replay) and single-stream kernel stats, the two PMC passes, then the default bench line
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r03; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
PROFILE_NOTE="Default step (captured once, re-issued on three streams by crog_replay_launch): weight gradients and the text tower overlap the main chain, so per-kernel durations include what the neighbours cost." python3 scripts/summarize_profile.py r03 $out/stats $out/fetch $out/write 8 5 > $out/summarize.log 2>&1
export CROG_SINGLE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/stats1.log 2>&1
cp profiles/pmc_traffic.json $out/pmc_keep.json
PROFILE_ENV="CROG_SINGLE_STREAM=1 " PROFILE_NOTE="Single-stream run (no weight-gradient / text-tower side streams): per-kernel durations are the kernels' own." python3 scripts/summarize_profile.py r03_serial $out/stats1 $out/fetch $out/write 8 5 > $out/summarize1.log 2>&1
cp $out/pmc_keep.json profiles/pmc_traffic.json
unset CROG_SINGLE_STREAM
cp profiles/r03_summary.md profiles/r03_serial_summary.md profiles/r03_kernel_stats.csv profiles/r03_serial_kernel_stats.csv profiles/pmc_traffic.json $out/ 2>/dev/null
python3 bench.py > $out/bench.json 2> $out/bench.err
tail -c 2500 $out/bench.json
tail -3 $out/summarize.log; tail -3 $out/summarize1.log
grep -c . $out/stats.log | head -1
find $out -name "*kernel_trace.csv" -size +15M -delete; find $out -name "*counter_collection.csv" -size +15M -delete
