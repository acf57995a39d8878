#!/bin/bash
# Round profile on the GPU box: kernel trace + stats, then the two PMC passes (separate runs), then the default bench line.
# usage (from the repo root on the box): bash scripts/profile_round.sh <tag>
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
python3 scripts/summarize_profile.py $tag $out/stats $out/fetch $out/write 5 3 > $out/summarize.log 2>&1
cp profiles/${tag}_summary.md profiles/${tag}_kernel_stats.csv profiles/pmc_traffic.json $out/ 2>/dev/null
python3 bench.py > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json
tail -3 $out/summarize.log
