#!/bin/bash
# A/B of environment switches on the default bench line (GPU box): scripts/ab_env.sh OUTFILE "VAR=val VAR2=val" "VAR=val" ...
# each argument is one configuration ("-" = defaults); prints ms_per_step per configuration, two passes.
out=$1; shift
: > $out
for pass in $(seq 1 ${AB_PASSES:-2}); do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    r=$(env $e python bench.py --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "pass $pass [$cfg] $r" >> $out
  done
done
cat $out
