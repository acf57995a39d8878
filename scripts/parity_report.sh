#!/bin/bash
# The numbers the GPU parity tests print, from ONE run of the whole GPU suite on the tree as it stands (`pytest -q` hides them).
# usage (GPU box, repo root): bash scripts/parity_report.sh <tag>   -> gpurun_out/parity_<tag>/{parity.txt, gpu_tests.log}; copy parity.txt to profiles/<tag>_parity.txt
tag=${1:-r06}; o=gpurun_out/parity_$tag; rm -rf $o; mkdir -p $o
python -m pytest tests/ -m gpu -q -s -p no:cacheprovider --durations=12 > $o/gpu_tests.log 2>&1
rc=$?
{
  echo "# Parity numbers printed by the GPU tests (pytest -m gpu -q -s), tree $(cat .git_head 2>/dev/null), $(date -u +%Y-%m-%dT%H:%MZ); exit code $rc"
  grep -v "^\[Gloo\]\|^/opt/amdgpu\|Warning\|warnings.warn\|^$\|amdgpu.ids\|^\[W[0-9]" $o/gpu_tests.log | grep -E "[0-9]e[-+][0-9]|passed|failed|max err|cosine|img/s|bit-identical|exchanges|ranks" | sed -e 's/^\.*//' | cut -c1-600
} > $o/parity.txt
tail -5 $o/gpu_tests.log
