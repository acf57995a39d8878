#!/bin/bash
# Build the in-tree library (and fail loudly) before shipping the tree to the GPU box: scripts/gpu.sh TIMEOUT 'command'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" | tail -1
python -m pytest tests/test_abi_host.py -q -x -m "not gpu" 2>&1 | tail -1
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
