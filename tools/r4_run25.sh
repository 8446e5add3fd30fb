#!/bin/bash
# round 4: the resident loops beside copy-engine traffic
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4y
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "under_memory_traffic" > $O/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $O/t.log
