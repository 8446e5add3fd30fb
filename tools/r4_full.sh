#!/bin/bash
# round 4: the whole GPU suite, then the bench line as the driver runs it
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4full
mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -5 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; cat $O/bench.json
