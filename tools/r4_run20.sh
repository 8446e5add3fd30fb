#!/bin/bash
# round 4: the whole GPU suite on the tree with the vector-ALU mode, then rows polling without sleeps (autoregressive batch 1 / 32)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4t
mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -4 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for dbg in 0 32; do
GVX_RS_DEBUG=$dbg timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --only-extra ar_b1_1000,ar_b64_1000 > $O/bench_$dbg.log 2>&1; echo "dbg=$dbg rc=$? $(grep -o '"ar_b[0-9]*_1000": {"us_per_step": [0-9.]*' $O/bench_$dbg.log | tr '\n' ' ') $(grep -o '"value": [0-9.]*' $O/bench_$dbg.log)"
done
