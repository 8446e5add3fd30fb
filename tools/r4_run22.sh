#!/bin/bash
# round 4: autoregressive resident loop for rows of 129-256 tokens - parity, timing at batch 1, L = 190
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4v
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "autoregressive_resident_loop" > $O/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/t.log
[ $rc -ne 0 ] && exit $rc
for v in 1 0; do
GVX_AR_RESIDENT_LOOP=$v timeout -k 10 400 python bench.py --steps 5 --warmup 2 --tokens 190 --no-cpu-baseline --only-extra ar_b1_1000 > $O/bench_L190_$v.log 2>&1; echo "L=190 loop=$v rc=$? $(grep -o '"ar_b[0-9]*_1000": {"us_per_step": [0-9.]*' $O/bench_L190_$v.log | tr '\n' ' ') $(grep -o '"value": [0-9.]*' $O/bench_L190_$v.log)"
done
