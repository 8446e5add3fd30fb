#!/bin/bash
# round 4: rows' flag waits with two looks in flight - parity subset, headline + autoregressive timings
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4u
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "autoregressive_resident_loop or resident_decoder_equals or timeout" > $O/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/t.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --only-extra ar_b1_1000,ar_b64_1000 > $O/bench_$i.log 2>&1; echo "run $i rc=$? $(grep -o '"ar_b[0-9]*_1000": {"us_per_step": [0-9.]*' $O/bench_$i.log | tr '\n' ' ') $(grep -o '"value": [0-9.]*' $O/bench_$i.log) $(grep -o '"decoder_loop": [0-9.]*' $O/bench_$i.log | head -1)"
done
