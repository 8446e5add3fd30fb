#!/bin/bash
# round 4: first runs of the autoregressive resident loop
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4k
mkdir -p $O
timeout -k 10 420 python -m pytest tests/test_parity_gpu.py -x -q -k "autoregressive_resident_loop" > $O/t_ar.log 2>&1; rc=$?; echo "ar tests rc=$rc"; tail -15 $O/t_ar.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 420 python -m pytest tests/test_parity_gpu.py -x -q -k "autoregressive or resident_decoder_equals" > $O/t_ar2.log 2>&1; rc=$?; echo "ar+tf tests rc=$rc"; tail -5 $O/t_ar2.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --only-extra ar_b1_1000,ar_b64_1000 > $O/bench_ar.log 2>&1; echo "bench rc=$?"; grep -o '"ar_b[0-9]*_1000": {"us_per_step": [0-9.]*' $O/bench_ar.log; grep -o '"value": [0-9.]*' $O/bench_ar.log
