#!/bin/bash
# round 4: autoregressive resident loop - parity tests, then the decode timed with and without it
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4l
mkdir -p $O
timeout -k 10 420 python -m pytest tests/test_parity_gpu.py -x -q -k "autoregressive_resident_loop" > $O/t_ar.log 2>&1; rc=$?; echo "ar tests rc=$rc"; tail -5 $O/t_ar.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in 1 0; do
GVX_AR_RESIDENT_LOOP=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --only-extra ar_b1_1000 > $O/bench_ar_$v.log 2>&1; echo "bench rc=$?"; echo "loop=$v $(grep -o '"ar_b[0-9]*_1000": {"us_per_step": [0-9.]*' $O/bench_ar_$v.log)"
done
