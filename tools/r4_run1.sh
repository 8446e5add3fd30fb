#!/bin/bash
# round 4, first run of the resident decoder kernel: bit-equality test on the smallest shapes first, then the bench with / without it
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4a
mkdir -p $O
timeout -k 10 240 python -m pytest tests/test_parity_gpu.py -x -q -k "resident_decoder_equals" > $O/t1.log 2>&1
rc=$?; echo "test rc=$rc"; tail -15 $O/t1.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_res.log 2>&1; echo "bench resident rc=$?"; tail -2 $O/bench_res.log
GVX_TF_RESIDENT=0 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_step.log 2>&1; echo "bench per-step rc=$?"; tail -2 $O/bench_step.log
