#!/bin/bash
# round 4: teacher-forced rows of 129-256 tokens, <= 16 rows, on the 224-workgroup deal - parity, timing against the 192-workgroup deal
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4w
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -x -q -k "resident_decoder_equals or long_rows" > $O/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/t.log
[ $rc -ne 0 ] && exit $rc
for v in 1 0; do
for B in 1 16; do
GVX_TF_LONG_224=$v timeout -k 10 400 python bench.py --steps 8 --warmup 3 --tokens 190 --batch $B --no-cpu-baseline --no-extra > $O/bench_L190_B${B}_$v.log 2>&1; echo "L=190 B=$B deal224=$v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_L190_B${B}_$v.log) $(grep -o '"decoder_loop": [0-9.]*' $O/bench_L190_B${B}_$v.log | head -1)"
done
done
