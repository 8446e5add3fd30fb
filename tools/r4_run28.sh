#!/bin/bash
# round 4: autoregressive loop - parity, stamps, timings
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4ab
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "autoregressive or fixture or memory_traffic" > $O/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/t.log
[ $rc -ne 0 ] && exit $rc
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_ar.py 32 128 > $O/stamps_b32.txt 2>&1; echo "rc=$?"; sed -n 5,26p $O/stamps_b32.txt
for i in 1 2; do
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --only-extra ar_b1_1000,ar_b64_1000,ar_b1_1000_L190 > $O/bench_$i.log 2>&1; echo "run $i rc=$? $(grep -o '"ar_b[0-9_L]*": {"us_per_step": [0-9.]*' $O/bench_$i.log | tr '\n' ' ') $(grep -o '"value": [0-9.]*' $O/bench_$i.log)"
done
