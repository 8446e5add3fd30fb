#!/bin/bash
# round 4: autoregressive resident loop - parity, stamps, timing
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4n
mkdir -p $O
timeout -k 10 420 python -m pytest tests/test_parity_gpu.py -x -q -k "autoregressive_resident_loop" > $O/t_ar.log 2>&1; rc=$?; echo "ar tests rc=$rc"; tail -5 $O/t_ar.log
if [ $rc -ne 0 ]; then exit $rc; fi
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_ar.py 1 128 > $O/stamps_b1.txt 2>&1; echo "rc=$?"
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_ar.py 32 128 > $O/stamps_b32.txt 2>&1; echo "rc=$?"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --only-extra ar_b1_1000 > $O/bench_ar.log 2>&1; echo "bench rc=$?"; echo "$(grep -o '"ar_b[0-9]*_1000": {"us_per_step": [0-9.]*' $O/bench_ar.log)"
head -30 $O/stamps_b1.txt
