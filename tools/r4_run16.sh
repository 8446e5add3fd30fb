#!/bin/bash
# round 4: both resident loops - parity, stamps of the autoregressive one, timings (headline + autoregressive extras)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4p
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "autoregressive or resident_decoder_equals or timeout or two_models" > $O/t_ar.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/t_ar.log
if [ $rc -ne 0 ]; then exit $rc; fi
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_ar.py 1 128 > $O/stamps_b1.txt 2>&1; echo "rc=$?"
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_ar.py 32 128 > $O/stamps_b32.txt 2>&1; echo "rc=$?"
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --only-extra ar_b1_1000,ar_b64_1000 > $O/bench_ar.log 2>&1; echo "bench rc=$?"; echo "$(grep -o '"ar_b[0-9]*_1000": {"us_per_step": [0-9.]*' $O/bench_ar.log) $(grep -o '"value": [0-9.]*' $O/bench_ar.log)"
sed -n 5,28p $O/stamps_b1.txt
