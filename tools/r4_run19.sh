#!/bin/bash
# round 4: vector-ALU mode of the resident kernels (B <= 2) - parity, stamps at batch 1, timings (autoregressive batch 1, single utterance)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4s
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "autoregressive or resident_decoder_equals or reference_fixture or batch1" > $O/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/t.log
if [ $rc -ne 0 ]; then exit $rc; fi
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_ar.py 1 128 > $O/stamps_b1.txt 2>&1; echo "rc=$?"
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --only-extra ar_b1_1000,c1_b1x568 > $O/bench_v.log 2>&1; echo "bench rc=$?"; echo "$(grep -o '"ar_b[0-9]*_1000": {"us_per_step": [0-9.]*' $O/bench_v.log) $(grep -o '"c1_b1x568": {"ms_per_utterance": [0-9.]*' $O/bench_v.log) $(grep -o '"value": [0-9.]*' $O/bench_v.log)"
GVX_RS_DEBUG=64 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --only-extra ar_b1_1000,c1_b1x568 > $O/bench_m.log 2>&1; echo "bench rc=$?"; echo "mfma: $(grep -o '"ar_b[0-9]*_1000": {"us_per_step": [0-9.]*' $O/bench_m.log) $(grep -o '"c1_b1x568": {"ms_per_utterance": [0-9.]*' $O/bench_m.log)"
sed -n 5,28p $O/stamps_b1.txt
