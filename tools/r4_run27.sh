#!/bin/bash
# round 4: location convolution on 16 waves (16 x 16 x 4 MFMA tiles) - parity, stamps at batch 1, timings
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4aa
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "fixture or resident_decoder_equals or autoregressive_resident_loop or long_rows or oracle" > $O/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/t.log
[ $rc -ne 0 ] && exit $rc
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_resident.py 1 200 > $O/stamps_tf_b1.txt 2>&1; echo "rc=$?"; grep -A6 "persistent attention" $O/stamps_tf_b1.txt
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_resident.py 32 200 > $O/stamps_tf_b32.txt 2>&1; echo "rc=$?"; grep -A6 "persistent attention" $O/stamps_tf_b32.txt
for i in 1 2; do
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --only-extra c1_b1x568,ar_b1_1000 > $O/bench_$i.log 2>&1; echo "run $i rc=$? $(grep -o '"ar_b[0-9]*_1000": {"us_per_step": [0-9.]*' $O/bench_$i.log | tr '\n' ' ') $(grep -o '"c1_b1x568": {"ms_per_utterance": [0-9.]*' $O/bench_$i.log) $(grep -o '"value": [0-9.]*' $O/bench_$i.log) $(grep -o '"decoder_loop": [0-9.]*' $O/bench_$i.log | head -1)"
done
