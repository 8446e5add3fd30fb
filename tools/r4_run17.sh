#!/bin/bash
# round 4: teacher-forced resident loop with the Prenet columns in the kernel - parity, stamps, headline
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4q
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "resident_decoder_equals or autoregressive_resident_loop" > $O/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/t.log
if [ $rc -ne 0 ]; then exit $rc; fi
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_resident.py 32 200 > $O/stamps_tf.txt 2>&1; echo "rc=$?"
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $O/bench_$i.log 2>&1; echo "bench rc=$? $(grep -o '"value": [0-9.]*' $O/bench_$i.log) $(grep -o '"stage_ms": {[^}]*}' $O/bench_$i.log | head -1)"
done
head -32 $O/stamps_tf.txt
