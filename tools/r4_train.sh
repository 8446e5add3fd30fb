#!/bin/bash
# round 4: the training forward on the resident decoder kernel (tape in its cell epilogues) vs the launch per step
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r4train
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_training_gpu.py -x -q -m gpu > $O/train_tests.log 2>&1; rc=$?; echo "training tests rc=$rc"; tail -3 $O/train_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/train_bench.py > $O/train_bench_resident.txt 2>&1; echo "resident rc=$?"; grep train_step $O/train_bench_resident.txt
GVX_TRAIN_RESIDENT_LOOP=0 timeout -k 10 300 python tools/train_bench.py > $O/train_bench_launches.txt 2>&1; echo "launch per step rc=$?"; grep train_step $O/train_bench_launches.txt
