#!/bin/bash
set -u
# HBM traffic of the training step's kernels: FETCH_SIZE / WRITE_SIZE in separate --pmc passes (counter collection serialises
# kernels, so the training forward takes the launch-per-attention-step loop here: GVX_TRAIN_RESIDENT=0)
set -o pipefail
R="$GRAFT_REPO_ROOT"; cd /tmp; export TMPDIR=/tmp GVX_TRAIN_RESIDENT=0
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $R/gpurun_out/pmc_$c -- python3 $R/tools/train_profile.py 32 128 100 > $R/gpurun_out/pmc_train_$c.log 2>&1 || { tail -5 $R/gpurun_out/pmc_train_$c.log; exit 1; }
done
cd $R && python tools/pmc_summary.py r03train 32 | head -40
