#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4d
mkdir -p $O
export GVX_TF_RESIDENT=1
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_resident.py 32 200 > $O/stamps.txt 2>&1; echo "stamps rc=$?"; grep -v "^   wave\|events:" $O/stamps.txt
for dbg in 0 8 16 32 48 24 56; do
  for rep in 1 2; do
  GVX_RS_DEBUG=$dbg timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_dbg${dbg}_$rep.log 2>&1
  echo "debug $dbg rep $rep: rc=$? $(grep -o '"decoder_step_us": [0-9.]*' $O/bench_dbg${dbg}_$rep.log)"
  done
done
