#!/bin/bash
# round 4: where the resident decoder kernel's step goes (stamps) and what its engine does without dependencies
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4b
mkdir -p $O
export GVX_TF_RESIDENT=1
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_resident.py 32 200 > $O/stamps.txt 2>&1; echo "stamps rc=$?"; cat $O/stamps.txt
for dbg in 0 1 3 7 4; do
  GVX_RS_DEBUG=$dbg timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_dbg$dbg.log 2>&1
  echo "debug $dbg: rc=$? $(grep -o '"decoder_step_us": [0-9.]*' $O/bench_dbg$dbg.log) $(grep -o 'decoder_loop": [0-9.]*' $O/bench_dbg$dbg.log)"
done
