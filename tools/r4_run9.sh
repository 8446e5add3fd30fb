#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4i
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -x -q -k "not autoregressive" > $O/t1.log 2>&1; echo "test rc=$?"; tail -4 $O/t1.log
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_resident.py 32 200 > $O/stamps.txt 2>&1; echo "stamps rc=$?"; grep -A8 "persistent attention, row 0" $O/stamps.txt; grep -A3 "attention rows" $O/stamps.txt
for L in 128 190; do for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --tokens $L --no-extra --no-cpu-baseline > $O/bench_L${L}_$rep.log 2>&1
  echo "L=$L rep $rep: rc=$? $(grep -o '"decoder_step_us": [0-9.]*' $O/bench_L${L}_$rep.log) $(grep -o '"value": [0-9.]*' $O/bench_L${L}_$rep.log)"
done; done
