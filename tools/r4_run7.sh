#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4g
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -k "resident_decoder_equals or strict or two_models or timeout_is_loud" > $O/t1.log 2>&1; echo "test rc=$?"; tail -3 $O/t1.log
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_resident.py 32 200 > $O/stamps.txt 2>&1; echo "stamps rc=$?"; grep -v "^   wave [1-7]\|events:" $O/stamps.txt
for rep in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_$rep.log 2>&1
  echo "rep $rep: rc=$? $(grep -o '"decoder_step_us": [0-9.]*' $O/bench_$rep.log) $(grep -o '"value": [0-9.]*' $O/bench_$rep.log)"
done
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_prof.log 2>&1; echo "rocprof rc=$?"; tail -2 $O/bench_prof.log | cut -c1-400
find $O/prof -name "*kernel_stats.csv" | head -3
