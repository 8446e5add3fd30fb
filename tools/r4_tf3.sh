#!/bin/bash
# round 4: A/B of the bench line (three runs of 20 steps) + the resident-loop parity tests
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r4tf3
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "resident_decoder or resident_loop" > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -2 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_$i.json 2>/dev/null
  echo "bench $i: rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.json) $(grep -o '"decoder_step_us": [0-9.]*' $O/bench_$i.json) $(grep -o '"avg_launch_us": [0-9.]*' $O/bench_$i.json)"
done
