#!/bin/bash
# round 4: one resident decoder kernel per deal (224 / 192 workgroups) - parity of the resident loops, bench line, training bench, stamps
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r4split
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_training_gpu.py -x -q -m gpu -k "resident or timeout or long_rows or fixture or training_forward or whole_backward" > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_$i.json 2>/dev/null
  echo "bench $i: rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.json) $(grep -o '"decoder_step_us": [0-9.]*' $O/bench_$i.json) $(grep -o '"stage_ms": {[^}]*}' $O/bench_$i.json)"
done
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --tokens 190 --no-extra --no-cpu-baseline > $O/bench_L190.json 2>/dev/null
echo "L=190: rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_L190.json) $(grep -o '"decoder_step_us": [0-9.]*' $O/bench_L190.json)"
timeout -k 10 300 python tools/train_bench.py > $O/train_bench.txt 2>&1; echo "train bench rc=$?"; grep train_step $O/train_bench.txt
bash tools/r4_stamps.sh
