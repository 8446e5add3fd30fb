#!/bin/bash
# round 4: narrow-N GEMM on few rows as 64 x 96 tiles of six waves - parity fixtures, single-utterance and batch-1 autoregressive bench legs
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r4c1
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "fixture or oracle_and_stages or shapes_against or c1 or config1" > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -2 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --only-extra c1_b1x568,ar_b1_1000 > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$?"
  python - <<P
import json
d=json.load(open("$O/bench_$i.json"))
x=d["extra"]
print(d["ms_per_step"], x["c1_b1x568"]["ms_per_utterance"], x["ar_b1_1000"]["us_per_step"])
P
done
