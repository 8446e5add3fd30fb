#!/bin/bash
# round 4: 64 x 800 teacher-forced as two 32-row chunks with ONE Postnet call over all rows - the north_star tests, bench line with extras
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r4b64
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "north_star or config2 or 64" > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --only-extra tf_b64x800 > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$?"
  python - <<P
import json
d=json.load(open("$O/bench_$i.json"))
x=d["extra"]["tf_b64x800"]
print(d["ms_per_step"], x["ms_per_step"], x["mel_frames_per_s"], x["stage_ms"])
P
done
