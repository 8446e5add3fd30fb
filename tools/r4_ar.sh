#!/bin/bash
# round 4: the autoregressive resident kernel with its own arguments read from the kernel-argument segment - parity tests, bench line with extras
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r4ar
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "autoregressive" > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$?"
  python - <<P
import json
d=json.load(open("$O/bench_$i.json"))
x=d["extra"]
print(d["ms_per_step"], {k: x[k].get("us_per_step") for k in ("ar_b64_1000","ar_b1_1000","ar_b1_1000_L190")})
P
done
