#!/bin/bash
set -u
# GPU run 4 of round 3: rows of up to 256 positions beside the split resident kernel
set -o pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_run4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -q -x -k "persistent or handoff or depth6" > $O/tests_pa.log 2>&1 || { tail -40 $O/tests_pa.log; exit 1; }
tail -2 $O/tests_pa.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 10 > $O/bench_L128_$i.json 2>/dev/null || exit 1
  timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 10 --tokens 190 > $O/bench_L190_pa_$i.json 2>$O/bench_L190_pa_$i.err || { tail -5 $O/bench_L190_pa_$i.err; exit 1; }
  GVX_ATTN_PERSISTENT=0 timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 10 --tokens 190 > $O/bench_L190_perstep_$i.json 2>/dev/null || exit 1
  timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 10 --tokens 256 > $O/bench_L256_pa_$i.json 2>/dev/null || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_run4/bench_*.json")):
    d=json.load(open(f)); r=d["roofline"]
    print(f.split("/")[-1], d["ms_per_step"], "launch", r["avg_launch_us"], "step", r["decoder_step_us"], d["stage_ms"])
PY
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests_all.log 2>&1 || { tail -40 $O/tests_all.log; exit 1; }
tail -2 $O/tests_all.log
