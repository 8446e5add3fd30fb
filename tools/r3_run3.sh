#!/bin/bash
set -u
# GPU run 3 of round 3: autoregressive loop beside the resident attention kernel - parity, A/B, kernel stats
set -o pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_run3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_synthesizer_gpu.py -q -x -k "autoregressive or config3 or synthesizer or graph_replay or rebinding" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for i in 1 2; do
  GVX_AR_RESIDENT=0 timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra ar_b64_1000,ar_b1_1000 > $O/bench_ar_perstep_$i.json 2>/dev/null || exit 1
  timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra ar_b64_1000,ar_b1_1000 > $O/bench_ar_resident_$i.json 2>/dev/null || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_run3/bench_ar_*.json")):
    e=json.load(open(f))["extra"]
    print(f.split("/")[-1], "b64", e["ar_b64_1000"]["us_per_step"], "b1", e["ar_b1_1000"]["us_per_step"])
PY
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_run3/prof_ar1 -- python3 $R/tools/run_config.py ar 1 > $R/$O/prof_ar1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_run3/prof_ar64 -- python3 $R/tools/run_config.py ar 64 > $R/$O/prof_ar64.log 2>&1 || exit 1
cd $R
python tools/kstats.py $O/prof_ar1 > $O/kstats_ar1.txt 2>&1
python tools/kstats.py $O/prof_ar64 > $O/kstats_ar64.txt 2>&1
head -12 $O/kstats_ar1.txt
