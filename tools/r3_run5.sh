#!/bin/bash
set -u
# GPU run 5 of round 3: 64-row teacher-forced loop beside the resident attention kernel
set -o pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_run5; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -q -x -k "shapes_against_oracle" > $O/tests_shapes.log 2>&1 || { tail -40 $O/tests_shapes.log; exit 1; }
tail -2 $O/tests_shapes.log
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -q -x -k "north_star" > $O/tests_ns.log 2>&1 || { tail -40 $O/tests_ns.log; exit 1; }
tail -2 $O/tests_ns.log
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra tf_b64x800 --steps 5 > $O/bench_64_$i.json 2>$O/bench_64_$i.err || { tail -5 $O/bench_64_$i.err; exit 1; }
  GVX_ATTN_PERSISTENT=0 timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra tf_b64x800 --steps 5 > $O/bench_64_lanes_$i.json 2>/dev/null || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_run5/bench_64*.json")):
    d=json.load(open(f)); e=d["extra"]["tf_b64x800"]
    print(f.split("/")[-1], "headline", d["ms_per_step"], "| b64:", e["ms_per_step"], e["mel_frames_per_s"], e["decoder_step_us_per_64_rows"], e.get("stage_ms") or e.get("lane_stage_ms"), (e.get("roofline") or {}).get("avg_launch_us"))
PY
