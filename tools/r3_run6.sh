#!/bin/bash
set -u
set -o pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_run6; mkdir -p $O
for rot in 0 1 3 7 0 3; do
  GVX_SK_ROT=$rot timeout -k 10 300 python bench.py --no-cpu-baseline --only-extra tf_b64x800 --steps 5 > $O/bench_rot${rot}_$RANDOM.json 2>/dev/null || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_run6/bench_rot*.json")):
    d=json.load(open(f)); e=d["extra"]["tf_b64x800"]; r=d["roofline"]
    print(f.split("/")[-1], "B32:", d["ms_per_step"], "launch", r["avg_launch_us"], "step", r["decoder_step_us"], "| b64:", e["ms_per_step"], "step", e["decoder_step_us_per_64_rows"], "launch", (e.get("roofline") or {}).get("avg_launch_us"))
PY
