#!/bin/bash
# round 4: rocprofv3's per-kernel summary and the kernel timeline of the last of four training steps at 32 x 200 frames
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4trainprof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/train_step_once.py 32 128 200 > $O/run.log 2>&1; echo "rocprof rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_train_step.csv
cd $R
python - <<'P' > $O/timeline_last_step.txt
import csv, glob
f = sorted(glob.glob("gpurun_out/r4trainprof/prof/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last step: everything after the last gap > 50 ms
starts = [int(r["Start_Timestamp"]) for r in rows]; ends = [int(r["End_Timestamp"]) for r in rows]
cut = 0
for i in range(1, len(rows)):
    if starts[i] - max(ends[:i][-200:]) > 50_000_000: cut = i
rows = rows[cut:]; base = int(rows[0]["Start_Timestamp"])
# collapse runs of the same kernel name on the same queue
out, run = [], None
for r in rows:
    name = r["Kernel_Name"].replace("gvx::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    s, e, q = int(r["Start_Timestamp"]) - base, int(r["End_Timestamp"]) - base, r.get("Queue_Id", "?")
    if run and run[0] == name and run[1] == q: run[3] = e; run[4] += 1; run[5] += e - s
    else:
        if run: out.append(run)
        run = [name, q, s, e, 1, e - s]
out.append(run)
for name, q, s, e, n, busy in out:
    if busy >= 20000 or n > 4: print(f"{s/1e3:9.1f} {e/1e3:9.1f} us  q{q:>3} x{n:<4} busy {busy/1e3:8.1f} us  {name}")
P
rm -rf $O/prof
tail -2 $O/run.log
