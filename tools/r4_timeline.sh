#!/bin/bash
# round 4, final tree: kernel timeline of one teacher-forced 32 x 800 forward (what runs beside what around the resident loop)
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4timeline
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/tools/run_config.py tf 32 2 > $O/run.log 2>&1; echo "rocprof rc=$?"
cd $R
python3 tools/kernel_timeline.py $O/prof > $O/timeline_forward_b32.txt 2>&1; echo "timeline rc=$?"
rm -rf $O/prof
tail -1 $O/run.log
