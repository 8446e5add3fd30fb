#!/bin/bash
# round 4: kernel timeline of one teacher-forced 1 x 800 forward (the single-utterance shape)
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4timeline
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof1 -- python3 $R/tools/run_config.py tf 1 2 > $O/run1.log 2>&1; echo "rocprof rc=$?"
cd $R
python3 tools/kernel_timeline.py $O/prof1 > $O/timeline_forward_b1.txt 2>&1; echo "timeline rc=$?"
rm -rf $O/prof1
tail -1 $O/run1.log
