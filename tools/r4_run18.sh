#!/bin/bash
# round 4: kernel timeline of an autoregressive decode of 64 rows (two 32-row chunks, resident loop)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r4r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/tools/run_config.py ar 64 1 > $O/run.log 2>&1; echo "rc=$?"; tail -2 $O/run.log
cd $GRAFT_REPO_ROOT
python3 tools/ar_timeline.py $O/trace 2 > $O/timeline.txt; python3 tools/kstats.py $O/trace 14 > $O/kstats.txt
rm -rf $O/trace
cat $O/timeline.txt | tail -80
