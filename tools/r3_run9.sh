#!/bin/bash
set -u
# GPU run 9 of round 3: profiles of the final-ish tree
set -o pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_run9; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps.py 32 190 > $O/stamps_lstm_L190.txt 2>&1 || exit 1
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_pa.py 32 190 > $O/stamps_pa_L190.txt 2>&1 || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- python3 $R/bench.py > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_ar1 -- python3 $R/tools/run_config.py ar 1 > $R/$O/prof_ar1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_ar64 -- python3 $R/tools/run_config.py ar 64 > $R/$O/prof_ar64.log 2>&1 || exit 1
cd $R
for d in prof_bench prof_ar1 prof_ar64; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$d.csv
  find $O/$d -name "*kernel_trace.csv" -delete; find $O/$d -name "*.db" -delete
done
python tools/kstats.py $O/prof_ar1 8; python tools/kstats.py $O/prof_bench 6
