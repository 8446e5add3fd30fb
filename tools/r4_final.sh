#!/bin/bash
# round 4, final tree: smoke, the whole GPU suite, the bench line as the driver runs it, rocprofv3's kernel summary of the headline
# configuration, rows of 190 / 256 tokens, stamps of both resident loops (the stamps steps need the diagnostic library:
# `python -m genvox_amd.build --stamps` in the container before the run - build() removes one older than the C ABI's header)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r4final
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for L in 190 256; do
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --tokens $L --no-extra --no-cpu-baseline > $O/bench_L$L.json 2>/dev/null
  echo "L=$L: rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_L$L.json) $(grep -o '"decoder_step_us": [0-9.]*' $O/bench_L$L.json)"
done
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_resident.py 32 200 > $O/stamps_tf.txt 2>&1; echo "stamps tf rc=$?"
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_ar.py 1 128 > $O/stamps_ar_b1.txt 2>&1; echo "stamps ar b1 rc=$?"
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_ar.py 32 128 > $O/stamps_ar_b32.txt 2>&1; echo "stamps ar b32 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_bench_no_extra.csv
rm -rf $O/prof
cut -c1-600 $O/bench.json
