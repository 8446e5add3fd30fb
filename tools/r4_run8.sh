#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4h
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -x -q -k "resident_decoder_equals or long_rows or edge_shapes" > $O/t1.log 2>&1; echo "test rc=$?"; tail -5 $O/t1.log
for L in 128 190 256; do
  for res in 1 0; do
    GVX_TF_RESIDENT=$res timeout -k 10 200 python bench.py --steps 8 --warmup 2 --tokens $L --no-extra --no-cpu-baseline > $O/bench_L${L}_res$res.log 2>&1
    echo "L=$L resident=$res: rc=$? $(grep -o '"decoder_step_us": [0-9.]*' $O/bench_L${L}_res$res.log) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_L${L}_res$res.log)"
  done
done
