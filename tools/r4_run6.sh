#!/bin/bash
# round 4: robustness tests (strict forward, two threads, training recovery, RCCL in-process) + the bf16-neighbour question on CU masks
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4f
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_training_gpu.py tests/test_dist_gpu.py -x -q -k "strict or two_models or timeout or recovers or survives or rccl or resident_decoder or sticky or serialized" > $O/t.log 2>&1; echo "tests rc=$?"; tail -15 $O/t.log
for m in none halves xcds same; do
  if [ $m = none ]; then timeout -k 10 120 tools/micro/mfma_bf16_neighbour 1024 2>&1 | head -6 > $O/neigh_$m.txt; else MASK=$m timeout -k 10 120 tools/micro/mfma_bf16_neighbour 1024 > $O/neigh_$m.txt 2>&1; fi
  echo "== mask $m"; cat $O/neigh_$m.txt
done
