#!/bin/bash
# round 4: the multi-rank flow of bench.py rehearsed on one GPU (both ranks on cuda:0 over gloo: launcher, barrier, JSON line)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4x
mkdir -p $O
timeout -k 10 500 python bench.py --gpus 2 --rehearse-one-gpu --steps 2 --warmup 1 > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "rc=$?"; tail -3 $O/bench_2rank.err; cut -c1-700 $O/bench_2rank.json
