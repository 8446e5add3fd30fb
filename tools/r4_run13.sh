#!/bin/bash
# round 4: phase stamps of the autoregressive resident loop
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4m
mkdir -p $O
python -m genvox_amd.build --stamps > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_ar.py 1 128 > $O/stamps_b1.txt 2>&1; echo "rc=$?"
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_ar.py 32 128 > $O/stamps_b32.txt 2>&1; echo "rc=$?"
cat $O/stamps_b1.txt
