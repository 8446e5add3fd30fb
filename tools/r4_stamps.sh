#!/bin/bash
# round 4: stamps of both resident loops (the stamps build: python -m genvox_amd.build --stamps)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r4final
mkdir -p $O
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_resident.py 32 200 > $O/stamps_tf.txt 2>&1; echo "stamps tf rc=$?"
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_ar.py 1 128 > $O/stamps_ar_b1.txt 2>&1; echo "stamps ar b1 rc=$?"
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_ar.py 32 128 > $O/stamps_ar_b32.txt 2>&1; echo "stamps ar b32 rc=$?"
