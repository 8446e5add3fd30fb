#!/bin/bash
# round 4: stamps of the teacher-forced resident loop at batch 1 (vector-ALU mode)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4z
mkdir -p $O
GVX_LIB=libgenvox_amd_stamps.so timeout -k 10 200 python tools/stamps_resident.py 1 200 > $O/stamps_tf_b1.txt 2>&1; echo "rc=$?"
grep -v "amdgpu.ids" $O/stamps_tf_b1.txt
