#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4j
mkdir -p $O
GVX_GL_SLIM=1 timeout -k 10 400 python -m pytest tests/test_audio_gpu.py tests/test_fullsize_gpu.py -x -q -k "griffin or gl or audio or wav" > $O/t_slim.log 2>&1; echo "tests slim rc=$?"; tail -3 $O/t_slim.log
for slim in 0 1 0 1; do
  GVX_GL_SLIM=$slim timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --only-extra gl_60it_b256x800 > $O/bench_gl_$slim.log 2>&1
  echo "slim=$slim: rc=$? $(grep -o '"gl_60it_b256x800": {"ms": [0-9.]*' $O/bench_gl_$slim.log)"
done
