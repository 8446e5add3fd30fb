#!/bin/bash
# round 4: where the caller's Prenet products start beside the encoder (GVX_ENC_FORK_AFTER = convolutions in front of the fork)
set -u
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r4enc2
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
for n in 3 2 1 0 3 2 1 0; do
  GVX_ENC_FORK_AFTER=$n timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_fork$n.json 2>/dev/null
  echo "fork after $n: rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_fork$n.json) $(grep -o '"stage_ms": {[^}]*}' $O/bench_fork$n.json)"
done
