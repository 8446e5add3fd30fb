#!/bin/bash
set -u
# GPU run 2 of round 3: MFMA query slabs A/B, stamps, SQ / TCC counters of the pa kernel
set -o pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_run2; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -q -x -k "persistent or handoff or depth6 or fixture" > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
for i in 1 2; do
  GVX_LIB=libgenvox_amd_noslab.so python bench.py --no-extra --no-cpu-baseline --steps 10 > $O/bench_noslab_$i.json 2>/dev/null || exit 1
  python bench.py --no-extra --no-cpu-baseline --steps 10 > $O/bench_slab_$i.json 2>/dev/null || exit 1
done
GVX_LIB=libgenvox_amd_stamps.so python tools/stamps.py 32 > $O/stamps_lstm.txt 2>&1 || exit 1
GVX_LIB=libgenvox_amd_stamps.so python tools/stamps_pa.py 32 > $O/stamps_pa.txt 2>&1 || exit 1
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc_pa/sq -- python3 $R/tools/pa_pmc.py > $R/$O/pmc_sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_pa/tcc -- python3 $R/tools/pa_pmc.py > $R/$O/pmc_tcc.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/pmc_pa/fetch -- python3 $R/tools/pa_pmc.py > $R/$O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/pmc_pa/write -- python3 $R/tools/pa_pmc.py > $R/$O/pmc_write.log 2>&1 || exit 1
cd $R
python tools/pmc_pa_summary.py r03 32 > $O/pmc_summary.txt 2>&1
tail -3 $O/tests.log
