#!/bin/bash
set -u
set -o pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3_run7
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc_pa64/sq -- python3 $R/tools/pa_pmc.py 64 > $R/gpurun_out/r3_run7/pmc_sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_pa64/tcc -- python3 $R/tools/pa_pmc.py 64 > $R/gpurun_out/r3_run7/pmc_tcc.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/pmc_pa64/fetch -- python3 $R/tools/pa_pmc.py 64 > $R/gpurun_out/r3_run7/pmc_fetch.log 2>&1 || exit 1
cd $R
python tools/pmc_pa_summary.py r03 64 decoder_lstm_step_pa64_kernel pmc_pa64
