#!/bin/bash
# round 4: the 128 x 128 GEMM tile on eight waves (default) vs four (GVX_GEMM_8W=0), A/B on the bench line; SQ counters of the
# Postnet's GEMM shapes
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
cd "$R"
O=$R/gpurun_out/r4gemm3
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "fixture or oracle_and_stages or shapes_against" > $O/gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -3 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_$name.json 2>/dev/null
  echo "$name: rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.json) $(grep -o '"stage_ms": {[^}]*}' $O/bench_$name.json)"
}
run default A=1
run fourw GVX_GEMM_8W=0
run default2 A=1
run fourw2 GVX_GEMM_8W=0
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq -- python3 $R/tools/run_config.py postnet 32 > $O/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $O/pmc_tcc -- python3 $R/tools/run_config.py postnet 32 > $O/pmc_tcc.log 2>&1; echo "pmc tcc rc=$?"
cd $R
python - <<'P'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
rows = []
for f in glob.glob("gpurun_out/r4gemm3/pmc_sq/**/*counter_collection.csv", recursive=True) + glob.glob("gpurun_out/r4gemm3/pmc_tcc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "") + " grid " + r.get("Grid_Size", "?")
        if "gvx::" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    a = {n: sum(v) / len(v) for n, v in c.items()}
    wc = a.get("SQ_WAVE_CYCLES", 0)
    if not wc: continue
    row = {"kernel": k, "n": len(c["SQ_WAVE_CYCLES"]), "wait_any": round(a["SQ_WAIT_ANY"] / wc, 3), "wait_inst": round(a["SQ_WAIT_INST_ANY"] / wc, 3),
           "active": round(a["SQ_ACTIVE_INST_ANY"] / wc, 3), "mfma_busy": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["SQ_BUSY_CYCLES"] / 32 * 1024), 3),
           "busy_cycles_per_se": round(a["SQ_BUSY_CYCLES"] / 32), "lds_conflict": round(a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1), 3)}
    if a.get("TCC_HIT_sum") is not None: row["l2_hit"] = round(a["TCC_HIT_sum"] / max(a["TCC_HIT_sum"] + a["TCC_MISS_sum"], 1), 3)
    rows.append(row); print(json.dumps(row))
json.dump({"round": "r04", "command": "rocprofv3 --kernel-trace --pmc <SQ_* | TCC_*> -- python3 tools/run_config.py postnet 32 (tools/r4_gemm3.sh)",
           "note": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024): share of the kernel's cycles a SIMD's matrix pipe is busy, averaged over the chip",
           "kernels": rows}, open("gpurun_out/r4gemm3/pmc_sq_postnet_gemm.json", "w"), indent=1)
P
rm -rf $O/pmc_sq $O/pmc_tcc
