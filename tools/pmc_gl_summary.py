#!/usr/bin/env python3
"""Per-launch averages of the rocprofv3 --pmc passes over `tools/run_config.py gl 256 1` (gpurun_out/pmc_gl/{a,b,c,d}: two SQ
counter sets, FETCH_SIZE, WRITE_SIZE - one pass each) for gl_iteration_kernel.  Traffic: FETCH_SIZE KiB x 1024 x 2 (gfx950
half-count, MI355X_MICROARCH.md) + WRITE_SIZE KiB x 1024."""
import collections
import csv
import glob
import json
import sys

out = {}
for d in "abcd":
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_gl/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gl_iteration_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = round(sum(v) / len(v), 1)
    for f in glob.glob(f"gpurun_out/pmc_gl/{d}/**/*kernel_trace.csv", recursive=True):
        ds = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "gl_iteration_kernel" in r["Kernel_Name"]]
        out[f"avg_us_pass_{d}"] = round(sum(ds) / len(ds) / 1e3, 1)
        out["launches"] = len(ds)
wc = out.get("SQ_WAVE_CYCLES")
if wc:
    out["wait_any_frac"] = round(out["SQ_WAIT_ANY"] / wc, 3)
    out["wait_inst_frac"] = round(out["SQ_WAIT_INST_ANY"] / wc, 3)
    out["active_inst_frac"] = round(out["SQ_ACTIVE_INST_ANY"] / wc, 3)
    out["lds_conflict_frac"] = round(out["SQ_LDS_BANK_CONFLICT"] / out["SQ_LDS_IDX_ACTIVE"], 3)
if out.get("SQ_WAVES"):
    out["valu_per_wave"] = round(out["SQ_INSTS_VALU"] / out["SQ_WAVES"], 1)
    out["lds_per_wave"] = round(out["SQ_INSTS_LDS"] / out["SQ_WAVES"], 1)
    out["vmem_rd_per_wave"] = round(out["SQ_INSTS_VMEM_RD"] / out["SQ_WAVES"], 1)
    # a wave64 VALU instruction occupies its SIMD for 4 cycles; 1024 SIMDs at 2.4 GHz
    out["valu_busy_frac"] = round(out["SQ_INSTS_VALU"] / 1024 * 4 / (out["avg_us_pass_b"] * 2400), 3)
if out.get("FETCH_SIZE") and out.get("WRITE_SIZE") is not None:
    out["hbm_bytes_per_launch"] = round(out["FETCH_SIZE"] * 2048 + out["WRITE_SIZE"] * 1024)
    out["hbm_bytes_per_frame"] = round(out["hbm_bytes_per_launch"] / (256 * 800), 1)
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
