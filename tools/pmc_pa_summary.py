#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pa_pmc.py (gpurun_out/pmc_pa/<set>/...) for decoder_lstm_step_pa_kernel:
only the LAST 64 dispatches of the kernel count (the back-to-back replays; the loop's own launches ran serialised against
the resident attention kernel and spent their time in hand-off time-outs).  Units as in tools/pmc_sq_summary.py; FETCH_SIZE
is doubled (gfx950 half-count, MI355X_MICROARCH.md section HBM).
    python tools/pmc_pa_summary.py r03 [B [kernel [subdir of gpurun_out]]]"""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "decoder_lstm_step_pa_kernel"
SUB = sys.argv[4] if len(sys.argv) > 4 else "pmc_pa"
REPLAYS = 64
vals = collections.defaultdict(list)
durs = []
for f in sorted(glob.glob(f"gpurun_out/{SUB}/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)   # counter -> [(dispatch id, value)]
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, lst in per.items():
        lst.sort()
        vals[c] += [v for _, v in lst[-REPLAYS:]]
for f in sorted(glob.glob(f"gpurun_out/{SUB}/**/*kernel_trace.csv", recursive=True)):
    rows = [(int(r["Dispatch_Id"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]]
    rows.sort()
    durs += [d for _, d in rows[-REPLAYS:]]
a = {c: sum(v) / len(v) for c, v in vals.items() if v}
row = {"kernel": KERNEL, "batch": B, "dispatches_per_counter": REPLAYS, **{c: round(v, 1) for c, v in a.items()}}
wc = a.get("SQ_WAVE_CYCLES")
if wc:
    row["wait_any_frac"] = round(a.get("SQ_WAIT_ANY", 0) / wc, 3)
    row["wait_inst_frac"] = round(a.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
    row["active_inst_frac"] = round(a.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
if a.get("SQ_LDS_IDX_ACTIVE"):
    row["lds_conflict_frac"] = round(a.get("SQ_LDS_BANK_CONFLICT", 0) / a["SQ_LDS_IDX_ACTIVE"], 3)
if a.get("SQ_BUSY_CYCLES") and a.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
    row["mfma_busy_frac"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["SQ_BUSY_CYCLES"] / 32 * 1024), 3)
if durs:
    row["avg_duration_us_under_pmc"] = round(sum(durs) / len(durs) / 1e3, 2)
    if a.get("SQ_BUSY_CYCLES"):
        row["clock_ghz_est"] = round(a["SQ_BUSY_CYCLES"] / 32 / (sum(durs) / len(durs)), 2)
if a.get("TCC_HIT_sum") is not None and a.get("TCC_MISS_sum") is not None and a["TCC_HIT_sum"] + a["TCC_MISS_sum"] > 0:
    row["l2_hit_rate"] = round(a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"]), 3)
if a.get("FETCH_SIZE") is not None:
    row["fetch_bytes_corrected"] = round(a["FETCH_SIZE"] * 1024 * 2)
if a.get("WRITE_SIZE") is not None:
    row["write_bytes"] = round(a["WRITE_SIZE"] * 1024)
if "fetch_bytes_corrected" in row and "write_bytes" in row:
    row["hbm_bytes_per_launch"] = row["fetch_bytes_corrected"] + row["write_bytes"]
json.dump({"round": tag, "command": "rocprofv3 --kernel-trace --output-format csv --pmc <set> -d gpurun_out/pmc_pa/<set> -- python3 tools/pa_pmc.py",
           "note": "last 64 dispatches (back-to-back replays of a mid-sequence launch) only", "kernel": row},
          open(f"profiles/{tag}_pmc_sq_{'pa' if SUB == 'pmc_pa' else SUB}.json", "w"), indent=1)
print(json.dumps(row, indent=1))
