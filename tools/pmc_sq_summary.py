#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc SQ_* / TCC_* passes of `bench.py --steps 1 --warmup 1` into profiles/ (per-kernel
averages and the ratios DESIGN.md quotes).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed
over waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs, SQ_BUSY_CYCLES cycles summed over shader engines
(MI355X_MICROARCH.md, PMC slots)."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_sq", "pmc_tcc"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "gvx::" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)   # kernel durations during the SQ pass (counter collection serialises and slows launches)
for f in glob.glob("gpurun_out/pmc_sq/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for k, c in agg.items():
    a = {n: sum(v) / len(v) for n, v in c.items()}
    n = len(next(iter(c.values())))
    row = {"kernel": k, "launches": n, **{m: round(v, 1) for m, v in a.items()}}
    wc = a.get("SQ_WAVE_CYCLES")
    if wc:
        row["wait_any_frac"] = round(a.get("SQ_WAIT_ANY", 0) / wc, 3)            # waves parked at s_waitcnt / barrier
        row["wait_inst_frac"] = round(a.get("SQ_WAIT_INST_ANY", 0) / wc, 3)      # issue stalls (MFMA RAW / pipe)
        row["active_inst_frac"] = round(a.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
    if a.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_conflict_frac"] = round(a.get("SQ_LDS_BANK_CONFLICT", 0) / a["SQ_LDS_IDX_ACTIVE"], 3)
    if a.get("SQ_BUSY_CYCLES") and a.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines, MFMA busy cycles over the 1024 SIMDs:
        # share of the kernel's cycles in which a SIMD's MFMA pipe is busy, averaged over the chip
        row["mfma_busy_frac"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["SQ_BUSY_CYCLES"] / 32 * 1024), 3)
        if dur.get(k):
            row["avg_duration_us_under_pmc"] = round(sum(dur[k]) / len(dur[k]) / 1e3, 2)
            row["clock_ghz_est"] = round(a["SQ_BUSY_CYCLES"] / 32 / (sum(dur[k]) / len(dur[k])), 2)
    if a.get("TCC_HIT_sum") is not None and a.get("TCC_MISS_sum") is not None and a["TCC_HIT_sum"] + a["TCC_MISS_sum"] > 0:
        row["l2_hit_rate"] = round(a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"]), 3)
    rows.append(row)
rows.sort(key=lambda r: -r.get("SQ_BUSY_CYCLES", 0) * r["launches"])
json.dump({"round": tag, "command": "rocprofv3 --kernel-trace --pmc <SQ_* | TCC_* GRBM_GUI_ACTIVE> -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline",
           "kernels": rows}, open(f"profiles/{tag}_pmc_sq_b32.json", "w"), indent=1)
for r in rows[:8]:
    print(json.dumps(r))
