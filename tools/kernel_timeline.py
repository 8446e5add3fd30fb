#!/usr/bin/env python3
"""Developer tool: kernel timeline of the LAST teacher-forced forward in a rocprofv3 --kernel-trace csv directory
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/run_config.py tf 32 2; python3 tools/kernel_timeline.py DIR).
Start / end in us from the first kernel shown, hardware queue, grid size; the decoder loop's step launches are collapsed."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
enc = [i for i, r in enumerate(rows) if "encoder_lstm_persistent" in r["Kernel_Name"] or "encoder_lstm_step" in r["Kernel_Name"]]
last_enc_start = int(rows[enc[-1]]["Start_Timestamp"])
t0 = last_enc_start - (1500000 if "persistent" in rows[enc[-1]]["Kernel_Name"] else 3000000)
base, steps, step_t = None, 0, [0, 0]
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0: continue
    if base is None: base = s
    name = r["Kernel_Name"].replace("gvx::", "").replace("(anonymous namespace)::", "")
    if "decoder_lstm_step" in name or "encoder_lstm_step" in name:
        if steps == 0: step_t[0] = s
        steps += 1; step_t[1] = e
        continue
    if steps:
        print(f"{(step_t[0]-base)/1e3:9.1f} {(step_t[1]-base)/1e3:9.1f} us  ...  {steps} step launches")
        steps = 0
    print(f"{(s-base)/1e3:9.1f} {(e-base)/1e3:9.1f} us  q{r.get('Queue_Id','?'):>3} grid {r.get('Grid_Size_X','?'):>8}  {name[:70]}")
