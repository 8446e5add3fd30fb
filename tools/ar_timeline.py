#!/usr/bin/env python3
"""Developer tool: kernel timeline of the LAST autoregressive call in a rocprofv3 --kernel-trace csv directory
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/run_config.py ar 64 1; python3 tools/ar_timeline.py DIR [n_chunks]).
Start / end in us from the first kernel shown, hardware queue, grid size; gaps above 20 us between kernels are marked."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
res = [i for i, r in enumerate(rows) if "decoder_ar_resident_kernel" in r["Kernel_Name"]]
first = res[-n_chunks]
t0 = int(rows[first]["Start_Timestamp"]) - 3000000
base, last_end = None, None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0: continue
    if base is None: base = s
    name = r["Kernel_Name"].replace("gvx::", "").replace("(anonymous namespace)::", "")
    gap = "" if last_end is None or s - last_end < 20000 else f"   <- {(s - last_end) / 1e3:.0f} us after the previous kernel ended"
    print(f"{(s-base)/1e3:10.1f} {(e-base)/1e3:10.1f} us  q{r.get('Queue_Id','?'):>3} grid {r.get('Grid_Size_X','?'):>8}  {name[:64]}{gap}")
    last_end = max(last_end or 0, e)
