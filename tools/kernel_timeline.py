import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last encoder_lstm_persistent launch and print kernels from 1.5 ms before to its end + 0.3 ms
idx = [i for i, r in enumerate(rows) if "encoder_lstm_persistent" in r["Kernel_Name"]][-1]
t_end = int(rows[idx]["End_Timestamp"]); t0 = int(rows[idx]["Start_Timestamp"]) - 1500000
print("columns:", list(rows[0].keys()))
base = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or s > t_end + 100000: continue
    if base is None: base = s
    name = r["Kernel_Name"].replace("gvx::", "").replace("(anonymous namespace)::", "")[:46]
    print(f"{(s-base)/1e3:9.1f} {(e-base)/1e3:9.1f} us  q{r.get('Queue_Id','?'):>3} grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>7}  {name}")
