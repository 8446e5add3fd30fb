#!/usr/bin/env python3
"""Developer tool: the last burst of kernels in a rocprofv3 --kernel-trace csv directory (bursts are separated by > 20 ms of idle
GPU): span, time with at least one kernel running, the largest idle gaps with the kernels on either side, time per kernel name."""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("gvx::", "").replace("(anonymous namespace)::", "")[:60]) for r in rows]
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 8):i]) > 20_000_000: cut = i
ev = ev[cut:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy, cur_end, gaps = 0, ev[0][0], []
last = ev[0]
for s, e, n in ev:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end - t0, last[2], n))
        busy += 0
        cur_start = s
    busy += max(0, e - max(s, cur_end))
    if e > cur_end: cur_end, last = e, (s, e, n)
print(f"{len(ev)} kernels, span {(t1 - t0) / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
small = sum(g[0] for g in gaps if g[0] < 20000)
print(f"gaps below 20 us: {small / 1e6:.2f} ms in total")
for g in sorted(gaps, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    print(f"  {g[0] / 1e3:8.1f} us idle at {g[1] / 1e3:9.1f} us: after {g[2]}  ->  before {g[3]}")
by = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev: by[n][0] += e - s; by[n][1] += 1
for n, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {t / 1e6:7.2f} ms {c:6d} x  {n}")
