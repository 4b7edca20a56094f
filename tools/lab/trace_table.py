#!/usr/bin/env python3
"""Prints the LAST `n` kernel launches of a rocprofv3 --kernel-trace CSV in launch order: name, duration, gap to the
previous launch's end (us).  usage: trace_table.py <kernel_trace.csv> <launches per iteration>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
rows = rows[-n:]
prev = None
tot = gaps = 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    tot += (e - s) / 1e3
    gaps += max(gap, 0.0)
    print(f"{(e - s) / 1e3:9.1f} {gap:7.1f}  {r['Kernel_Name'][:110]}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
print(f"kernels {tot:.1f} us, gaps {gaps:.1f} us, span {(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3:.1f} us")
