#!/usr/bin/env python3
"""Per-kernel durations of the decode steps in a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -- python3
tools/profile_decode.py B`): the last 200 tokens, grouped by kernel and launch geometry."""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start"))
per_token = int(sys.argv[2]) if len(sys.argv) > 2 else 124
last = rows[-per_token * 200:]
d = collections.defaultdict(list)
for n, s, e, gx, gy, wx in last:
    m = re.search(r'(\w+)<', n) or re.search(r'(\w+)\(', n)
    d[f"{m.group(1) if m else n[:40]} grid={gx // wx}x{gy} wg={wx}"].append((e - s) / 1e3)
print(f"span per token {(last[-1][2] - last[0][1]) / 200e3:.1f} us")
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{n:56s} n={len(v):6d} med={v[len(v) // 2]:7.2f} us  per token {sum(v) / 200:8.2f} us")
