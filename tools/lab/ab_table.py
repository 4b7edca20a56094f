#!/usr/bin/env python3
"""LAB: one line per JSON record of tools/lab/p8_ab.py on stdin (shape, bit-identical, repeatable, TFLOP/s of the two arms, ratio)."""
import json
import sys

for l in sys.stdin:
    l = l.strip()
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    print(d["shape"][:46].ljust(46), d["bit_identical"], d["differing"], d["p8_repeatable"], d["ring_tflops_med"], d["p8_tflops_med"], d["speedup_med"])
