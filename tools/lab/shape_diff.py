#!/usr/bin/env python3
"""LAB: per-shape ms per step of two bench.py JSON lines (files a, b): the rows of roofline.per_shape side by side."""
import json
import sys


def load(p):
    for l in open(p):
        l = l.strip()
        if l.startswith("{"):
            return json.loads(l)


a, b = load(sys.argv[1]), load(sys.argv[2])
ra = {r["shape"]: r for r in a["roofline"]["per_shape"]}
rb = {r["shape"]: r for r in b["roofline"]["per_shape"]}
print(f"step {a['ms_per_step']} -> {b['ms_per_step']} ms; family frac {a['roofline']['frac']} -> {b['roofline']['frac']}")
for k in ra:
    if k in rb:
        print(k.ljust(44), f"{ra[k]['ms_per_step']:8.3f} -> {rb[k]['ms_per_step']:8.3f} ms  {ra[k]['tflops']:7.1f} -> {rb[k]['tflops']:7.1f} TFLOP/s  x{ra[k]['ms_per_step'] / max(rb[k]['ms_per_step'], 1e-9):.3f}")
