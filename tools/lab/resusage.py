#!/usr/bin/env python3
"""usage: resusage.py <hipcc -Rpass-analysis=kernel-resource-usage stderr log>  ->  one line per kernel"""
import re
import sys

txt = open(sys.argv[1]).read()
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0]

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    short = re.sub(r'_ZN\d+_GLOBAL__N_1', '', name)[:64]
    print(f"{short:64s} vgpr={g('VGPRs'):4d} agpr={g('AGPRs'):4d} vspill={g('VGPRs Spill'):4d} sspill={g('SGPRs Spill'):4d} "
          f"scratch={g('ScratchSize .bytes/lane.'):5d} occ={g('Occupancy .waves/SIMD.')}")
