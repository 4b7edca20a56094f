"""Child process of tests/test_race_screens_gpu.py (test infrastructure): loads the RACE-SCREEN build of the library
(MELGPT_LAB_LIB = lib/libmelgpt_hip_vm0.so: the same sources with every hand-counted s_waitcnt turned into a full drain,
csrc/common.h MELGPT_VMCNT0), runs every form of tests/race_shapes.py once and writes the outputs' checksums as JSON.
usage: MELGPT_LAB_LIB=... python tests/race_worker.py <out.json>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch

import race_shapes
from melspec_gpt_vqvae_amd import _ffi, ops


def main():
    assert os.environ.get("MELGPT_LAB_LIB", "").endswith("libmelgpt_hip_vm0.so") and _ffi.LIB_PATH.endswith("_vm0.so")
    dev = "cuda:0"
    out = {}
    for name, (_, build) in race_shapes.FORMS.items():
        fn = build(torch, ops, dev)
        out[name] = [race_shapes.checksum(torch, t) for t in fn()]
    _ffi.lib().melgpt_set_gemm_pingpong(0)      # the ring K loop (csrc/gemm256.hip)
    for name in race_shapes.RING_FORMS:
        fn = race_shapes.FORMS[name][1](torch, ops, dev)
        out["ring: " + name] = [race_shapes.checksum(torch, t) for t in fn()]
    _ffi.lib().melgpt_set_gemm_pingpong(1)
    torch.cuda.synchronize()
    with open(sys.argv[1], "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
