#!/usr/bin/env python3
"""Build the CLOCK-STAMP flavour of the library (diagnostic, never shipped): gemm8p.hip, attn.hip and conv_fused.hip compiled
with -DMELGPT_CLOCK_STAMPS (csrc/common.h: thread 0 of every workgroup stamps s_memtime / s_memrealtime around the kernel),
linked with the in-tree objects of the other sources -> tools/lab/bin/libmelgpt_clock.so, loaded through MELGPT_LAB_LIB by
tools/lab/clock_lab.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from melspec_gpt_vqvae_amd import build as b

STAMPED = ("gemm8p.hip", "attn.hip", "conv_fused.hip")


def main():
    out = os.path.join(ROOT, "tools", "lab", "bin", "libmelgpt_clock.so")
    b.build(flavours=("bf16",))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objs = []
    for src in b.sources():
        name = os.path.basename(src)
        if name in STAMPED:
            obj = out[:-3] + "." + name[:-4] + ".o"
            subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(name, []), "-DMELGPT_CLOCK_STAMPS", "-c", src, "-o", obj])
            objs.append(obj)
        else:
            objs.append(os.path.join(b.OBJ, name[:-4] + ".o"))
    subprocess.check_call([b._hipcc(), "-shared", "-fPIC", f"--offload-arch={b.ARCH}", "-o", out, *objs])
    print(out)


if __name__ == "__main__":
    main()
