#!/usr/bin/env python3
"""Build a LAB flavour of libmelgpt_hip.so in which one source of csrc/ is replaced by a lab copy (same C ABI), for A/B runs
through MELGPT_LAB_LIB (see _ffi.py): the other objects are the ones of the in-tree build.
  python tools/lab/build_lab_lib.py gemm8p=melspec_gpt_vqvae_amd/csrc/gemm8p.hip tools/lab/bin/libmelgpt_p8lab.so -DP8_LAB [-DFLAG ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from melspec_gpt_vqvae_amd import build as b


def main():
    name, src = sys.argv[1].split("=")
    out = os.path.abspath(sys.argv[2])
    extra = sys.argv[3:]
    b.build()                                                    # the in-tree objects are current
    os.makedirs(os.path.dirname(out), exist_ok=True)
    obj = out[:-3] + "." + name + ".o"
    subprocess.check_call([b._hipcc(), *b.FLAGS, *extra, "-I", b.CSRC, "-I", os.path.join(ROOT, "include"), "-c",
                           os.path.join(ROOT, src), "-o", obj])
    others = [os.path.join(b.OBJ, os.path.basename(s)[:-4] + ".o") for s in b.sources() if os.path.basename(s)[:-4] != name]
    subprocess.check_call([b._hipcc(), "-shared", "-fPIC", f"--offload-arch={b.ARCH}", "-o", out, obj, *others])
    print(out)


if __name__ == "__main__":
    main()
