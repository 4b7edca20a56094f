"""Build libmelgpt_hip.so: every csrc/*.hip compiled for gfx950 with hipcc and linked in-tree
(melspec_gpt_vqvae_amd/lib/).  hipcc cross-compiles without a GPU, so this runs in the build
container; the resulting .so travels to the GPU box with the tree."""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmelgpt_hip.so")
ARCH = "gfx950"
# -Wno-inline-asm: the LDS-DMA asm blocks list m0 as clobbered (they set it); clang warns that m0 is "reserved"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
         "-ffp-contract=fast", "-fno-gpu-rdc"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found - libmelgpt_hip.so cannot be built")


def _digest(paths):
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(PKG), "include", "melgpt.h"))
    return sorted(hs)


def _compile(src, stamp, verbose):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    tag = obj + ".sha"
    if os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == stamp:
        return obj, False
    cmd = [_hipcc(), *FLAGS, "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    with open(tag, "w") as f:
        f.write(stamp)
    return obj, True


def build(verbose=False, force=False, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sources()
    hdr = _headers()
    if force:
        shutil.rmtree(OBJ)
        os.makedirs(OBJ)
    stamps = {s: _digest([s] + hdr) for s in srcs}
    jobs = jobs or min(6, max(1, (os.cpu_count() or 2) - 1))
    with cf.ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(lambda s: _compile(s, stamps[s], verbose), srcs))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB + ".tmp", *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
