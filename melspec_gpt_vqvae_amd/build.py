"""Build libmelgpt_hip.so (+ libmelgpt_hip_fp16.so): every csrc/*.hip compiled for gfx950 with hipcc and linked
in-tree (melspec_gpt_vqvae_amd/lib/).  hipcc cross-compiles without a GPU, so this runs in the build
container; the resulting .so files travel to the GPU box with the tree.  Two flavours of the same sources: the
16-bit lane as bfloat16 (default) and as IEEE half (-DMELGPT_HALF_FP16; csrc/common.h)."""
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


# per-source extra flags.  conv_fused.hip: no SLP vectorisation - hipcc packs the staging pass' scalar f32 arithmetic into
# v_pk_fma_f32 / v_pk_mul_f32, which run slower beside a SIMD partner's MFMAs (guide, cycle constants: "an anti-lever
# beside MFMAs"; the wave-specialised conv's conversion pass: 106 k -> 90 k cycles per 5 tiles, profiles/r04_conv_lab.md)
EXTRA_FLAGS = {"conv_fused.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found - libmelgpt_hip.so cannot be built")


def _digest(paths):
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
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


FLAVOURS = {"bf16": ([], "libmelgpt_hip.so", ""), "fp16": (["-DMELGPT_HALF_FP16"], "libmelgpt_hip_fp16.so", "_fp16"),
            # race-screen build (csrc/common.h MELGPT_VMCNT0): every hand-counted wait is a full drain.  Test infrastructure:
            # only the sources that count their waits are recompiled, the rest are the bf16 flavour's objects; loaded by
            # tests/test_race_screens_gpu.py in a child process (MELGPT_LAB_LIB), never by the product
            "vm0": (["-DMELGPT_VMCNT0"], "libmelgpt_hip_vm0.so", "_vm0")}
VM0_SOURCES = ("gemm8p.hip", "gemm256.hip", "conv_fused.hip")


def lib_path(flavour="bf16"):
    return os.path.join(LIBDIR, FLAVOURS[flavour][1])


def _compile(src, stamp, verbose, flavour="bf16"):
    extra, _, suffix = FLAVOURS[flavour]
    obj = os.path.join(OBJ + suffix, os.path.basename(src)[:-4] + ".o")
    tag = obj + ".sha"
    if os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == stamp:
        return obj, False
    cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), *extra, "-c", src, "-o", obj]
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


def build(verbose=False, force=False, jobs=None, flavours=("bf16", "fp16", "vm0")):
    """-> path of the default (bf16) library; builds every flavour in `flavours` ("vm0" needs "bf16" in front of it)."""
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sources()
    hdr = _headers()
    stamps = {s: _digest([s] + hdr) for s in srcs}
    jobs = jobs or min(6, max(1, (os.cpu_count() or 2) - 1))
    for fl in flavours:
        objdir = OBJ + FLAVOURS[fl][2]
        if force and os.path.isdir(objdir):
            shutil.rmtree(objdir)
        os.makedirs(objdir, exist_ok=True)
        own = [s for s in srcs if fl != "vm0" or os.path.basename(s) in VM0_SOURCES]
        with cf.ThreadPoolExecutor(jobs) as ex:
            res = list(ex.map(lambda s: _compile(s, stamps[s], verbose, fl), own))
        objs = [o for o, _ in res]
        lib = lib_path(fl)
        stale = False
        if fl == "vm0":
            # the sources that do not count their waits are borrowed from the bf16 flavour's objects: they must exist (build
            # "bf16" first) and a relink is due whenever one of them is newer than the library
            borrowed = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs if s not in own]
            missing = [o for o in borrowed if not os.path.exists(o)]
            if missing:
                raise RuntimeError(f"vm0 flavour borrows the bf16 flavour's objects: build 'bf16' first (missing {missing})")
            stale = os.path.exists(lib) and any(os.path.getmtime(o) > os.path.getmtime(lib) for o in borrowed)
            objs += borrowed
        if any(ch for _, ch in res) or stale or not os.path.exists(lib):
            cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib + ".tmp", *objs]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            os.replace(lib + ".tmp", lib)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
