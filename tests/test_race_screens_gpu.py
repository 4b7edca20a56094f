"""Races as a suite property.  Four kernels of this library order their LDS traffic BY HAND - counted s_waitcnt, raw
s_barrier phases, LDS / global-memory counters - instead of leaving it to the compiler: the ping-pong GEMM
(csrc/gemm8p.hip), the ring GEMM (csrc/gemm256.hip), the wave-specialised fused conv
(csrc/conv_fused.hip) and the attention kernels (csrc/attn.hip).  A wait that is one piece short passes almost every
launch (round 4: one launch in a few thousand differed, found by a lab screen only).  So, in the driver-run suite:

 (a) REPEAT screens: every launch form of tests/race_shapes.py N_SCREEN (4 000) times, each output compared on the device
     with the first launch's; one flag per form;
 (b) a PERTURBED arm: the same screen beside a second stream that keeps an HBM <-> HBM copy running (arrival order of the
     LDS-DMA pieces is what a missing wait depends on: shift it on purpose), and again with 16 CUs reserved (another grid,
     other XCD blocks, the copy really concurrent on the free CUs);
 (c) a DIFFERENTIAL build: libmelgpt_hip_vm0.so = the same sources with every counted wait replaced by a full drain
     (it cannot read a piece early, so its outputs are the intended ones); the production library must give the same
     bits on every form - a miscounted wait is a difference in the production arm only.
Reference call sites of the screened ops: transformer/minGPT.py:76-88,100-117; vqvae/big_model_attn_gan.py:75-135."""
import ctypes
import json
import os
import subprocess
import sys

import pytest
import torch

import race_shapes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
N_SCREEN = int(os.environ.get("MELGPT_SCREEN", "4000"))


def _counters():
    from melspec_gpt_vqvae_amd import _ffi

    r, p = ctypes.c_longlong(0), ctypes.c_longlong(0)
    assert _ffi.lib().melgpt_gemm_loop_launches(ctypes.byref(r), ctypes.byref(p)) == 0
    return r.value, p.value


class _CopyBeside:
    """keeps a 64 MB device-to-device copy per call queued on a side stream (HBM + L2 traffic beside the screened launches)"""

    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.src = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
        self.dst = torch.empty_like(self.src)

    def kick(self):
        with torch.cuda.stream(self.stream):
            self.dst.copy_(self.src, non_blocking=True)


def _screen(fn, n, beside=None):
    first = [t.clone() for t in fn()]
    bad = torch.zeros((), dtype=torch.int32, device=DEV)
    for i in range(n):
        if beside is not None and (i & 1) == 0:
            beside.kick()
        for a, b in zip(fn(), first):
            bad += (a != b).any().to(torch.int32)
    return int(bad)


def _arms(name, fn, family):
    """plain / beside a copy stream / 16 reserved CUs + copy stream -> differing launches per arm"""
    from melspec_gpt_vqvae_amd import _ffi

    L = _ffi.lib()
    out = {"plain": _screen(fn, N_SCREEN)}
    beside = _CopyBeside()
    out["beside a copy stream"] = _screen(fn, N_SCREEN // 2, beside)
    if family != "attn":                      # (the attention kernels' grids do not depend on the reservation)
        L.melgpt_set_reserved_cus(16)
        try:
            out["16 reserved CUs + copy stream"] = _screen(fn, N_SCREEN // 2, beside)
        finally:
            L.melgpt_set_reserved_cus(0)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("name", list(race_shapes.FORMS))
def test_repeat_screen(name):
    from melspec_gpt_vqvae_amd import ops

    family, build = race_shapes.FORMS[name]
    fn = build(torch, ops, DEV)
    r0, p0 = _counters()
    res = _arms(name, fn, family)
    r1, p1 = _counters()
    if family == "gemm8p":                    # the loop this form means to screen is the one that ran
        assert p1 - p0 >= 2 * N_SCREEN and r1 == r0, (name, r1 - r0, p1 - p0)
    assert all(v == 0 for v in res.values()), f"{name}: launches that differ from the first one: {res}"


@pytest.mark.parametrize("name", race_shapes.RING_FORMS)
def test_repeat_screen_ring_kernel(name):
    """the ring K loop (melgpt_set_gemm_pingpong(0): five half-unit slots, a counted wait per K unit) gives the bits of the
    ping-pong loop, launch after launch"""
    from melspec_gpt_vqvae_amd import _ffi, ops

    L = _ffi.lib()
    fn = race_shapes.FORMS[name][1](torch, ops, DEV)
    static = [t.clone() for t in fn()]
    L.melgpt_set_gemm_pingpong(0)
    try:
        r0, p0 = _counters()
        res = _arms(name, fn, "ring")
        r1, p1 = _counters()
        L.melgpt_set_reserved_cus(0)
        ring = fn()
        assert r1 - r0 >= 2 * N_SCREEN and p1 == p0, (name, r1 - r0, p1 - p0)
        assert all(torch.equal(a, b) for a, b in zip(ring, static)), "the ring loop changed the bits"
    finally:
        L.melgpt_set_gemm_pingpong(1)
        L.melgpt_set_reserved_cus(0)
    assert all(v == 0 for v in res.values()), f"{name}: launches that differ from the first one: {res}"


def test_production_build_equals_the_full_drain_build_bit_for_bit(tmp_path):
    """(c) of the module docstring.  The child loads lib/libmelgpt_hip_vm0.so through MELGPT_LAB_LIB and writes checksums."""
    from melspec_gpt_vqvae_amd import _ffi, build, ops

    vm0 = build.lib_path("vm0")
    assert os.path.exists(vm0), "build.build() makes the race-screen flavour next to the production library"
    assert not _ffi.LIB_PATH.endswith("_vm0.so"), "this process must run the PRODUCTION library"
    out = os.path.join(str(tmp_path), "vm0.json")
    env = dict(os.environ, MELGPT_LAB_LIB=vm0)
    r = subprocess.run([sys.executable, os.path.join(HERE, "race_worker.py"), out], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    ref = json.load(open(out))
    L = _ffi.lib()
    diff = []
    for name, (_, bld) in race_shapes.FORMS.items():
        got = [race_shapes.checksum(torch, t) for t in bld(torch, ops, DEV)()]
        if got != ref[name]:
            diff.append(name)
    L.melgpt_set_gemm_pingpong(0)
    try:
        for name in race_shapes.RING_FORMS:
            got = [race_shapes.checksum(torch, t) for t in race_shapes.FORMS[name][1](torch, ops, DEV)()]
            if got != ref["ring: " + name]:
                diff.append("ring: " + name)
    finally:
        L.melgpt_set_gemm_pingpong(1)
    assert len(ref) == len(race_shapes.FORMS) + len(race_shapes.RING_FORMS)
    assert not diff, f"production build differs from the full-drain build on: {diff}"
