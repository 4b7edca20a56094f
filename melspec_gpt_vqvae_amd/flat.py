"""Flat parameter / gradient storage sized for one MI355X (288 GB HBM3E).

All parameters of a module tree live in ONE contiguous f32 buffer (Linear weights first = the AdamW
weight-decay group of the reference, minGPT.py:618-665; everything else after), with `param.data` re-pointed
to views, so that
  * key/query/value weights (and biases) of a block are adjacent -> the packed [3C, C] QKV operand is a view,
  * gradients are written by the GEMM epilogues straight into one contiguous f32 buffer -> the optimizer is a
    single fused launch per group and data-parallel all-reduce runs over a few large contiguous buckets,
  * a bf16 shadow of the whole buffer (the MFMA operands) is refreshed with one cast launch per step.
The reference's state_dict names/shapes are untouched: nn.Parameters stay, only their storage is shared.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _ffi, ops

ALIGN = 8  # elements: keeps every view 16-byte aligned in both the f32 buffer and the bf16 shadow
SHADOW_EPOCH = [0]  # bumped whenever ANY 16-bit shadow is rewritten: views of a shadow carry no version of their own, so
                    # whatever is derived from such a view (ops._ln_folded) keys on this


def _is_decay(module, pname):
    return isinstance(module, nn.Linear) and pname == "weight"


class FlatParams:
    def __init__(self, root: nn.Module):
        named = []
        seen = set()
        for mn, mod in root.named_modules():
            for pn, p in mod.named_parameters(recurse=False):
                if id(p) in seen:
                    continue
                seen.add(id(p))
                named.append(((mn + "." + pn) if mn else pn, p, _is_decay(mod, pn)))
        if not named:
            raise ValueError("module has no parameters")
        dev = named[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("parameters must be on the GPU before the first forward (there is no CPU path)")
        ordered = [x for x in named if x[2]] + [x for x in named if not x[2]]
        self.names, self.params, self.offsets = [], [], []
        off = 0
        self.n_decay = 0
        for name, p, dec in ordered:
            if p.dtype != torch.float32:
                raise RuntimeError(f"{name}: master parameters are kept in float32 (got {p.dtype})")
            self.names.append(name)
            self.params.append(p)
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            if dec:
                self.n_decay = off
        self.total = off
        self.device = dev
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = None
        self._shadow_key = None
        self._index = {}
        # writes that go THROUGH the flat buffer (the fused optimizer, dp.broadcast_parameters) do not bump the
        # parameters' tensor version counters: they bump this instead, and tensor_version() folds it into every cache key
        self.generation = 0
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            v = self.data[o:o + p.numel()].view(p.shape)
            with torch.no_grad():
                ops.cast(p.data.contiguous(), torch.float32, out=v)
            old_grad = p.grad
            p.data = v
            p._melgpt_fp = self
            self._index[id(p)] = i
            p.grad = None
            if old_grad is not None:
                gv = self.grad_view(p)
                ops.cast(old_grad.contiguous().float(), torch.float32, out=gv)
                p.grad = gv
        self._ptrs = [p.data_ptr() for p in self.params]

    # ------------------------------------------------------------------ validity
    def intact(self):
        return all(p.data_ptr() == q for p, q in zip(self.params, self._ptrs))

    def covers(self, p):
        return id(p) in self._index

    # ------------------------------------------------------------------ views
    def _slice(self, buf, p, n_after=0):
        i = self._index[id(p)]
        o = self.offsets[i]
        return buf[o:o + p.numel()]

    def grad_view(self, p):
        return self._slice(self.grad, p).view(p.shape)

    def packed(self, buf, plist):
        """one view over consecutive parameters (e.g. key/query/value weights) -> (sum rows, cols)."""
        i0 = self._index[id(plist[0])]
        o = self.offsets[i0]
        n = 0
        for j, p in enumerate(plist):
            i = self._index[id(p)]
            assert i == i0 + j and self.offsets[i] == o + n, "parameters are not adjacent in the flat buffer"
            n += p.numel()
        shape = (sum(p.shape[0] for p in plist),) + tuple(plist[0].shape[1:])
        return buf[o:o + n].view(shape)

    # ------------------------------------------------------------------ compute copies
    def compute_buffer(self, dtype):
        """the buffer MFMA operands are read from: the f32 master itself, or the bf16 shadow (refreshed with
        one cast launch whenever any parameter was modified in place)."""
        if dtype == torch.float32:
            return self.data
        key = (sum(p._version for p in self.params), self.generation)
        if self.shadow is None:
            self.shadow = torch.empty(self.total, dtype=_ffi.HALF_DTYPE, device=self.device)
            self._shadow_key = None
        if key != self._shadow_key:
            ops.cast(self.data, _ffi.HALF_DTYPE, out=self.shadow)
            self._shadow_key = key
            SHADOW_EPOCH[0] += 1
        return self.shadow

    def mark_shadow_fresh(self):
        """called by the fused optimizer, which writes the bf16 shadow itself."""
        self._shadow_key = (sum(p._version for p in self.params), self.generation)
        SHADOW_EPOCH[0] += 1

    # ------------------------------------------------------------------ gradients
    def grad_target(self, p):
        """(tensor to write dW into, accumulate?) - re-attaches the flat view after zero_grad(set_to_none=True)."""
        gv = self.grad_view(p)
        if p.grad is None:
            p.grad = gv
            return gv, False
        if p.grad.data_ptr() == gv.data_ptr():
            return gv, True
        return p.grad, True  # foreign .grad tensor: accumulate into it

    def packed_grad_target(self, plist):
        accs = []
        for p in plist:
            gv = self.grad_view(p)
            if p.grad is None:
                p.grad = gv
                accs.append(False)
            else:
                assert p.grad.data_ptr() == gv.data_ptr(), "packed gradient needs the flat .grad views"
                accs.append(True)
        assert all(a == accs[0] for a in accs), "mixed None / set .grad inside one packed parameter group"
        return self.packed(self.grad, plist), accs[0]

    def zero_grad(self):
        self.grad.zero_()

    def zero_missing_grads(self):
        """Zero the flat-gradient slices of parameters that received NO gradient since zero_grad (p.grad is None:
        unused branch, frozen sub-module, class embedder when no class token is fed).  The fused optimizer and the
        data-parallel exchange run over the whole buffer, so a stale slice would otherwise be applied / summed;
        torch.optim.AdamW in the reference skips such parameters - a zero gradient with zero moments is the same
        no-op only while their moments are zero and without weight decay: FusedAdamW.step() therefore puts such
        parameters (and their moments) back after its launches, using the list returned here."""
        missing = [p for p in self.params if p.grad is None]
        for p in missing:
            self._slice(self.grad, p).zero_()
        return missing


def tensor_version(t):
    """Cache key component for anything derived from a parameter (packed / repacked weights, the prepared codebook
    image): storage, in-place version, and the generation of the flat store that owns it (see FlatParams.generation)."""
    if t is None:
        return None
    fp = getattr(t, "_melgpt_fp", None)
    return (t.data_ptr(), t._version, fp.generation if fp is not None else 0)


def ensure_flat(module: nn.Module) -> FlatParams:
    """The FlatParams owning `module`'s parameters: an intact one made for it or for an ancestor, else a new one."""
    fp = getattr(module, "_melgpt_flat", None)
    if fp is not None and fp.intact():
        return fp
    fp = FlatParams(module)
    for m in module.modules():
        object.__setattr__(m, "_melgpt_flat", fp)
    return fp
