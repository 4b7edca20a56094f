"""Data-parallel gradient exchange for one 8 x MI355X node: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm).

The reference trains with Lightning's stock DDP (GPT_VAE_train.py:172-174): parameters replicated, batch sharded
by DistributedSampler, gradients averaged.  Here the gradients already live in ONE contiguous f32 buffer
(flat.FlatParams), laid out so that each transformer block owns two contiguous slices (its Linear weights and
its biases/LayerNorm), and the backward pass announces a block the moment its gradients are final.  So the
exchange is a handful of LARGE all-reduces (~50 MB per block for the VAS model, ~105 MB for GPT-VAE XL) issued
while the earlier blocks are still in their backward GEMMs - sized for 7 x ~153 GB/s point-to-point links rather
than for NVSwitch-style many-small-bucket traffic; no parameter broadcast per step (the `mask` buffers the
reference re-broadcasts every forward do not exist here) and averaging is folded into the optimizer's
grad_scale.  The engine only needs a flat gradient tensor and (lo, hi) segments, so it is exercised on CPU with
the gloo backend in tests/test_dp_cpu.py.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradientExchange:
    """All-reduce (sum) of `grad` in segments; segments may be launched early (overlap) and in any order."""

    def __init__(self, grad: torch.Tensor, group=None, max_bucket_elems: int = 64 << 20):
        assert grad.dim() == 1
        self.grad = grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.max_bucket = int(max_bucket_elems)
        self._works = []
        self._done = []  # (lo, hi) already launched this step

    def launch(self, lo: int, hi: int):
        """start reducing grad[lo:hi] (asynchronously on GPU backends); safe to call from a backward hook."""
        if self.world == 1 or hi <= lo:
            return
        self._done.append((lo, hi))
        pos = lo
        while pos < hi:
            end = min(hi, pos + self.max_bucket)
            self._works.append(dist.all_reduce(self.grad[pos:end], op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))
            pos = end

    def finish(self):
        """reduce everything not launched yet, then make the current stream wait for all of it."""
        if self.world > 1:
            covered = sorted(self._done)
            pos = 0
            for lo, hi in covered:
                assert lo >= pos, "overlapping gradient segments"
                self.launch_uncounted(pos, lo)
                pos = hi
            self.launch_uncounted(pos, self.grad.numel())
            for w in self._works:
                w.wait()
        self._works, self._done = [], []

    def launch_uncounted(self, lo, hi):
        if hi > lo:
            pos = lo
            while pos < hi:
                end = min(hi, pos + self.max_bucket)
                self._works.append(dist.all_reduce(self.grad[pos:end], op=dist.ReduceOp.SUM, group=self.group,
                                                   async_op=True))
                pos = end


def block_segments(fp, block):
    """the (lo, hi) slices of fp.grad owned by one transformer Block: [Linear weights], [biases + LayerNorm]."""
    idx = sorted(fp._index[id(p)] for p in block.parameters())
    segs = []
    start = prev = None
    for i in idx:
        if start is None:
            start = prev = i
        elif i == prev + 1:
            prev = i
        else:
            segs.append((start, prev))
            start = prev = i
    segs.append((start, prev))
    out = []
    for a, b in segs:
        lo = fp.offsets[a]
        hi = fp.offsets[b] + (fp.params[b].numel() + 7) // 8 * 8
        out.append((lo, min(hi, fp.total)))
    return out


class DataParallel:
    """Wraps a GPT-like module: hooks every Block's end-of-backward to start its gradient all-reduce, and
    exposes finish() to be called once per step before the optimizer (grad_scale = 1/world there)."""

    def __init__(self, module, group=None):
        from .flat import ensure_flat

        self.module = module
        self.fp = ensure_flat(module)
        self.ex = GradientExchange(self.fp.grad, group)
        self.world = self.ex.world
        blocks = getattr(module, "blocks", None)
        if blocks is None and hasattr(module, "transformer"):
            blocks = getattr(module.transformer, "blocks", None)
        self._segs = {}
        if blocks is not None and self.world > 1:
            for blk in blocks:
                self._segs[id(blk)] = block_segments(self.fp, blk)
                object.__setattr__(blk, "_grad_ready_hook", self._on_block_done)

    def _on_block_done(self, blk):
        for lo, hi in self._segs[id(blk)]:
            self.ex.launch(lo, hi)

    def finish(self):
        self.ex.finish()

    def broadcast_parameters(self, src=0):
        """C1 of SURVEY §2b: one broadcast of the flat parameter buffer at start-up (skipped when every rank
        seeds identically)."""
        if self.world > 1:
            dist.broadcast(self.fp.data, src=src, group=self.ex.group)


def distributed_shard(n: int, rank: int, world: int, seed: int = 0, epoch: int = 0, shuffle: bool = True,
                      drop_last: bool = False):
    """The index partition Lightning's DDP gives the reference (torch DistributedSampler rule):
    a seeded permutation, padded (or truncated) to a multiple of `world`, rank r takes indices r::world."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        indices = torch.randperm(n, generator=g).tolist()
    else:
        indices = list(range(n))
    if drop_last and n % world != 0:
        total = (n // world) * world
        indices = indices[:total]
    else:
        total = -(-n // world) * world
        pad = total - len(indices)
        if pad > 0:
            indices += (indices * (-(-pad // max(len(indices), 1))))[:pad]
    return indices[rank:total:world]
