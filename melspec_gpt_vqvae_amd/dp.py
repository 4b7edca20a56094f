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
the gloo backend in tests/test_host_cpu.py; the DataParallel wrapper itself (hooks + segments) runs with two ranks
sharing one GPU in tests/test_dp_gpu.py.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


RCCL_CHANNELS_DEFAULT = 0      # 0: RCCL picks its own channel count and no CU is reserved (see pin_rccl_channels)


def pin_rccl_channels(n=None):
    """OPT-IN pairing for real multi-GPU nodes: bound RCCL's footprint BEFORE the process group is created
    (NCCL_MAX_NCHANNELS = n: one persistent workgroup = one CU per channel) so that DataParallel can RESERVE exactly that
    many CUs for it while an exchange can be in flight.  Call from every rank before dist.init_process_group("nccl");
    n = None reads MELGPT_RCCL_CHANNELS (default 0 = leave RCCL alone, reserve nothing); a value the user already exported
    (NCCL_MAX_NCHANNELS) is respected and returned.
    Why it exists, and why it is not the default (tools/lab/overlap_lab.hip beside a stand-in collective,
    profiles/r05_overlap_lab_*.jsonl; one rank through a real RCCL group, profiles/r05_c_bench_dp.json): the persistent
    GEMM / conv grids own one CU per workgroup and walk static tile lists, so a CU held by an all-reduce kernel costs the
    launch beside it one workgroup's WHOLE list: the backward MLP chain of a Block 0.94-0.96 -> 1.35-1.43 ms while 16 or 32
    channels are resident (tiles claimed from a counter on the ring loop, removed in round 6: 1.42-1.46 - no better), but 0.98 -> 1.06 when as many CUs are
    reserved as the collective has channels (and 1.73 when it has twice as many: the count must be KNOWN, hence the pin).
    The reservation is paid all the time, though: the split-K weight gradients lose their exact one-round fit (256 tiles on
    256 CUs -> 10 batches on 240), +5.7 % on the whole step with nothing resident (94.4 -> 99.7 ms), against +35-47 % on the
    backward GEMMs only WHILE a collective is resident - the exchange is 1.2 GB per ~55 ms backward pass (8.4 GB per ~300 ms
    for GPT-VAE XL), i.e. resident for 15-25 % of it at 100-150 GB/s, which prices both choices within a millisecond of each
    other.  Without a node to measure on, the default is the one whose cost is zero when RCCL is idle."""
    if n is None:
        n = int(os.environ.get("MELGPT_RCCL_CHANNELS", str(RCCL_CHANNELS_DEFAULT)))
    if n > 0:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(n))
        if int(os.environ.get("NCCL_MIN_NCHANNELS", "1")) > int(os.environ["NCCL_MAX_NCHANNELS"]):
            os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"]
        os.environ.setdefault("MELGPT_RESERVE_CUS", str((pinned_rccl_channels() + 7) // 8 * 8))
    return pinned_rccl_channels()


def pinned_rccl_channels():
    """the channel bound RCCL was started under (NCCL_MAX_NCHANNELS), 0 = unknown / unbounded"""
    try:
        return max(0, int(os.environ.get("NCCL_MAX_NCHANNELS", "0")))
    except ValueError:
        return 0


def _convert(src, dst):
    """dtype-converting copy dst <- src: the library's cast kernel on the GPU, a torch copy on the CPU rig of the tests."""
    if src.is_cuda:
        from . import ops

        ops.cast(src, dst.dtype, out=dst)
    else:
        dst.copy_(src)


class GradientExchange:
    """All-reduce (sum) of `grad` in segments; segments may be launched early (overlap) and in any order.

    wire_dtype (default None = the gradient's own f32): a 16-bit format to exchange in - SURVEY 2b C2, for the GPT-VAE XL
    job whose f32 gradient is 8.37 GB per step: a launched slice is cast into a staging buffer of that format, the
    staging slice is all-reduced, and finish() converts the sums back into the f32 buffer (moments and master weights
    stay f32).  The reduction itself then runs in 16 bits: ~2^-9 relative per addition for bf16; tested on two ranks at
    6e-3 of the gradient's maximum / 3e-3 rms (tests/test_host_cpu.py: 1e-3 would be below ONE bf16 rounding), off by
    default.  IEEE half is refused as a wire format by DataParallel: loss-scaled gradients summed over ranks in a 5-bit
    exponent overflow, the step is skipped and the scale drops - every step."""

    def __init__(self, grad: torch.Tensor, group=None, max_bucket_elems: int = 64 << 20, wire_dtype=None):
        assert grad.dim() == 1
        self.grad = grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a group of ONE rank has nothing to exchange; MELGPT_DP_FORCE_EXCHANGE=1 (debug aid, 1-GPU boxes) launches the
        # all-reduces anyway, so that a single rank drives RCCL exactly as a rank of an N-GPU run does
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("MELGPT_DP_FORCE_EXCHANGE") == "1")
        self.max_bucket = int(max_bucket_elems)
        self.wire_dtype = None if wire_dtype in (None, torch.float32) else wire_dtype
        self._wire = None             # staging buffer in wire_dtype (allocated on first use)
        self._works = []
        self._done = []  # (lo, hi) already launched this step
        self._wired = []              # (lo, hi) whose sums sit in the staging buffer until finish()
        self.last_wait_ms = 0.0       # host time finish() spent in the waits (GPU backends: the time to ENQUEUE them)
        self.time_events = False      # GPU: bracket finish()'s waits with stream events -> exposed_ms()
        self._events = []
        self._ev0 = None              # mark_backward_end(): the event behind the backward pass' last kernel

    @property
    def bytes_per_step(self):
        return self.grad.numel() * (4 if self.wire_dtype is None else 2)

    def _reduce(self, lo, hi):
        buf = self.grad
        if self.wire_dtype is not None:
            if self._wire is None:
                self._wire = torch.empty(self.grad.numel(), dtype=self.wire_dtype, device=self.grad.device)
            _convert(self.grad[lo:hi], self._wire[lo:hi])
            self._wired.append((lo, hi))
            buf = self._wire
        pos = lo
        while pos < hi:
            end = min(hi, pos + self.max_bucket)
            self._works.append(dist.all_reduce(buf[pos:end], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            pos = end

    def launch(self, lo: int, hi: int):
        """start reducing grad[lo:hi] (asynchronously on GPU backends); safe to call from a backward hook."""
        if not self.active or hi <= lo:
            return
        if (lo, hi) in self._done:
            # a block's backward ran twice before finish() (gradient accumulation / two backward passes): the first
            # all-reduce already summed a PARTIAL gradient across ranks - refuse instead of producing wrong sums
            raise RuntimeError(f"gradient segment [{lo}, {hi}) was launched twice in one step: call finish() once per "
                               "backward pass, or disable early launches (DataParallel(..., overlap=False)) when "
                               "accumulating gradients")
        self._done.append((lo, hi))
        self._reduce(lo, hi)

    def mark_backward_end(self):
        """with time_events: record the start of the exposed window NOW (the caller is about to queue stream work that
        waits for the collectives - DataParallel's has-gradient mask all-reduce shares the communicator with every
        early-launched gradient all-reduce, so an event recorded after it would already sit behind the exposed wait)."""
        if self.active and self.time_events and self.grad.is_cuda and self._ev0 is None:
            self._ev0 = torch.cuda.Event(enable_timing=True)
            self._ev0.record()

    def finish(self):
        """reduce everything not launched yet, then make the current stream wait for all of it."""
        if self.active:
            import time

            covered = sorted(self._done)
            pos = 0
            for lo, hi in covered:
                assert lo >= pos, "overlapping gradient segments"
                self.launch_uncounted(pos, lo)
                pos = hi
            self.launch_uncounted(pos, self.grad.numel())
            ev = None
            if self.time_events and self.grad.is_cuda:
                self.mark_backward_end()        # behind the last kernel of the backward pass (unless marked earlier)
                ev = (self._ev0, torch.cuda.Event(enable_timing=True))
            t0 = time.perf_counter()
            for w in self._works:
                w.wait()
            self.last_wait_ms = 1e3 * (time.perf_counter() - t0)
            for lo, hi in self._wired:          # the 16-bit sums back into the f32 gradient
                _convert(self._wire[lo:hi], self.grad[lo:hi])
            if ev is not None:
                ev[1].record()                  # behind the last collective (and the conversion back)
                self._events.append(ev)
        self._works, self._done, self._wired, self._ev0 = [], [], [], None

    def launch_uncounted(self, lo, hi):
        if hi > lo:
            self._reduce(lo, hi)

    def exposed_ms(self, reset=True):
        """with time_events: per finish() call, the GPU time between the end of the backward pass' last kernel and the
        end of the last all-reduce - the part of the exchange that was NOT hidden under the backward GEMMs.  Synchronises."""
        if not self._events:
            return []
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self._events]
        if reset:
            self._events = []
        return out


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
    """Wraps a module tree holding minGPT Blocks - a GPT / GPTClass, a Lit_minGPT, or a GPT_VAE with its TWO
    transformers (encoder.transformer + decoder.transformer, Lit_GPT_VAE.py:42-43; the only model the reference
    trains under DDP, GPT_VAE_train.py:172-174).  All parameters of the tree live in one flat buffer; every Block,
    wherever it sits in the tree, announces the end of its backward and its two gradient slices are all-reduced
    at once while earlier blocks are still in their GEMMs.  finish() is called once per step before the optimizer
    (whose grad_scale = 1/world folds the averaging in).

    Has-gradient verdict: a grad-set mismatch between ranks that first appears at step N > 1 is raised at step N + 1's
    finish() - by then optimizer step N has run on diverged replicas.  DRAIN THE VERDICT BEFORE SAVING: call `check()`
    (or `detach()`) before writing a checkpoint or after the last step of a loop; both wait for and raise the last one."""

    def __init__(self, module, group=None, overlap=True, max_bucket_elems: int = 64 << 20, reserve_cus=None,
                 grad_dtype=None):
        from . import _ffi
        from .flat import ensure_flat

        self.module = module
        self.fp = ensure_flat(module)
        if isinstance(grad_dtype, str):      # "f32" | "bf16", as the CLIs spell it
            grad_dtype = {"f32": None, "bf16": torch.bfloat16, "fp16": torch.float16, "half": _ffi.HALF_DTYPE}[grad_dtype]
        if grad_dtype not in (None, torch.float32):
            # the cast kernel converts to the LIBRARY's 16-bit format only, and IEEE half cannot carry loss-scaled
            # gradients summed over ranks (GradientExchange's docstring): a 16-bit wire exists in the bf16 flavour only
            if grad_dtype != torch.bfloat16 or _ffi.HALF_DTYPE != torch.bfloat16:
                raise ValueError(f"gradient wire format {grad_dtype} is not supported in the "
                                 f"{str(_ffi.HALF_DTYPE)[6:]} flavour of the library: use f32 (or bf16 in the bf16 flavour)")
        self.ex = GradientExchange(self.fp.grad, group, max_bucket_elems=max_bucket_elems, wire_dtype=grad_dtype)
        self.world = self.ex.world
        self.overlap = bool(overlap)
        # The persistent GEMM / conv kernels own one CU per workgroup for a whole launch; an RCCL kernel that holds a CU
        # when such a launch starts leaves one workgroup waiting for another one's ENTIRE static tile list.  One switch acts
        # while an (overlapped) all-reduce can be in flight - from the first Block's early launch to finish() - and it is
        # OFF by default (measurements: pin_rccl_channels' docstring): `reserve_cus` (MELGPT_RESERVE_CUS; set by
        # pin_rccl_channels together with RCCL's channel bound): the persistent grids leave that many CUs to RCCL - the
        # remedy that works (1.35-1.43 -> 1.06 ms per backward MLP chain beside a collective) when the collective's channel
        # count is known, at +5.7 % on the step when it is idle.  (Tiles claimed from a run-time counter - rounds 2-5 - ran on
        # the ring loop and were the worst cell of every column of profiles/r05_dp_lab.md: removed in round 6.)
        # The forward pass, the head's and the last Block's backward and the optimizer run on the whole chip.
        active = self.ex.active
        if reserve_cus is None:
            reserve_cus = int(os.environ.get("MELGPT_RESERVE_CUS", "0"))
        self.reserve_cus = int(reserve_cus)
        self._reserved = False
        self._on_gpu = self.fp.device.type == "cuda"
        if self._on_gpu:
            _ffi.call("melgpt_set_reserved_cus", 0)
        self.hook_calls = 0           # Block hooks fired since construction (tests count them)
        # which parameters got a gradient this step must be the SAME on every rank: FusedAdamW skips a parameter whose
        # .grad is None (torch.optim.AdamW's rule) and decides that per rank, so a parameter missing on one rank only
        # would be restored there and stepped elsewhere - silently diverging replicas (the reference's DDP raises).  The
        # has-gradient bitmask is all-reduced with the gradients; the first step is checked synchronously, step N > 1 at
        # step N + 1's finish() through a pinned host flag: ALWAYS exactly one step late (the flag's event is waited for,
        # never polled - it is a whole step old, so the wait is free), hence on the same step on every rank - the verdict
        # comes from the all-reduced mask, so either every rank raises there or none does; detach() reads the last one.
        self._mask_steps = 0
        self._mask_flag = None        # (pinned host tensor, event) of the previous step's check
        self._mask_host = None
        self._segs = {}
        self.blocks = [m for m in module.modules() if hasattr(m, "_layer_index") and hasattr(m, "attn")]
        if (active and self._on_gpu and self.world > 1 and self.reserve_cus == 0 and pinned_rccl_channels() == 0
                and (not dist.is_initialized() or dist.get_rank() == 0)):
            import sys

            print("melgpt dp: RCCL shares the chip with the persistent GEMM / conv grids and no CUs are reserved for it "
                  "(measured beside a stand-in collective: backward GEMMs 0.94 -> 1.35-1.43 ms while it is resident). "
                  "To pair a pinned channel count with as many reserved CUs: MELGPT_RCCL_CHANNELS=16 before init_process_group "
                  "(dp.pin_rccl_channels); costs +5.7 % on the step when RCCL is idle, so it is not the default.", file=sys.stderr)
        if active and self.overlap:
            for blk in self.blocks:
                self._segs[id(blk)] = block_segments(self.fp, blk)
                object.__setattr__(blk, "_grad_ready_hook", self._on_block_done)

    def _reserve(self, on):
        """enter / leave the part of a step in which RCCL kernels may share the chip with the compute kernels"""
        if self._on_gpu and on != self._reserved and self.reserve_cus > 0:
            from . import _ffi

            _ffi.call("melgpt_set_reserved_cus", self.reserve_cus if on else 0)
            self._reserved = on

    def _on_block_done(self, blk):
        self.hook_calls += 1
        self._reserve(True)   # the launches from here on share the chip with RCCL kernels
        for lo, hi in self._segs[id(blk)]:
            self.ex.launch(lo, hi)

    def finish(self):
        """all-reduce whatever the Block hooks have not launched (stem, head, ln_f, parameters without a gradient this
        step - zeroed first so that no stale slice is summed), then wait for every launched piece."""
        if self.ex.active:
            self.fp.zero_missing_grads()
        self.ex.mark_backward_end()     # exposed_comm_ms counts from here: the mask all-reduce below already waits
        self._check_grad_sets()
        self.ex.finish()
        self._reserve(False)  # everything launched after this is stream-ordered behind the last all-reduce

    def _raise_if_flagged(self):
        if self._mask_flag is None:
            return
        host, ev = self._mask_flag
        if ev is not None:
            ev.synchronize()
        self._mask_flag = None
        # the verdict is taken on the HOST from the all-reduced mask (a few hundred floats): no device kernel of torch's
        if bool(((host != 0) & (host != float(self.world))).any()):
            raise RuntimeError("data-parallel ranks disagree on which parameters received a gradient this step "
                               "(a parameter with .grad None on some ranks only): the replicas would diverge - "
                               "make the unused branch the same on every rank")

    def _check_grad_sets(self):
        """all-reduce the has-gradient bitmask of the flat store's parameters; see __init__.  On the GPU the step costs
        two small copies (pinned host -> device, device -> pinned host) around the all-reduce and NO torch kernel: the mask
        is written and compared on the host (tests/test_step_kernels_gpu.py runs the census through a 1-rank DataParallel
        with the exchange forced)."""
        if not self.ex.active:
            return
        self._raise_if_flagged()        # the previous step's verdict, before its slot is reused
        n = len(self.fp.params)
        has = torch.tensor([0.0 if p.grad is None else 1.0 for p in self.fp.params], dtype=torch.float32)
        if self._on_gpu:
            if self._mask_host is None:     # (allocated once: two slots, the previous step's verdict is read before reuse)
                self._mask_host = [(torch.zeros(n, dtype=torch.float32).pin_memory(), torch.zeros(n, dtype=torch.float32).pin_memory())
                                   for _ in range(2)]
                self._mask_dev = [torch.empty(n, dtype=torch.float32, device=self.fp.device) for _ in range(2)]
            src, dst = self._mask_host[self._mask_steps & 1]
            dev = self._mask_dev[self._mask_steps & 1]
            src.copy_(has)
            dev.copy_(src, non_blocking=True)
            dist.all_reduce(dev, op=dist.ReduceOp.SUM, group=self.ex.group)
            dst.copy_(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._mask_flag = (dst, ev)
        else:
            dist.all_reduce(has, op=dist.ReduceOp.SUM, group=self.ex.group)
            self._mask_flag = (has, None)
        self._mask_steps += 1
        if self._mask_steps == 1 or not self._on_gpu:
            self._raise_if_flagged()

    def describe(self):
        """what a bench line needs to explain a scaling record: the exchange's size and the two persistent-kernel switches"""
        return {"exchange_bytes": int(self.ex.bytes_per_step), "exchange_dtype": str(self.ex.wire_dtype or torch.float32)[6:],
                "dp_tiles": "static", "reserved_cus": int(self.reserve_cus),
                "rccl_channels_pinned": pinned_rccl_channels(),
                "overlap": bool(self.overlap), "backend": dist.get_backend(self.ex.group) if dist.is_initialized() else None}

    def check(self):
        """wait for and raise the last step's has-gradient verdict (call before saving a checkpoint / after the last step)"""
        self._raise_if_flagged()

    def detach(self):
        self._raise_if_flagged()
        for blk in self.blocks:
            if getattr(blk, "_grad_ready_hook", None) is not None:
                object.__setattr__(blk, "_grad_ready_hook", None)

    def broadcast_parameters(self, src=0):
        """C1 of SURVEY §2b: one broadcast of the flat parameter buffer at start-up (skipped when every rank
        seeds identically)."""
        if self.world > 1:
            dist.broadcast(self.fp.data, src=src, group=self.ex.group)
            self.fp.generation += 1         # a write through the flat buffer: the bf16 shadow and every cache derived
            self.fp._shadow_key = None      # from a parameter (flat.tensor_version) are rebuilt from the received copy

    def reduce_metrics(self, *values):
        """C4 of SURVEY §2b: the logged scalars of a step (Lit_GPT_VAE.py:310-313 logs loss / kl_weight / rec / KL with
        sync_dist=True = four separate all-reduces) as ONE all-reduce of a small f32 vector -> tuple of rank means
        (0-dim tensors on the values' device).  Accepts tensors or Python numbers."""
        dev = next((v.device for v in values if isinstance(v, torch.Tensor)), self.fp.device)
        vec = torch.stack([v.detach().float().reshape(()) if isinstance(v, torch.Tensor)
                           else torch.tensor(float(v), device=dev) for v in values]).to(dev)
        if self.world > 1:
            dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.ex.group)
            vec = vec / self.world
        return tuple(vec.unbind(0))


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
