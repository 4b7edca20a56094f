"""Fused AdamW over the flat parameter store (one launch per weight-decay group + bf16 shadow refresh).

Same update rule and grouping as the reference's configure_optimizers (transformer/minGPT.py:618-665,
Lit_GPT_VAE.py:895-943): weights of nn.Linear get weight_decay (default 0.01), every bias / LayerNorm /
Embedding weight / pos_emb gets 0; torch.optim.AdamW semantics with betas (0.9, 0.95)."""
from __future__ import annotations

import torch

from . import ops
from .flat import ensure_flat


class FusedAdamW:
    def __init__(self, module, lr=1e-6, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01):
        self.module = module
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.grad_scale = 1.0
        self._fp = None
        self._m = self._v = None

    def _state(self):
        fp = ensure_flat(self.module)
        if fp is not self._fp:
            self._fp = fp
            self._m = torch.zeros_like(fp.data)
            self._v = torch.zeros_like(fp.data)
        return fp

    @property
    def param_groups(self):
        return [{"lr": self.lr, "betas": self.betas, "weight_decay": self.weight_decay},
                {"lr": self.lr, "betas": self.betas, "weight_decay": 0.0}]

    def zero_grad(self, set_to_none=False):
        """Gradients live in one flat buffer that the backward pass overwrites (beta = 0) when `.grad is None`,
        so dropping the views is free and no memset is needed."""
        fp = self._state()
        for p in fp.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        fp = self._state()
        fp.zero_missing_grads()   # never apply a previous step's gradient to a parameter this step did not touch
        self.step_count += 1
        nd = fp.n_decay
        shadow = fp.shadow
        for lo, hi, wd in ((0, nd, self.weight_decay), (nd, fp.total, 0.0)):
            if hi > lo:
                ops.adamw(fp.data[lo:hi], fp.grad[lo:hi], self._m[lo:hi], self._v[lo:hi], lr=self.lr, betas=self.betas,
                          eps=self.eps, weight_decay=wd, step=self.step_count,
                          param_bf16=None if shadow is None else shadow[lo:hi], grad_scale=self.grad_scale)
        # parameters were updated through the flat buffer (their version counters are unchanged) and the kernel
        # wrote the bf16 shadow itself, so the shadow is already fresh
        if shadow is not None:
            fp.mark_shadow_fresh()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self._m, "exp_avg_sq": self._v, "lr": self.lr,
                "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self._state()
        self.step_count = int(sd["step"])
        self._m.copy_(sd["exp_avg"])
        self._v.copy_(sd["exp_avg_sq"])
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in sd:
                setattr(self, k, tuple(sd[k]) if k == "betas" else sd[k])
