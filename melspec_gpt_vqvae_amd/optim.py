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
        # parameters that got NO gradient this step (p.grad is None: the class embedder when no class token is fed, an
        # unused branch): torch.optim.AdamW - the reference - skips them entirely, no weight decay, moments untouched.
        # The fused launches run over the whole buffer, so their slices are zeroed (never apply a stale gradient) and
        # parameter + moments are put back afterwards.  Rare and small; the common step takes neither branch.
        missing = fp.zero_missing_grads()
        keep = [(sl, fp.data[sl].clone(), self._m[sl].clone(), self._v[sl].clone())
                for sl in (self._slice_of(fp, p) for p in missing)]
        self.step_count += 1
        nd = fp.n_decay
        shadow = fp.shadow
        for lo, hi, wd in ((0, nd, self.weight_decay), (nd, fp.total, 0.0)):
            if hi > lo:
                ops.adamw(fp.data[lo:hi], fp.grad[lo:hi], self._m[lo:hi], self._v[lo:hi], lr=self.lr, betas=self.betas,
                          eps=self.eps, weight_decay=wd, step=self.step_count,
                          param_bf16=None if shadow is None else shadow[lo:hi], grad_scale=self.grad_scale)
        for sl, d, m, v in keep:
            fp.data[sl].copy_(d)
            self._m[sl].copy_(m)
            self._v[sl].copy_(v)
            if shadow is not None:
                ops.cast(fp.data[sl], shadow.dtype, out=shadow[sl])
        # parameters were updated through the flat buffer (their tensor version counters are unchanged): the store's
        # generation tells every cache derived from a parameter; the kernel wrote the bf16 shadow itself, so that is fresh
        fp.generation += 1
        if shadow is not None:
            fp.mark_shadow_fresh()

    @staticmethod
    def _slice_of(fp, p):
        o = fp.offsets[fp._index[id(p)]]
        return slice(o, o + (p.numel() + 7) // 8 * 8)

    # ------------------------------------------------------------------------------------------ checkpoint layout
    def _ordered(self):
        """(name, parameter) in torch.optim.AdamW's index order for the reference's two groups: sorted decay names, then
        sorted no-decay names (minGPT.py:660-664, Lit_GPT_VAE.py:937-941) - names relative to `self.module`."""
        from .transformer.minGPT import decay_groups

        decay, no_decay = decay_groups(self.module)
        pd = dict(self.module.named_parameters())
        return [(n, pd[n]) for n in decay], [(n, pd[n]) for n in no_decay]

    def state_dict(self):
        """torch.optim.AdamW's state_dict layout (what a Lightning checkpoint of the reference holds under
        `optimizer_states`): per-parameter `step` / `exp_avg` / `exp_avg_sq` indexed in the reference's group order, two
        param_groups; `param_names` (extra key) spells the index -> name map out."""
        fp = self._state()
        dec, nod = self._ordered()
        state, names = {}, []
        for i, (n, p) in enumerate(dec + nod):
            o = fp.offsets[fp._index[id(p)]]
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self._m[o:o + p.numel()].view(p.shape).clone(),
                        "exp_avg_sq": self._v[o:o + p.numel()].view(p.shape).clone()}
            names.append(n)
        base = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "amsgrad": False, "maximize": False,
                "foreach": None, "capturable": False, "differentiable": False, "fused": None}
        groups = [dict(base, weight_decay=self.weight_decay, params=list(range(len(dec)))),
                  dict(base, weight_decay=0.0, params=list(range(len(dec), len(dec) + len(nod))))]
        return {"state": state, "param_groups": groups, "param_names": names}

    def load_state_dict(self, sd):
        """accepts torch.optim.AdamW's layout (a reference / Lightning checkpoint, or state_dict() above) and the flat
        layout this class wrote before (`step` / `exp_avg` / `exp_avg_sq` over the whole buffer)."""
        fp = self._state()
        if "state" in sd and "param_groups" in sd:
            dec, nod = self._ordered()
            order = dec + nod
            groups = sd["param_groups"]
            n_saved = sum(len(g["params"]) for g in groups)
            if n_saved != len(order):
                raise RuntimeError(f"optimizer state holds {n_saved} parameters, this model has {len(order)}")
            if "param_names" in sd and list(sd["param_names"]) != [n for n, _ in order]:
                raise RuntimeError("optimizer state was saved for a different parameter list")
            ids = [i for g in groups for i in g["params"]]
            steps = []
            self._m.zero_()
            self._v.zero_()
            for pid, (name, p) in zip(ids, order):
                st = sd["state"].get(pid, sd["state"].get(str(pid)))
                if st is None:      # torch keeps no entry for a parameter that never received a gradient
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise RuntimeError(f"optimizer state of {name}: shape {tuple(st['exp_avg'].shape)} != {tuple(p.shape)}")
                o = fp.offsets[fp._index[id(p)]]
                self._m[o:o + p.numel()].view(p.shape).copy_(st["exp_avg"])
                self._v[o:o + p.numel()].view(p.shape).copy_(st["exp_avg_sq"])
                steps.append(int(float(st["step"])))
            self.step_count = max(steps) if steps else 0
            g0 = groups[0]
            self.lr, self.betas, self.eps = g0["lr"], tuple(g0["betas"]), g0["eps"]
            self.weight_decay = g0.get("weight_decay", self.weight_decay)
            return
        if "exp_avg" not in sd or "step" not in sd:
            raise RuntimeError("unknown optimizer state layout (expected torch.optim.AdamW's or FusedAdamW's flat one)")
        if sd["exp_avg"].numel() != self._m.numel():
            raise RuntimeError(f"flat optimizer state of {sd['exp_avg'].numel()} elements does not fit this model's "
                               f"{self._m.numel()} (saved at another world size / module tree?)")
        self.step_count = int(sd["step"])
        self._m.copy_(sd["exp_avg"])
        self._v.copy_(sd["exp_avg_sq"])
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in sd:
                setattr(self, k, tuple(sd[k]) if k == "betas" else sd[k])
