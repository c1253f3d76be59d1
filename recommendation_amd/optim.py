"""Fused dense Adam on the embedding tables (the optimiser step on the other side of the hot path):
`torch.optim.Adam(model.parameters(), lr=...)` of ncl.py:305, lightgcn.py:84, gcl.py:201 (with weight_decay) as ONE
streaming HIP kernel per parameter (gcr_adam_step_f32): reads p, g, m, v and writes p, m, v once (28 B per element).
Same update rule and defaults as torch.optim.Adam (no amsgrad, L2 weight_decay folded into the gradient)."""
from __future__ import annotations

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        """capturable: keep the step count in device memory (gcr_adam_step_dev_f32), so that a step captured in a
        hipGraph replays with the current bias corrections (torch.optim.Adam's flag of the same name)."""
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=bool(capturable)))

    @torch.no_grad()
    def step(self, closure=None, extra_grads=None):
        """extra_grads: optional {param: [tensor, ...]} of up to two further gradient pieces per parameter that were
        NOT accumulated into p.grad; they are summed inside the kernel."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                _lib.require_cuda(p, p.grad)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise ValueError("FusedAdam needs contiguous float32 parameters")
                st = self.state[p]
                if not st:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("FusedAdam: the first step() must run eagerly (it allocates the state; inside a "
                                           "stream capture the allocation and the zero count would be replayed every time)")
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                    if group.get("capturable"):            # device-side step count, allocated with the rest of the state
                        st["step_dev"] = torch.zeros(1, dtype=torch.int64, device=p.device)
                st["step"] += 1
                extra = list((extra_grads or {}).get(p, []))
                if len(extra) > 2:
                    raise ValueError("at most two extra gradient pieces")
                g = p.grad.contiguous()
                ex = [e.contiguous() for e in extra] + [None, None]
                if group.get("capturable"):
                    if "step_dev" not in st:               # state loaded from a non-capturable run
                        if torch.cuda.is_current_stream_capturing():
                            raise RuntimeError("FusedAdam: no device step count yet; run one eager step() before capturing")
                        st["step_dev"] = torch.full((1,), st["step"] - 1, dtype=torch.int64, device=p.device)
                    st["step_dev"].add_(1)
                    _lib.check(L.gcr_adam_step_dev_f32(_lib.dptr(p), _lib.dptr(g), _lib.dptr(ex[0]), _lib.dptr(ex[1]),
                                                       _lib.dptr(st["exp_avg"]), _lib.dptr(st["exp_avg_sq"]), p.numel(),
                                                       float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                       float(group["weight_decay"]), _lib.dptr(st["step_dev"]), 1.0,
                                                       _lib.cur_stream(p.device)), "gcr_adam_step_dev_f32")
                    continue
                _lib.check(L.gcr_adam_step_f32(_lib.dptr(p), _lib.dptr(g), _lib.dptr(ex[0]), _lib.dptr(ex[1]),
                                               _lib.dptr(st["exp_avg"]), _lib.dptr(st["exp_avg_sq"]), p.numel(),
                                               float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                               float(group["weight_decay"]), int(st["step"]), 1.0, _lib.cur_stream(p.device)),
                           "gcr_adam_step_f32")
        return loss

    def sync_step_counts(self):
        """Host step counts <- device step counts (one read-back per capturable parameter): graph replays advance only the
        device counter.  `state_dict()` calls it, so that a checkpoint resumed on either path gets the right bias corrections."""
        for st in self.state.values():
            if "step_dev" in st:
                st["step"] = int(st["step_dev"].item())

    def state_dict(self):
        self.sync_step_counts()
        return super().state_dict()
