"""The optimiser of the tracking loop with its step as one kernel launch.

The reference builds ``torch.optim.Adam`` with one parameter group per tensor (/root/reference/src/tracking/train_utils.py:152-164);
torch then dispatches every group separately: ~0.2 ms of host time per iteration for eight small tensors, next to ~0.4 ms for the
whole loss step.  ``FusedAdam`` is a ``torch.optim.Adam`` subclass -- same constructor, same ``param_groups`` / ``state`` layout
(``step``, ``exp_avg``, ``exp_avg_sq``: the density control of ``gsdyn/densify.py`` edits them exactly as the reference edits
torch's) -- whose ``step()`` hands all tensors to ``gsr_adam_step`` (gsr_step.hip) in one call.  Anything the kernel does not
cover (weight decay, amsgrad, maximize, capturable / differentiable, sparse or non-HIP tensors) goes through ``super().step()``.
"""
from __future__ import annotations

import torch


class FusedAdam(torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:        # first: the gradients gathered below are the ones the closure produces
            with torch.enable_grad():
                loss = closure()
        entries = []
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad") or group.get("maximize") or group.get("capturable") \
                    or group.get("differentiable"):
                return self._fallback(loss)
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or g.is_sparse or not p.is_contiguous():
                    return self._fallback(loss)
                entries.append((group, p, g if g.is_contiguous() else g.contiguous(), float(beta1), float(beta2)))
        from diff_gaussian_rasterization import _hip
        packed = []
        for group, p, g, beta1, beta2 in entries:
            state = self.state[p]
            if len(state) == 0:       # as torch initialises it (a host scalar tensor for the count)
                state["step"] = torch.tensor(0.0, dtype=torch.float32)
                state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            state["step"] += 1
            m, v = state["exp_avg"], state["exp_avg_sq"]
            if not (m.is_contiguous() and v.is_contiguous()):
                state["exp_avg"], state["exp_avg_sq"] = m, v = m.contiguous(), v.contiguous()
            packed.append((p, g, m, v, float(group["lr"]), beta1, beta2, float(group["eps"]), float(state["step"])))
        _hip.adam_step(packed)
        # the kernel wrote the parameters behind autograd's back: bump their version counters so that caches keyed on
        # (data_ptr, _version) -- the per-view colour stack of gsdyn.step, converted settings tensors -- see the update
        for p, *_ in packed:
            torch.autograd.graph.increment_version(p)
        return loss

    def _fallback(self, loss):
        """torch's own step for what the kernel does not cover (the closure, if any, has already run)."""
        super().step()
        return loss
