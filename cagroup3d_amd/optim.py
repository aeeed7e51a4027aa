"""The optimiser step of the reference's training loop -- `clip_grad_norm_(params, GRAD_NORM_CLIP)` followed by AdamW
(`tools/train_utils/train_utils.py:40-47`, `tools/train_utils/optimization/__init__.py:24-26`) -- as ONE call on cached
lists.

`torch.nn.utils.clip_grad_norm_` + `torch.optim.AdamW(fused=True).step()` are already multi-tensor on the device (two sets of
~14 launches for the detector's 432 parameters), but each call re-derives its tensor lists in Python -- grouping by device
and dtype, walking the param groups, checking every state entry -- 2.7 ms of host time per step on a step that is bound by the
host.  This subclass runs the SAME torch kernels (`_foreach_norm`, `_foreach_mul_`, `_foreach_add_`,
`_fused_adamw_`) on lists built once; state layout, `state_dict()` and the update itself are torch's."""
import torch


class ClippedAdamW(torch.optim.AdamW):
    def __init__(self, params, **kw):
        kw.setdefault("fused", True)
        super().__init__(params, **kw)
        self._lean = None

    def _lists(self):
        if self._lean is None:
            lean = []
            for group in self.param_groups:
                ps = [p for p in group["params"] if p.requires_grad]
                if any(len(self.state.get(p, ())) == 0 for p in ps):
                    return None                                   # state not created yet: the first step goes through torch
                lean.append((group, ps, [self.state[p]["exp_avg"] for p in ps], [self.state[p]["exp_avg_sq"] for p in ps],
                             [self.state[p]["step"] for p in ps]))
            self._lean = lean
        return self._lean

    @torch.no_grad()
    def clip_and_step(self, max_norm, norm_type=2.0):
        """clip_grad_norm_(all parameters, max_norm) + step(); returns the total gradient norm (a device scalar)."""
        lean = self._lists()
        if lean is None or any(p.grad is None for _, ps, _, _, _ in lean for p in ps):
            params = [p for g in self.param_groups for p in g["params"]]
            total = torch.nn.utils.clip_grad_norm_(params, max_norm, norm_type)
            self.step()
            return total
        grads = [[p.grad for p in ps] for _, ps, _, _, _ in lean]
        flat = [g for gs in grads for g in gs]
        norms = torch._foreach_norm(flat, norm_type)
        total = torch.linalg.vector_norm(torch.stack(norms), norm_type)
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)      # clip_grad_norm_: always multiplied, 1.0 when below the bar
        torch._foreach_mul_(flat, coef)
        for (group, ps, m1, m2, steps), gs in zip(lean, grads):
            beta1, beta2 = group["betas"]
            torch._foreach_add_(steps, 1)
            torch._fused_adamw_(ps, gs, m1, m2, [], steps, amsgrad=False, lr=group["lr"], beta1=beta1, beta2=beta2,
                                weight_decay=group["weight_decay"], eps=group["eps"], maximize=False, grad_scale=None,
                                found_inf=None)
        return total

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._lean = None

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._lean = None
