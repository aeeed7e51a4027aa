"""The optimiser step of the reference's training loop -- `clip_grad_norm_(params, GRAD_NORM_CLIP)` followed by AdamW
(`tools/train_utils/train_utils.py:40-47`, `tools/train_utils/optimization/__init__.py:24-26`) -- as ONE call on cached
lists.

`torch.nn.utils.clip_grad_norm_` + `torch.optim.AdamW(fused=True).step()` are already multi-tensor on the device (two sets of
~14 launches for the detector's 432 parameters), but each call re-derives its tensor lists in Python -- grouping by device
and dtype, walking the param groups, checking every state entry -- 2.7 ms of host time per step on a step that is bound by the
host.  State layout and `state_dict()` are torch's.  The update itself:

* default (`CG3D_FUSED_ADAMW=1`, device library bound, fp32 contiguous tensors, equal hyper-parameters and equal step
  counters over all parameters): ONE launch of the library's `cg3d_adamw_step` (csrc/optim.hip) -- gradient x clip
  coefficient and the AdamW update in the arithmetic of torch's fused kernel (bit-identical parameters, tested).  Unlike
  `clip_grad_norm_` it does NOT write the clipped gradients back to `p.grad` (nothing reads them: `zero_grad` follows);
* otherwise: torch's own kernels (`_foreach_norm`, `_foreach_mul_`, `_foreach_add_`, `_fused_adamw_`) on lists built once,
  or plain `clip_grad_norm_` + `step()` when a gradient is missing or the state does not exist yet."""
import os
from ctypes import c_float, c_int64

import numpy as np
import torch

CHUNK = 1 << 15          # elements per table row of the fused step (one 256-thread workgroup each)
FUSED_NORM = os.environ.get("CG3D_FUSED_NORM", "1") != "0"      # gradient norm + clip coefficient by cg3d_grad_norm_clip
FUSED_STEP = os.environ.get("CG3D_FUSED_ADAMW", "1") != "0"


class ClippedAdamW(torch.optim.AdamW):
    def __init__(self, params, **kw):
        kw.setdefault("fused", True)
        super().__init__(params, **kw)
        self._lean = None
        self._early = None          # ids of the parameters the next forward reads first (set_early)
        self._late_hold = None      # what the late rows of the last step read (kept until the next step: see _fused_step)

    def set_early(self, params):
        """Split the fused update in two: the rows of `params` (everything the next step reads in its device-bound half) now, on
        the current stream; the rest -- 107 of the detector's 126.5 M parameters sit in the class branches, first read behind the
        dense head's first blocking read -- deferred to `me.run_late()` (me.LATE_MODE "defer": launched by the head after that
        read, on the current stream, in front of the conversion of the same weights), or on `me.late_stream()` ("stream").  The
        caller owns the consequence: after clip_and_step() the late parameters are final only once `finish_late()` has run (the
        next detector forward does by itself, `state_dict()` too).  None: one launch for all rows, as before."""
        self._early = None if params is None else {id(p) for p in params}
        self._plan = None

    def finish_late(self):
        """The current stream waits for the late rows of the last step (call before reading parameters outside a detector
        forward: checkpoints, evaluation of another module, parameter statistics)."""
        from . import me
        if self._late_hold is not None:
            me.run_late()
            self._late_hold = None

    def state_dict(self):
        self.finish_late()
        return super().state_dict()

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
        if norm_type == 2.0 and FUSED_NORM:
            total = self._fused_step(lean, flat, None, float(max_norm))      # norm, clip coefficient and update: three launches
            if total is not None:
                return total
        norms = torch._foreach_norm(flat, norm_type)
        total = torch.linalg.vector_norm(torch.stack(norms), norm_type)
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)      # clip_grad_norm_: always multiplied, 1.0 when below the bar
        if self._fused_step(lean, flat, coef) is not None:
            return total
        torch._foreach_mul_(flat, coef)
        for (group, ps, m1, m2, steps), gs in zip(lean, grads):
            beta1, beta2 = group["betas"]
            torch._foreach_add_(steps, 1)
            torch._fused_adamw_(ps, gs, m1, m2, [], steps, amsgrad=False, lr=group["lr"], beta1=beta1, beta2=beta2,
                                weight_decay=group["weight_decay"], eps=group["eps"], maximize=False, grad_scale=None,
                                found_inf=None)
        return total

    # -- one launch for the gradient scaling and the update of every parameter (cg3d_adamw_step, csrc/optim.hip)
    def _fused_step(self, lean, flat_grads, coef, max_norm=None):
        """The total gradient norm (a device scalar; `coef` given: True) if the step was taken by the library's kernels, else
        None: device library bound, every tensor fp32 and contiguous.  coef None: the norm and the clip coefficient come from
        cg3d_grad_norm_clip over the same chunk table (one pass over the gradients instead of torch's _foreach_norm chain).
        Unlike the torch path the gradients are NOT overwritten with their clipped values (nothing reads them afterwards:
        `zero_grad` follows); the parameters, moments and step counters end up as torch's kernel leaves them (tested)."""
        if not FUSED_STEP:
            return None
        from . import _lib, me
        lib = _lib.get()
        dev = flat_grads[0].device
        if not lib.is_device or dev.type != "cuda":
            return None
        plan = getattr(self, "_plan", None)
        if plan not in (None, False):
            # the table holds raw addresses: rebuild it when a parameter or a moment has moved (model.to(), .data reassigned,
            # state reloaded into new tensors) -- 432 integer compares, no device work
            cur = [t.data_ptr() for _, ps, m1, m2, _ in lean for ts in (ps, m1, m2) for t in ts]
            if cur != plan[4]:
                plan = self._plan = None
        if plan is None:
            rows, pid, k = [], [], 0
            late_rows, late_pid = [], []         # the late parameters' rows trail the table (pid keeps the parameter's place in `grads`)
            for group, ps, m1, m2, steps in lean:
                for p, a, b in zip(ps, m1, m2):
                    if not (p.dtype == a.dtype == b.dtype == torch.float32 and p.is_contiguous() and a.is_contiguous() and b.is_contiguous()):
                        self._plan = False
                        return None
                    n = p.numel()
                    late = self._early is not None and id(p) not in self._early
                    for o in range(0, n, CHUNK):
                        (late_rows if late else rows).append((p.data_ptr(), a.data_ptr(), b.data_ptr(), o, min(CHUNK, n - o)))
                        (late_pid if late else pid).append(k)
                    k += 1
            split = len(rows) if (late_rows and rows) else 0
            rows, pid = rows + late_rows, pid + late_pid
            plan = self._plan = (me.h2d(np.asarray(rows, dtype=np.int64), torch.int64, dev), me.h2d(np.asarray(pid, dtype=np.int32), torch.int32, dev),
                                 len(rows), [g["params"] for g in self.param_groups],
                                 [t.data_ptr() for _, ps, m1, m2, _ in lean for ts in (ps, m1, m2) for t in ts],
                                 split if (split > 0 and me.LATE_WEIGHTS) else 0)
        if plan is False:
            return None
        if any(g.dtype != torch.float32 or not g.is_contiguous() for g in flat_grads):
            return None
        # hyper-parameters are per group in torch; the table is one launch: require them equal (they are for this model)
        g0 = lean[0][0]
        if any((g["lr"], g["betas"], g["eps"], g["weight_decay"]) != (g0["lr"], g0["betas"], g0["eps"], g0["weight_decay"]) for g, *_ in lean):
            return None
        steps = [s for *_, ss in lean for s in ss]
        t = self._host_step = getattr(self, "_host_step", None) or 0
        if t == 0:
            # first fused step after torch-managed ones / a reload: ONE host read.  The launch applies a single bias correction
            # to every parameter, torch keeps a step counter per parameter: they must all agree (a parameter that joined late,
            # a state loaded with unequal steps, a step() that skipped gradient-less parameters) or torch's kernels take over
            st = torch.stack([x.reshape(()) for x in steps]).cpu()
            if not bool((st == st[0]).all()):
                self._host_step = None
                return None
            t = int(st[0].item())
        torch._foreach_add_(steps, 1)
        t += 1
        self._host_step = t
        beta1, beta2 = g0["betas"]
        gp = me.h2d(np.fromiter((g.data_ptr() for g in flat_grads), dtype=np.int64, count=len(flat_grads)), torch.int64, dev)
        total = True
        if coef is None:
            sc = torch.empty(4, dtype=torch.float64, device=dev)            # [sum of squares | norm, coefficient as floats]
            nc = sc[1:2].view(torch.float32)
            lib.call("cg3d_grad_norm_clip", plan[0].data_ptr(), plan[1].data_ptr(), c_int64(plan[2]), gp.data_ptr(), c_float(max_norm),
                     sc.data_ptr(), nc[0:1].data_ptr(), nc[1:2].data_ptr(), lib.stream())
            total, coef = nc[0], nc[1:2]
        hyper = (c_float(g0["lr"]), c_float(beta1), c_float(beta2), c_float(g0["eps"]), c_float(g0["weight_decay"]),
                 c_float(1.0 - beta1 ** t), c_float(1.0 - beta2 ** t))
        ne = plan[5]
        self.finish_late()                  # (a previous step's late rows nobody asked for yet: they read the buffers we are about to drop)
        if ne:
            # early rows here; the late rows behind the norm (= behind every gradient) but NOT now: deferred to me.run_late()
            # (the dense head calls it when the device-bound half of the next step is over), or -- stream mode -- on the late
            # stream.  What the late rows read -- the gradients (zero_grad(set_to_none=True) drops the parameters' references right
            # after this call, and the allocator would hand their memory to the next allocation on THIS stream), the pointer
            # table, the coefficient -- stays referenced until they have been launched and the next step begins
            lib.call("cg3d_adamw_step", plan[0].data_ptr(), plan[1].data_ptr(), c_int64(ne), gp.data_ptr(), coef.data_ptr(), *hyper, lib.stream())
            late = (plan[0].data_ptr() + ne * 5 * 8, plan[1].data_ptr() + ne * 4, c_int64(plan[2] - ne), gp.data_ptr(), coef.data_ptr()) + hyper
            hold = (dev, flat_grads, gp, coef, total, plan)
            if me.LATE_MODE == "defer":
                me.defer(lambda: (hold, lib.call("cg3d_adamw_step", *late, lib.stream())))
            else:
                ls = me.late_stream(dev)
                ls.wait_stream(torch.cuda.current_stream())
                lib.call("cg3d_adamw_step", *late, ls.cuda_stream)
                me.late_mark(dev)
            self._late_hold = hold
        else:
            lib.call("cg3d_adamw_step", plan[0].data_ptr(), plan[1].data_ptr(), c_int64(plan[2]), gp.data_ptr(), coef.data_ptr(), *hyper, lib.stream())
        return total

    def step(self, closure=None):
        self.finish_late()
        self._host_step = None            # torch advances the step counters itself here: re-read them at the next fused step
        return super().step(closure)

    def load_state_dict(self, state_dict):
        self.finish_late()
        super().load_state_dict(state_dict)
        self._lean = None
        self._plan = None
        self._host_step = None

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._lean = None
        self._plan = None
