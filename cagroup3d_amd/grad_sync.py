"""Data-parallel gradient exchange for the detector (one all-reduce of 126.5 M gradients per step).

torch DDP costs this model ~14 ms per step on ONE GPU before any byte is exchanged (432 parameters: a Python-visible
autograd hook and a bucket-copy launch each) -- a third of the 48 ms step.  The exchange here is three flat buckets over
RCCL instead, split where the model splits:

  early bucket  dense head + RoI head parameters (57 % of the bytes: the 18 class branches).  Their gradients are
                complete when the backward pass reaches the backbone output; a hook on THAT tensor packs them with one
                multi-tensor copy and starts an asynchronous all-reduce which runs under the whole backbone backward.
  mid bucket    the backbone from layer3 on (the 256/512/1024-channel layers: nearly all of the backbone's bytes), sent
                from a hook on layer2's output -- the backward pass still has the high-resolution stem in front of it.
  late bucket   conv1 / layer1 / layer2 (a few MB), packed and reduced after backward: the only exposed communication.

The buckets are persistent flat buffers; after the exchange every `p.grad` is a view into its bucket, so clipping and
the fused optimiser run on them unchanged.  ReduceOp.AVG does the 1/W."""
import torch
import torch.distributed as dist


class TwoBucketGradSync:
    def __init__(self, model, early_modules=("dense_head", "roi_head"), process_group=None,
                 stem_modules=("conv1", "layer1", "layer2")):
        self.group = process_group
        early_ids = set()
        for name in early_modules:
            m = getattr(model, name, None)
            if m is not None:
                early_ids.update(id(p) for p in m.parameters())
        params = [p for p in model.parameters() if p.requires_grad]
        self.early = [p for p in params if id(p) in early_ids]
        stem_ids = set()
        backbone = getattr(model, "backbone_3d", None)
        for name in (stem_modules if backbone is not None else ()):
            m = getattr(backbone, name, None)
            if m is not None:
                stem_ids.update(id(p) for p in m.parameters())
        rest = [p for p in params if id(p) not in early_ids]
        self.mid = [p for p in rest if id(p) not in stem_ids] if stem_ids else []
        self.late = [p for p in rest if id(p) in stem_ids] if stem_ids else rest
        self._mid_work, self._mid_sent = None, False
        if backbone is not None and stem_ids:
            backbone.grad_sync = self                 # BiResNet.forward hooks layer2's output
        self._buf = {}
        self._work = None
        # RCCL/NCCL average in the collective; gloo (CPU tests) only sums
        self._avg = dist.is_initialized() and dist.get_backend(self.group) == "nccl"
        self._world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        self._early_sent = False
        # optional bf16 wire format (CG3D_GRAD_BF16=1): the fp32 bucket is rounded into a bf16 buffer, that one is
        # all-reduced (half the bytes per xGMI link) and widened back before clipping / AdamW.  Off by default: the
        # reference exchanges fp32 (torch DDP), and on one node the early / mid buckets hide under the backward pass anyway.
        self.grad_dtype = torch.bfloat16 if __import__("os").environ.get("CG3D_GRAD_BF16") == "1" else None
        self._wire = {}
        self._ev = []                                # (start, end) CUDA events around finish()'s exposed wait, per step
        # identical starting point on every rank (what DDP's constructor does)
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            for p in params:
                dist.broadcast(p.data, src=0, group=self.group)
            for b in model.buffers():
                dist.broadcast(b.data, src=0, group=self.group)

    def _bucket(self, key, plist):
        hit = self._buf.get(key)
        if hit is None:
            n = sum(p.numel() for p in plist)
            flat = torch.zeros(max(n, 1), dtype=plist[0].dtype if plist else torch.float32, device=plist[0].device if plist else "cpu")
            views, off = [], 0
            for p in plist:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            hit = self._buf[key] = (flat, views)
        return hit

    def _pack(self, key, plist):
        flat, views = self._bucket(key, plist)
        have = [(v, p.grad) for v, p in zip(views, plist) if p.grad is not None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(views, plist):
            if p.grad is None:
                v.zero_()
        return flat, views

    def _reduce(self, flat, async_op):
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if self.grad_dtype is not None:
            wire = self._wire.get(flat.data_ptr())
            if wire is None:
                wire = self._wire[flat.data_ptr()] = (torch.empty_like(flat, dtype=self.grad_dtype), flat)
            wire[0].copy_(flat)
            return dist.all_reduce(wire[0], op=op, group=self.group, async_op=async_op)
        return dist.all_reduce(flat, op=op, group=self.group, async_op=async_op)

    def report(self):
        """What the exchange costs THIS rank per step: bucket sizes and the exposed time -- the part of `finish()` the
        main stream spends on the late bucket's all-reduce and on waiting for the early / mid buckets sent from the
        backward pass (HIP events on the main stream; the overlapped part of the early / mid all-reduces is not in it)."""
        sizes = {k: int(self._buf[k][0].numel() * self._buf[k][0].element_size()) for k in self._buf}
        ms = []
        for e0, e1 in self._ev[-64:]:
            try:
                ms.append(e0.elapsed_time(e1))
            except Exception:
                pass
        ms.sort()
        return {"world": self._world, "bucket_bytes": sizes, "collectives_per_step": len(sizes),
                "exposed_ms_per_step": ms[len(ms) // 2] if ms else None,          # (the median; the key the round-5 review asked for)
                "exposed_ms_per_step_median": ms[len(ms) // 2] if ms else None, "exposed_ms_per_step_max": ms[-1] if ms else None,
                "steps_sampled": len(ms), "wire_dtype": str(self.grad_dtype or torch.float32)}

    # -- called from the detector's forward (training): hook the tensor where the head(s) attach to the backbone
    def attach(self, tensor):
        self._early_sent = False
        self._work = None
        if tensor.requires_grad:
            tensor.register_hook(self._on_backbone_output_grad)

    def begin_mid(self):
        self._mid_sent = False
        self._mid_work = None

    def attach_mid(self, tensor):
        self.begin_mid()
        if self.mid and tensor.requires_grad:
            tensor.register_hook(self._on_stem_output_grad)

    # Both hooks send unconditionally: whether a bucket leaves from its hook or from finish() must not depend on the data
    # (a rank whose RoI head saw no proposal has no gradient for it -> zeros), or the ranks would issue their
    # collectives in different orders.
    def _on_stem_output_grad(self, grad):
        # ORDER: early -> mid -> late on every rank, whichever hooks fired.  If the backbone-output hook did not run on this
        # rank (the tensor was not hooked, or no loss term reached it) the early bucket leaves here, before the mid one --
        # never after it from finish(), which would swap two collectives against the peers' order (a deadlock on RCCL).
        if self.early and not self._early_sent:
            flat_e, _ = self._pack("early", self.early)
            self._work = self._reduce(flat_e, True)
            self._early_sent = True
        flat, _ = self._pack("mid", self.mid)
        self._mid_work = self._reduce(flat, True)
        self._mid_sent = True
        return grad

    def _on_backbone_output_grad(self, grad):
        if self.early:
            flat, _ = self._pack("early", self.early)
            self._work = self._reduce(flat, True)
            self._early_sent = True
        return grad

    # -- called after loss.backward()
    def finish(self):
        timed = bool(self.late or self.early) and torch.cuda.is_available() and (self.late or self.early)[0].is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if not self._early_sent and self.early:
            flat, _ = self._pack("early", self.early)
            self._work = self._reduce(flat, True)
        if self.mid and not self._mid_sent:
            flat, _ = self._pack("mid", self.mid)
            self._mid_work = self._reduce(flat, True)
        if self.late:
            flat, _ = self._pack("late", self.late)
            self._reduce(flat, False)
        for w in (self._work, self._mid_work):
            if w is not None:
                w.wait()
        self._work = self._mid_work = None
        self._mid_sent = False
        for wire, flat in self._wire.values():
            flat.copy_(wire)
        if timed:
            e1.record()
            self._ev.append((e0, e1))
            if len(self._ev) > 256:
                del self._ev[:128]
        if not self._avg and self._world > 1:
            for key in self._buf:
                self._buf[key][0].div_(self._world)
        for key, plist in (("early", self.early), ("mid", self.mid), ("late", self.late)):
            if plist:
                for v, p in zip(self._buf[key][1], plist):
                    p.grad = v
        self._early_sent = False
