"""Adam with coupled L2 weight decay on the gfx950 fused kernel.

Same update rule and ``state_dict`` layout as ``torch.optim.Adam`` (what the reference builds in
tools/base.py:47: lr 1e-4, betas (0.9, 0.999), weight_decay 1e-4 on every parameter), so the
reference's ``optimizer_state_dict`` checkpoints interchange.  One kernel launch per parameter
tensor, or one per flat bucket when the parameters were flattened by ``tools.distributed``.
"""
import torch

from .. import runtime as rt


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self.grad_scale = 1.0            # e.g. 1/world_size when gradients were sum-all-reduced
        self._flat = None                # optional [(param_flat, grad_flat)] installed by tools.distributed

    def attach_flat_buckets(self, buckets, layout=None):
        """buckets: list of (flat_param, flat_grad) fp32 GPU tensors covering all parameters in order.
        layout: per bucket, the list of (parameter, offset, numel) it holds — needed to save / restore the moments in
        ``torch.optim.Adam``'s per-parameter ``state_dict`` layout (``GradientBuckets.layout()``)."""
        self._flat = buckets
        self._layout = layout
        self._flat_state = [dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p)) for p, _ in buckets]
        self._dev_state = None           # {lr, step} on the device: set by use_device_state() for graph-captured steps

    def use_device_state(self):
        """Keep the learning rate and the step count in device memory (needed when step() is captured in a hipGraph:
        launch arguments are frozen at capture, the bias corrections and LR schedule must keep moving)."""
        dev = self._flat[0][0].device
        self._dev_state = torch.tensor([self.param_groups[0]["lr"], float(self._flat_state[0]["step"])], dtype=torch.float32,
                                       device=dev)
        self._dev_lr = self.param_groups[0]["lr"]

    # -- checkpoint interchange with torch.optim.Adam (reference tools/base.py:76-81,113) ------------------------
    def _host_step(self, i):
        if self._dev_state is not None:          # during graph replay only the device copy advances
            self._flat_state[i]["step"] = int(round(float(self._dev_state[1].item())))
        return self._flat_state[i]["step"]

    def state_dict(self):
        """Same layout as ``torch.optim.Adam.state_dict()``: per parameter ``{step, exp_avg, exp_avg_sq}`` — the flat
        moment buffers are scattered into per-parameter tensors (copies), so a checkpoint written here resumes under
        ``torch.optim.Adam`` and vice versa."""
        if self._flat is None:
            return super().state_dict()
        if self._layout is None:
            raise RuntimeError("flat buckets attached without a layout: the optimiser state cannot be serialised")
        saved = self.state
        self.state = type(saved)()
        try:
            for i, entries in enumerate(self._layout):
                step = self._host_step(i)
                if step == 0:
                    continue                      # torch.optim.Adam has no state before its first step either
                st = self._flat_state[i]
                for p, off, n in entries:
                    self.state[p] = {"step": torch.tensor(float(step)),
                                     "exp_avg": st["exp_avg"][off:off + n].clone().view_as(p),
                                     "exp_avg_sq": st["exp_avg_sq"][off:off + n].clone().view_as(p)}
            return super().state_dict()
        finally:
            self.state = saved

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)      # fills self.state per parameter (cast to the parameter's device)
        if self._flat is None:
            return
        if self._layout is None:
            raise RuntimeError("flat buckets attached without a layout: the optimiser state cannot be restored")
        for i, entries in enumerate(self._layout):
            st = self._flat_state[i]
            step = 0
            st["exp_avg"].zero_()
            st["exp_avg_sq"].zero_()
            for p, off, n in entries:
                ps = self.state.get(p)
                if not ps:
                    continue
                st["exp_avg"][off:off + n].copy_(ps["exp_avg"].reshape(-1))
                st["exp_avg_sq"][off:off + n].copy_(ps["exp_avg_sq"].reshape(-1))
                step = max(step, int(round(float(ps["step"]))))
            st["step"] = step
        self.state.clear()                       # the flat buffers are the state from here on
        if self._dev_state is not None:
            self._dev_lr = self.param_groups[0]["lr"]
            self._dev_state.copy_(torch.tensor([self._dev_lr, float(self._flat_state[0]["step"])], dtype=torch.float32))

    def sync_lr(self):
        """Push a changed param_groups lr to the device state (call outside graph replay, e.g. once per epoch)."""
        if self._dev_state is not None and self.param_groups[0]["lr"] != self._dev_lr:
            self._dev_lr = self.param_groups[0]["lr"]
            self._dev_state[0] = self._dev_lr

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = rt.lib()
        s = rt.stream()
        from .. import functional as F_
        F_.invalidate_packed()             # parameters change below without bumping torch's version counters
        if self._flat is not None:
            g0 = self.param_groups[0]
            b1, b2 = g0["betas"]
            if self._dev_state is not None:
                self._dev_state[1] += 1          # device-side step count (captured as a graph node)
                for (p, g), st in zip(self._flat, self._flat_state):
                    st["step"] += 1              # host mirror (checkpoints); during replay only the device copy advances
                    rt.check(L.hupr_adam_step_dev_f32(rt.ptr(p), rt.ptr(g), rt.ptr(st["exp_avg"]), rt.ptr(st["exp_avg_sq"]),
                                                      p.numel(), rt.ptr(self._dev_state), b1, b2, g0["eps"],
                                                      g0["weight_decay"], self.grad_scale, s))
                return loss
            for (p, g), st in zip(self._flat, self._flat_state):
                st["step"] += 1
                rt.check(L.hupr_adam_step_f32(rt.ptr(p), rt.ptr(g), rt.ptr(st["exp_avg"]), rt.ptr(st["exp_avg_sq"]),
                                              p.numel(), g0["lr"], b1, b2, g0["eps"], g0["weight_decay"], st["step"],
                                              self.grad_scale, s))
            return loss
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                rt.check(L.hupr_adam_step_f32(rt.ptr(p), rt.ptr(g), rt.ptr(st["exp_avg"]), rt.ptr(st["exp_avg_sq"]),
                                              p.numel(), group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                              int(st["step"].item()), self.grad_scale, s))
        return loss
