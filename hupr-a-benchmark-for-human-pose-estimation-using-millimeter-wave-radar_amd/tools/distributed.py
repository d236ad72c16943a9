"""Data-parallel training over one node: one process per GPU, gradient sum-all-reduce over
RCCL/xGMI on a side HIP stream, overlapped with backward.

The reference has no distributed layer at all (single device string, tools/base.py:14); this is
the new part BASELINE.json asks for.  Design:

  * parameters and gradients live in a few flat fp32 buckets (reverse registration order ==
    roughly the order backward produces gradients: GCN -> decoder -> encoders -> MNets), so one
    collective moves tens of MB instead of 165 small tensors;
  * ``p.grad`` is a *view* into its bucket; the operator kernels write weight/bias gradients straight
    into those views (``functional.GRAD_SINK``: no per-parameter AccumulateGrad add kernels), anything
    autograd still accumulates itself lands in place; either way an arrival counter launches the
    bucket's all-reduce on the communication stream the moment its last gradient lands;
  * ``finish()`` makes the compute stream wait for the collectives; the optimiser then runs one
    fused Adam launch per bucket with grad_scale = 1/world_size (sum -> mean);
  * BatchNorm statistics stay per rank (the reference is single-device, no SyncBN).

Works with the ``gloo`` backend on CPU tensors too (no streams), which is how the ``not gpu``
tests exercise world_size 2.
"""
import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params, device):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat_param = torch.empty(self.numel, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(self.numel, dtype=torch.float32, device=device)
        off = 0
        self.views = []
        for p in params:
            n = p.numel()
            self.flat_param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + n].view_as(p)
            self.views.append(self.flat_grad[off:off + n].view_as(p))
            off += n
        self.pending = len(params)
        self.work = None
        self.index = {id(p): i for i, p in enumerate(params)}


class GradientBuckets:
    """Flat parameter/gradient buckets with overlapped all-reduce."""

    def __init__(self, module, bucket_bytes=48 << 20, process_group=None, tail_bytes=8 << 20):
        self.group = process_group
        # HUPR_FORCE_ALLREDUCE=1: run the collectives even with one rank (exercises RCCL + the side stream on a 1-GPU box)
        import os
        self.force_collective = os.environ.get("HUPR_FORCE_ALLREDUCE", "0") == "1" and dist.is_initialized()
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        self.device = params[0].device
        self.use_streams = self.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.use_streams else None
        # reverse registration order, split by size.  The all-reduce of the LAST bucket cannot overlap with anything
        # (its final gradient is the end of backward), so the trailing parameters get a small bucket of their own
        # (<= tail_bytes): the exposed collective is then latency- instead of bandwidth-sized.
        groups = []
        cur, cur_bytes = [], 0
        for p in reversed(params):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            groups.append(cur)
        last = groups[-1]
        if tail_bytes and sum(p.numel() for p in last) * 4 > tail_bytes and len(last) > 1:
            n_tail, acc = 0, 0
            for p in reversed(last):
                if acc + p.numel() * 4 > tail_bytes and n_tail > 0:
                    break
                acc += p.numel() * 4
                n_tail += 1
            if 0 < n_tail < len(last):
                groups[-1:] = [last[:-n_tail], last[-n_tail:]]
        self.buckets = [_Bucket(g, self.device) for g in groups]
        self._owner = {}
        self._by_ptr = {}                       # parameter storage address -> (bucket, index): the direct-write sink
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._owner[id(p)] = b
                self._by_ptr[p.data_ptr()] = (b, i)
                p.register_post_accumulate_grad_hook(self._hook)
        self.direct = self.device.type == "cuda"
        self.prepare()

    # -- direct gradient sink (functional.GRAD_SINK protocol) -----------------------------------
    def take(self, param):
        """View the kernel should write ``param``'s gradient into, or None (unknown tensor / sink inactive)."""
        ent = self._by_ptr.get(param.data_ptr()) if self._armed else None
        if ent is None:
            return None
        b, i = ent
        if b.written[i]:
            raise RuntimeError("parameter received two direct gradient writes in one backward pass "
                               "(shared parameters are not supported by the gradient sink)")
        b.written[i] = True
        return b.views[i]

    def done(self, param):
        b, _ = self._by_ptr[param.data_ptr()]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    # -- per-iteration protocol ---------------------------------------------------------------
    def prepare(self):
        """Zero the flat gradients and (re)install the views; call before every backward."""
        from .. import functional as F_
        for b in self.buckets:
            b.flat_grad.zero_()
            b.pending = len(b.params)
            b.written = [False] * len(b.params)
            b.work = None
            for p, v in zip(b.params, b.views):
                p.grad = v
        self._armed = self.direct
        F_.GRAD_SINK = self if self.direct else None

    def _hook(self, p):
        b = self._owner[id(p)]
        v = b.views[b.index[id(p)]]
        if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
            # autograd replaced the view (accumulation into an undefined grad): copy back
            v.copy_(p.grad)
            p.grad = v
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        if self.world_size == 1 and not self.force_collective:
            return
        if self.use_streams:
            from .. import functional as F_
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                for s in F_.side_streams_in_use(self.device):      # gradients of the side-stream branch (host-ordered earlier)
                    self.comm_stream.wait_stream(s)
                b.work = dist.all_reduce(b.flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Block the compute stream on outstanding collectives (call after backward)."""
        from .. import functional as F_
        self._armed = False
        if F_.GRAD_SINK is self:
            F_.GRAD_SINK = None
        for b in self.buckets:
            if b.pending != 0 and (self.world_size > 1 or self.force_collective):
                # a parameter received no gradient this iteration: reduce what we have
                b.pending = 0
                self._launch(b)
            if b.work is not None:
                b.work.wait()
                b.work = None
        if self.use_streams and (self.world_size > 1 or self.force_collective):
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)

    def flat_pairs(self):
        return [(b.flat_param, b.flat_grad) for b in self.buckets]

    def broadcast_parameters(self, src=0):
        """Make every rank start from rank ``src``'s weights (and buffers are left per rank)."""
        if self.world_size == 1:
            return
        for b in self.buckets:
            dist.broadcast(b.flat_param, src=src, group=self.group)
