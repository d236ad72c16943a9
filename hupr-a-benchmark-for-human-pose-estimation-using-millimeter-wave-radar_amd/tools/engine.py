"""Training/inference engine shared by the Runner and bench.py: HuPRNet + LossComputer +
flat gradient buckets (+ RCCL all-reduce when world_size > 1) + fused Adam, optionally fed by the
on-GPU FFT loader (int16 ADC cubes -> normalised network input, config "C3" of BASELINE.json)."""

import os

import torch
import torch.distributed as dist

from ..misc.losses import LossComputer
from ..models import HuPRNet
from ..preprocessing.process_iwr1843 import fft_chain_loader, fft_chain_loader_means
from .distributed import GradientBuckets
from .optim import FusedAdam


class TrainEngine:
    def __init__(self, cfg, device="cuda", lr=None, seed=0, bucket_bytes=48 << 20):
        self.cfg = cfg
        self.device = torch.device(device)
        torch.manual_seed(seed)
        self.model = HuPRNet(cfg).to(self.device)
        self.lossComputer = LossComputer(cfg, self.device)
        self.buckets = GradientBuckets(self.model, bucket_bytes=bucket_bytes)
        self.buckets.broadcast_parameters(0)
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        # (data-parallel runs keep the library default of two compute streams: with the stream-ordered hupr_allreduce_bucket the
        # side-stream branch gains 3 % with collectives in flight — 1 368 -> 1 412 frames/s with forced single-rank collectives)
        self.optimizer = FusedAdam(self.model.parameters(), lr=lr if lr is not None else cfg.TRAINING.lr,
                                   betas=(0.9, 0.999), weight_decay=1e-4)
        self.optimizer.attach_flat_buckets(self.buckets.flat_pairs(), self.buckets.layout())
        self.optimizer.grad_scale = 1.0 / self.world_size
        self.G = cfg.DATASET.numGroupFrames
        self.fuse_elevation_mean = os.environ.get("HUPR_NO_FUSED_MEAN", "0") != "1"
        self._fft_ws = None
        self._seed = None
        self._graph = None
        self.keep_inference_graphs_current = True      # see _replay
        # the num_batches_tracked counters of all BatchNorms as views of one int64 vector: a training step bumps them
        # with ONE launch instead of one tiny add kernel per BatchNorm (30 per step); state_dict I/O is unchanged
        self._bns = [m for m in self.model.modules()
                     if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None]
        self._bn_counts = torch.zeros(max(len(self._bns), 1), dtype=torch.long, device=self.device)
        for i, m in enumerate(self._bns):
            self._bn_counts[i] = m.num_batches_tracked
            m._buffers["num_batches_tracked"] = self._bn_counts[i]

    def close(self):
        """Release the exchange step's native communicator (call before building another engine in the same process)."""
        self.buckets.close()

    # -- data -------------------------------------------------------------------------------------
    def preprocess(self, adc_hori, adc_vert):
        """int16 ADC cubes (B*G, 4, 192, 256, 2) per sensor -> the two network inputs.  Default (``fuse_elevation_mean``): the
        loader tensor already averaged over its elevation axis, (B, G, 16, R, A) planes — 1/8 of the bytes written by the FFT
        chain and read back by the MNet front end, bit-identical results; ``HUPR_NO_FUSED_MEAN=1`` keeps the reference-shaped
        (B,G,F,2,R,A,E) hand-over."""
        n = adc_hori.shape[0]
        B = n // self.G
        if self.fuse_elevation_mean:
            return (fft_chain_loader_means(adc_hori).view(B, self.G, 16, 64, 64),
                    fft_chain_loader_means(adc_vert).view(B, self.G, 16, 64, 64))
        h = fft_chain_loader(adc_hori).view(B, self.G, 8, 2, 64, 64, 8)
        v = fft_chain_loader(adc_vert).view(B, self.G, 8, 2, 64, 64, 8)
        return h, v

    # -- steps ------------------------------------------------------------------------------------
    def train_step(self, hori, vert, joints, decode=False, _last=True, _micro=1):
        """One forward + loss + backward (+ exchange + Adam when ``_last``).  ``decode="device"`` also runs the two
        arg-max decodes of the reference's per-iteration ``computeLoss`` (misc/losses.py:43-44) as kernels on the
        step's stream; their results stay on the device (``self.last_decode``)."""
        from .. import functional as F_
        self.model.train()
        self.buckets.prepare(reduce=_last)
        F_.BN_COUNTER_SINK = due = []
        try:
            preds = self.model(hori, vert)
        finally:
            F_.BN_COUNTER_SINK = None
        if len(due) == len(self._bns) and all(a is b for a, b in zip(sorted(due, key=id), sorted(self._bns, key=id))):
            self._bn_counts.add_(1)                         # every BatchNorm ran exactly once: one launch
        else:
            for m in due:
                m.num_batches_tracked.add_(1)
        loss, loss2, p2d, g2d = self.lossComputer.computeLoss(preds, joints, decode=decode)
        self.last_decode = (p2d, g2d)
        if self._seed is None or self._seed.device != loss.device:
            self._seed = torch.ones((), dtype=loss.dtype, device=loss.device)      # backward()'s own ones_like is a fill launch per step
        loss.backward(gradient=self._seed)
        if self.device.type == "cuda":
            for s in F_.side_streams_in_use(self.device):           # the side-stream branch's backward joins here
                torch.cuda.current_stream(self.device).wait_stream(s)
        if not _last:
            self.buckets.stash()
            return loss, loss2
        self.buckets.finish()
        self.optimizer.grad_scale = 1.0 / (self.world_size * _micro)
        self.optimizer.step()
        return loss, loss2

    def train_step_accumulated(self, micro_batches, from_adc=True, decode=False):
        """One optimiser step over several micro-batches (fixed GLOBAL batch, ``bench.py --strong``): gradients of the
        leading micro-batches are summed locally, the exchange happens once, overlapped with the last backward.
        ``micro_batches``: list of (hori, vert, joints) — ADC cubes when ``from_adc`` — each a per-rank micro-batch whose
        loss is its own mean, so the step's gradient is the mean over all ``world * len(micro_batches)`` of them.
        BatchNorm statistics are per micro-batch (as they are per rank in data parallel)."""
        m = len(micro_batches)
        out = None
        for i, (a, b, joints) in enumerate(micro_batches):
            h, v = self.preprocess(a, b) if from_adc else (a, b)
            out = self.train_step(h, v, joints, decode=decode, _last=i == m - 1, _micro=m)
        return out

    def train_step_from_adc(self, adc_hori, adc_vert, joints, decode=False):
        if self._graph is not None:
            return self._replay(adc_hori, adc_vert, joints)
        h, v = self.preprocess(adc_hori, adc_vert)
        return self.train_step(h, v, joints, decode=decode)

    # -- hipGraph capture of the whole step (single-GPU) --------------------------------------------------------
    def capture(self, adc_hori, adc_vert, joints, warmup=2, decode=False):
        """Capture preprocess + forward + loss + backward (+ the bucket all-reduces on the communication stream) + Adam
        as ONE hipGraph and replay it from then on (``train_step_from_adc``).  ~700 kernel launches per step otherwise
        cost 9-18 ms of host time, which bounds the step once the kernels are faster than that.  Data parallel: the
        exchange is ``hupr_allreduce_bucket`` (ncclAllReduce enqueued on a stream), which is capturable — the fork to the
        communication stream and the join before Adam become edges of the graph; every rank must capture and replay
        in lock step.  Requirements: fixed shapes (the static input buffers are refilled by copy), joints already on
        the device (a pageable host-to-device copy is illegal inside a capture), ``TRAINING.lossDecay == -1`` (the
        alpha/beta loss weights would be frozen at their capture-time values), ``sync_lr()`` after LR changes.
        The ``warmup`` eager steps are real optimisation steps."""
        tr = self.buckets.transport
        if self.buckets.active and not getattr(tr, "capturable", False):
            raise RuntimeError("graph capture needs the native RCCL transport (got %s)" % getattr(tr, "name", tr))
        if self.cfg.TRAINING.lossDecay != -1:
            raise RuntimeError("graph capture would freeze the loss weights alpha/beta (TRAINING.lossDecay != -1)")
        if not (adc_hori.is_cuda and adc_vert.is_cuda and joints.is_cuda):
            raise RuntimeError("graph capture needs ADC cubes and joints resident on the GPU")
        self.optimizer.use_device_state()
        self._g_decode = decode
        self._g_in = (adc_hori.clone(), adc_vert.clone(), joints.clone())
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                h, v = self.preprocess(self._g_in[0], self._g_in[1])
                self.train_step(h, v, self._g_in[2], decode=decode)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            h, v = self.preprocess(self._g_in[0], self._g_in[1])
            self._g_out = self.train_step(h, v, self._g_in[2], decode=decode)
        self._graph = g

    def sync_lr(self):
        self.optimizer.sync_lr()

    def _replay(self, adc_hori, adc_vert, joints):
        for dst, src in zip(self._g_in, (adc_hori, adc_vert, joints)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self._graph.replay()
        from .. import functional as F_
        F_.invalidate_packed()      # the graph's Adam node changed the parameters behind every host-side cache
        if self.keep_inference_graphs_current:
            # captured INFERENCE graphs (no_grad) read the cached packed / derived weights in place: refill them now, stream-ordered
            # behind the replay, so that such a graph replayed next sees this step's weights (one table launch)
            F_.refresh_packed(self.device)
        return self._g_out

    @torch.no_grad()
    def infer(self, hori, vert):
        self.model.eval()
        return self.model(hori, vert)
