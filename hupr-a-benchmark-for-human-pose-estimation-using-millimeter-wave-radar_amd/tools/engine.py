"""Training/inference engine shared by the Runner and bench.py: HuPRNet + LossComputer +
flat gradient buckets (+ RCCL all-reduce when world_size > 1) + fused Adam, optionally fed by the
on-GPU FFT loader (int16 ADC cubes -> normalised network input, config "C3" of BASELINE.json)."""
import torch
import torch.distributed as dist

from ..misc.losses import LossComputer
from ..models import HuPRNet
from ..preprocessing.process_iwr1843 import fft_chain_loader
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
        if self.world_size > 1 or self.buckets.force_collective:
            # measured with the RCCL path active (single rank, forced collectives): the side-stream branch costs 1 %
            # instead of gaining 2 % — the all-reduce kernels already fill the gaps it would use.  One compute stream then.
            from .. import functional as F_
            F_.TWO_STREAMS = False
        self.optimizer = FusedAdam(self.model.parameters(), lr=lr if lr is not None else cfg.TRAINING.lr,
                                   betas=(0.9, 0.999), weight_decay=1e-4)
        self.optimizer.attach_flat_buckets(self.buckets.flat_pairs())
        self.optimizer.grad_scale = 1.0 / self.world_size
        self.G = cfg.DATASET.numGroupFrames
        self._fft_ws = None
        self._graph = None
        # the num_batches_tracked counters of all BatchNorms as views of one int64 vector: a training step bumps them
        # with ONE launch instead of one tiny add kernel per BatchNorm (30 per step); state_dict I/O is unchanged
        self._bns = [m for m in self.model.modules()
                     if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None]
        self._bn_counts = torch.zeros(max(len(self._bns), 1), dtype=torch.long, device=self.device)
        for i, m in enumerate(self._bns):
            self._bn_counts[i] = m.num_batches_tracked
            m._buffers["num_batches_tracked"] = self._bn_counts[i]

    # -- data -------------------------------------------------------------------------------------
    def preprocess(self, adc_hori, adc_vert):
        """int16 ADC cubes (B*G, 4, 192, 256, 2) per sensor -> two (B,G,F,2,R,A,E) fp32 network inputs."""
        n = adc_hori.shape[0]
        B = n // self.G
        h = fft_chain_loader(adc_hori).view(B, self.G, 8, 2, 64, 64, 8)
        v = fft_chain_loader(adc_vert).view(B, self.G, 8, 2, 64, 64, 8)
        return h, v

    # -- steps ------------------------------------------------------------------------------------
    def train_step(self, hori, vert, joints):
        from .. import functional as F_
        self.model.train()
        self.buckets.prepare()
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
        loss, loss2, _, _ = self.lossComputer.computeLoss(preds, joints, decode=False)
        loss.backward()
        if self.device.type == "cuda":
            for s in F_.side_streams_in_use(self.device):           # the side-stream branch's backward joins here
                torch.cuda.current_stream(self.device).wait_stream(s)
        self.buckets.finish()
        self.optimizer.step()
        return loss, loss2

    def train_step_from_adc(self, adc_hori, adc_vert, joints):
        if self._graph is not None:
            return self._replay(adc_hori, adc_vert, joints)
        h, v = self.preprocess(adc_hori, adc_vert)
        return self.train_step(h, v, joints)

    # -- hipGraph capture of the whole step (single-GPU) --------------------------------------------------------
    def capture(self, adc_hori, adc_vert, joints, warmup=2):
        """Capture preprocess + forward + loss + backward + Adam as ONE hipGraph and replay it from then on
        (``train_step_from_adc``).  ~1000 kernel launches per step otherwise cost ~24 ms of host time, which bounds the
        step once the kernels are faster than that.  Requirements: fixed shapes (the static input buffers are refilled
        by copy), world size 1 (the RCCL all-reduce stays on the eager path), ``sync_lr()`` after LR changes.  The
        ``warmup`` eager steps are real optimisation steps."""
        if self.world_size != 1:
            raise RuntimeError("graph capture is only wired for single-GPU runs")
        self.optimizer.use_device_state()
        self._g_in = (adc_hori.clone(), adc_vert.clone(), joints.clone())
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                h, v = self.preprocess(self._g_in[0], self._g_in[1])
                self.train_step(h, v, self._g_in[2])
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            h, v = self.preprocess(self._g_in[0], self._g_in[1])
            self._g_out = self.train_step(h, v, self._g_in[2])
        self._graph = g

    def sync_lr(self):
        self.optimizer.sync_lr()

    def _replay(self, adc_hori, adc_vert, joints):
        for dst, src in zip(self._g_in, (adc_hori, adc_vert, joints)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self._graph.replay()
        return self._g_out

    @torch.no_grad()
    def infer(self, hori, vert):
        self.model.eval()
        return self.model(hori, vert)
