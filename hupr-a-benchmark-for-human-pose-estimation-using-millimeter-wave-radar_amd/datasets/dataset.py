"""Datasets feeding the GPU loader.

* ``SyntheticHuPR`` — build-owned synthetic IWR1843 ADC cubes + joints (no dataset ships with the
  repo; the real HuPR data is access-gated).  Items carry raw int16 cubes; the FFT chain and the
  loader normalisation run on the GPU inside the training step (config C3).
* ``window_indices`` — the reference's 8-frame window gather with sequence-edge clamping
  (datasets/dataset.py:120-139) as a pure function, used by ``HuPR3D_horivert``.
* ``HuPR3D_horivert`` — reader for the reference's on-disk layout (``single_<n>/{hori,vert}/%09d.npy``
  complex cubes + ``hrnet_annot_<phase>.json``); cubes go to the GPU as complex64 and through
  ``loader_normalize`` (the fused equivalent of the 256 per-slice transform calls of :144-150).  Same item
  dictionary, index arithmetic (``sampling_ratio``, random multiplier) and ``<phase>_gt.json`` /
  ``evaluate`` / ``evaluateEach`` protocol as the reference class.
* ``HuPRRawADC`` — the same items straight from the DCA1000 captures (``single_<n>/{hori,vert}/adc_data.bin``,
  what the reference's offline driver process_iwr1843.py:184-196 consumes): a sequence is de-interleaved and
  transformed ONCE on the GPU (``SequenceFFTCache``) and windows are gathers — no 8.39 MB/frame ``.npy`` hand-off.
"""
import json
import os
import random as _random
from collections import OrderedDict

import numpy as np
import torch
import torch.utils.data as data

from .. import synth
from ..misc.oks_eval import evaluate_keypoints
from ..preprocessing.process_iwr1843 import dca1000_frames, loader_normalize
from .base import generateGTAnnot


def window_indices(index, duration, num_group_frames):
    """Frame indices of the G-frame window centred on ``index`` within its ``duration``-frame sequence.
    Frames before the sequence start repeat the first frame, frames past the end repeat the last."""
    half = num_group_frames // 2
    pos = index % duration                      # position inside the sequence
    first, last = index - pos, index - pos + duration - 1
    out = []
    cur = index - half - 1
    for j in range(num_group_frames):
        if j + pos <= half:
            cur = first
        elif j > (duration - 1 - pos) + half:
            cur = last
        else:
            cur += 1
        out.append(cur)
    return out


class SequenceFFTCache:
    """Per-sequence cache of loader-ready sensor-frames: consecutive samples of a sequence share 7 of
    their 8 window frames, so transforming every frame of a sequence ONCE (``fft_chain_loader`` on the whole
    (frames,4,192,256,2) int16 cube) and assembling windows by gather does 1/8 of the FFT work of the un-cached loader.

    ``adc_hori`` / ``adc_vert``: int16 GPU tensors (duration, 4, 192, 256, 2) of one sequence.  ``window(index)`` returns
    the two (G, 8, 2, 64, 64, 8) fp32 network inputs of sample ``index`` (position inside the sequence), with the
    reference's edge clamping (``window_indices``).  Memory: 2.1 MB per cached sensor-frame (600 frames x 2 = 2.5 GB)."""

    def __init__(self, adc_hori, adc_vert, num_group_frames):
        from ..preprocessing.process_iwr1843 import fft_chain_loader
        if adc_hori.shape != adc_vert.shape:
            raise ValueError("hori / vert sequences must have the same number of frames")
        self.duration = adc_hori.shape[0]
        self.G = num_group_frames
        self.hori = fft_chain_loader(adc_hori)          # (duration, 8, 2, 64, 64, 8)
        self.vert = fft_chain_loader(adc_vert)

    def window(self, index):
        idx = torch.tensor(window_indices(index, self.duration, self.G), device=self.hori.device)
        return self.hori.index_select(0, idx), self.vert.index_select(0, idx)

    def batch(self, indices):
        """Network inputs (B, G, 8, 2, 64, 64, 8) x 2 for the samples at ``indices`` of this sequence."""
        idx = torch.tensor([window_indices(i, self.duration, self.G) for i in indices], device=self.hori.device)
        B = idx.shape[0]
        h = self.hori.index_select(0, idx.reshape(-1)).reshape(B, self.G, 8, 2, 64, 64, 8)
        v = self.vert.index_select(0, idx.reshape(-1)).reshape(B, self.G, 8, 2, 64, 64, 8)
        return h, v


class SyntheticHuPR(data.Dataset):
    def __init__(self, phase, cfg, args=None, length=256, seed=0):
        if phase not in ("train", "val", "test"):
            raise ValueError("Invalid phase: {}".format(phase))
        self.phase, self.cfg, self.length = phase, cfg, length
        self.G = cfg.DATASET.numGroupFrames
        self.seed = seed + {"train": 0, "val": 1000, "test": 2000}[phase]
        self.duration = cfg.DATASET.duration

    def __len__(self):
        return self.length

    def frame_cube(self, frame_idx, sensor):
        return synth.adc_cube_int16(self.seed, seq=0, frame=frame_idx, sensor=sensor)[0]

    def __getitem__(self, index):
        idxs = window_indices(index, self.duration, self.G)
        hori = np.stack([self.frame_cube(i, 0) for i in idxs])
        vert = np.stack([self.frame_cube(i, 1) for i in idxs])
        joints = synth.keypoints(1, self.seed * 100003 + index)[0]
        x0, y0 = joints.min(0)
        x1, y1 = joints.max(0)
        return {"adc_hori": torch.from_numpy(hori), "adc_vert": torch.from_numpy(vert),
                "imageId": index, "jointsGroup": torch.from_numpy(joints),
                "jointsFloat": torch.from_numpy(joints.astype(np.float64)),
                "bbox": torch.tensor([x0, y0, x1 - x0, y1 - y0], dtype=torch.float32)}


class _AnnotatedHuPR(data.Dataset):
    """Shared host logic of the two real-data datasets: the annotation list in ``<phase>_gt.json`` order, the reference's
    index arithmetic, and its evaluation protocol (datasets/dataset.py:17-46,48-88,120-124,161-162)."""

    def __init__(self, phase, cfg, args, random=True):
        if phase not in ("train", "val", "test"):
            raise ValueError("Invalid phase: {}".format(phase))
        self.phase, self.cfg = phase, cfg
        self.duration = cfg.DATASET.duration
        self.G = self.numGroupFrames = cfg.DATASET.numGroupFrames
        self.numKeypoints = cfg.DATASET.numKeypoints
        self.sampling_ratio = getattr(args, "sampling_ratio", 1)
        self.dirRoot = cfg.DATASET.dataDir
        self.idxToJoints = cfg.DATASET.idxToJoints
        self.random = random
        gt = generateGTAnnot(cfg, phase)                      # writes <dataDir>/<phase>_gt.json like the reference (:35)
        self.gtFile = os.path.join(self.dirRoot, "%s_gt.json" % phase)
        self.gt_annotations = gt["annotations"]
        self.imageIds = [im["id"] for im in gt["images"]]
        self.items = []
        for ann in self.gt_annotations:
            name = "%09d" % ann["image_id"]                   # sequence / frame decoded like :41-44
            kp = np.asarray(ann["keypoints"], dtype=np.float64).reshape(-1, 3)[:, :2]
            self.items.append({"seq": int(name[:4]), "frame": int(name[-4:]), "imageId": ann["image_id"], "joints": kp,
                               "bbox": ann["bbox"]})

    def __len__(self):
        return len(self.items) // self.sampling_ratio

    def _index(self, index):
        """:121-124 — with ``random`` (the reference default, also for val/test) the sample index is multiplied by a random
        factor in [1, sampling_ratio]; at ``-sr 1`` both forms are the identity."""
        return index * (_random.randint(1, self.sampling_ratio) if self.random else self.sampling_ratio)

    def _labels(self, index):
        it = self.items[index]
        return {"imageId": it["imageId"], "jointsGroup": torch.LongTensor(it["joints"]),        # truncates like :152
                "jointsFloat": torch.from_numpy(it["joints"].copy()),                          # what <phase>_gt.json holds
                "bbox": torch.FloatTensor(it["bbox"])}

    # ---- evaluation protocol (COCO OKS against <phase>_gt.json) -------------------------------------------------
    def _stats(self, loadDir, idx_keypoint=-1):
        with open(os.path.join(loadDir, "%s_results.json" % self.phase)) as fp:
            dts = json.load(fp)
        return evaluate_keypoints(self.gt_annotations, dts, idx_keypoint)

    def evaluate(self, loadDir):
        """AP of ``<loadDir>/<phase>_results.json`` (datasets/dataset.py:68-88): prints the ten numbers, returns AP."""
        stats = self._stats(loadDir)
        names = ["AP", "Ap .5", "AP .75", "AP (M)", "AP (L)", "AR", "AR .5", "AR .75", "AR (M)", "AR (L)"]
        for i, (n, v) in enumerate(zip(names, stats)):
            print("%s:\t%.3f\t" % (n, v), end="\n" if (i + 1) % 5 == 0 else "")
        return stats[0]

    def evaluateEach(self, loadDir):
        """Per-joint AP (``--keypoints``; datasets/dataset.py:48-66): prints one line per joint, returns the last AP."""
        aps = [self._stats(loadDir, i)[0] for i in range(self.numKeypoints)]
        for i, ap in enumerate(aps):
            print("%s: %.3f" % (self.idxToJoints[i], ap))
        return aps[-1]


class HuPR3D_horivert(_AnnotatedHuPR):
    """Real-data reader (reference layout).  Returns network-ready tensors on ``device``."""

    def __init__(self, phase, cfg, args, random=True, device="cuda"):
        super().__init__(phase, cfg, args, random)
        self.device = device
        self.VRDAEPaths_hori = [os.path.join(self.dirRoot, "single_%d/hori/%09d.npy" % (it["seq"], it["frame"])) for it in self.items]
        self.VRDAEPaths_vert = [os.path.join(self.dirRoot, "single_%d/vert/%09d.npy" % (it["seq"], it["frame"])) for it in self.items]

    def __getitem__(self, index):
        index = self._index(index)
        idxs = window_indices(index, self.duration, self.G)
        out = {}
        for sensor, paths in (("hori", self.VRDAEPaths_hori), ("vert", self.VRDAEPaths_vert)):
            cubes = torch.stack([torch.from_numpy(np.load(paths[i]).astype(np.complex64)) for i in idxs]).to(self.device)
            out["VRDAEmap_" + sensor] = loader_normalize(cubes)          # (G, F, 2, R, A, E)
        out.update(self._labels(index))
        return out


class HuPRRawADC(_AnnotatedHuPR):
    """Items straight from the raw captures: ``<rawDir>/single_<n>/{hori,vert}/adc_data.bin`` (the files the reference's
    offline driver reads, process_iwr1843.py:184-196) + the same ``hrnet_annot_<phase>.json``.  A sequence's two streams
    go to the GPU once (int16), are de-interleaved there (``hupr_dca1000_deinterleave``) and transformed by the fused FFT
    loader for all frames at once; items are window gathers out of an LRU of ``cache_sequences`` sequences (a 600-frame
    sequence costs 2 x 1.26 GB of loader-ready cubes and ~1 ms of FFT time)."""

    def __init__(self, phase, cfg, args, random=True, device="cuda", cache_sequences=4):
        super().__init__(phase, cfg, args, random)
        self.device = device
        self.rawRoot = getattr(cfg.DATASET, "rawDir", None) or self.dirRoot
        self.cache_sequences = cache_sequences
        self._cache = OrderedDict()
        self._seq_start = {}
        for i, it in enumerate(self.items):                   # first item of every sequence (frames are contiguous per sequence)
            self._seq_start.setdefault(it["seq"], i)

    def _sequence(self, seq):
        c = self._cache.get(seq)
        if c is None:
            streams = []
            for sensor in ("hori", "vert"):
                path = os.path.join(self.rawRoot, "single_%d" % seq, sensor, "adc_data.bin")
                raw = torch.from_numpy(np.fromfile(path, dtype=np.int16)).to(self.device)
                streams.append(dca1000_frames(raw))            # (frames, 4, 192, 256, 2) int16
            c = SequenceFFTCache(streams[0], streams[1], self.G)
            self._cache[seq] = c
            while len(self._cache) > self.cache_sequences:
                self._cache.popitem(last=False)
        else:
            self._cache.move_to_end(seq)
        return c

    def __getitem__(self, index):
        index = self._index(index)
        it = self.items[index]
        cache = self._sequence(it["seq"])
        pos = index - self._seq_start[it["seq"]]
        if cache.duration != self.duration:
            raise ValueError("single_%d holds %d frames, cfg.DATASET.duration is %d" % (it["seq"], cache.duration, self.duration))
        h, v = cache.window(pos)
        out = {"VRDAEmap_hori": h, "VRDAEmap_vert": v}
        out.update(self._labels(index))
        return out


class SequenceGroupedSampler(data.Sampler):
    """Shuffled training order for ``HuPRRawADC`` that keeps its sequence cache hot.  A uniformly shuffled loader touches a
    different sequence with nearly every sample, and a miss re-reads ~0.9 GB of ``adc_data.bin`` and transforms 2 x 600
    sensor-frames.  Here the SEQUENCES are shuffled (same order on every rank, seeded by epoch), every rank takes its own
    sequences (``rank::world``), and the windows of ``group`` sequences at a time are shuffled together: with
    ``group <= cache_sequences`` every sequence is read and transformed exactly once per epoch and a batch still mixes
    windows of ``group`` recordings.  Memory bound of the cache: ``cache_sequences`` x 2 sensors x duration x 2.1 MB
    (4 x 2.5 GB = 10 GB at the default).  All ranks yield the same number of indices in EVERY epoch (the smallest share any
    shuffle can produce).
    With ``-sr > 1`` and the reference's random index multiplier (datasets/dataset.py:121-124) a sampled index may land
    in an earlier sequence than the one it is grouped with: still correct, occasionally a miss."""

    def __init__(self, dataset, group=4, seed=0, rank=0, world=1):
        self.ds, self.group, self.seed, self.rank, self.world = dataset, max(1, int(group)), int(seed), rank, world
        self.epoch = 0
        sr = dataset.sampling_ratio
        self.by_seq = OrderedDict()
        for i in range(len(dataset)):
            self.by_seq.setdefault(dataset.items[min(i * sr, len(dataset.items) - 1)]["seq"], []).append(i)
        if len(self.by_seq) < world:
            raise ValueError("%d sequences cannot be sharded over %d ranks" % (len(self.by_seq), world))
        # The length must not depend on the epoch (tools/run.py fixes the LR warm-up step count and the logger from
        # epoch 0): every epoch yields the count of the SMALLEST share any shuffle can produce — the per-rank number of sequences
        # times the shortest sequence.  HuPR's sequences all hold `duration` windows, so nothing is dropped there.
        per_rank = len(self.by_seq) // world
        self._n = sum(sorted(len(v) for v in self.by_seq.values())[:per_rank])

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _plan(self):
        rng = np.random.default_rng([self.seed, self.epoch])
        seqs = list(self.by_seq)
        rng.shuffle(seqs)
        per_rank = len(seqs) // self.world
        shares = [seqs[r::self.world][:per_rank] for r in range(self.world)]
        return shares, self._n

    def __len__(self):
        return self._n

    def __iter__(self):
        shares, n = self._plan()
        mine = shares[self.rank]
        rng = np.random.default_rng([self.seed, self.epoch, self.rank + 1])
        out = []
        for g0 in range(0, len(mine), self.group):
            pool = np.concatenate([np.asarray(self.by_seq[s]) for s in mine[g0:g0 + self.group]])
            rng.shuffle(pool)
            out.extend(int(i) for i in pool)
        return iter(out[:n])


def getDataset(phase, cfg, args, random=True):
    """Reference signature (datasets/dataset.py:14).  ``dataDir: synthetic`` selects the build-owned generator,
    ``DATASET.rawDir`` (opt-in key) the raw-capture reader, anything else the reference's ``.npy`` layout — a missing
    directory is an error, never a silent switch to synthetic data."""
    root = str(cfg.DATASET.dataDir)
    if root.startswith("synthetic"):
        return SyntheticHuPR(phase, cfg, args, length=getattr(args, "synthetic_length", 64))
    if not os.path.isdir(root):
        raise FileNotFoundError("cfg.DATASET.dataDir %r does not exist (use dataDir: synthetic for the built-in generator)" % root)
    if getattr(cfg.DATASET, "rawDir", None):
        return HuPRRawADC(phase, cfg, args, random)
    return HuPR3D_horivert(phase, cfg, args, random)
