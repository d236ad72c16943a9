"""Datasets feeding the GPU loader.

* ``SyntheticHuPR`` — build-owned synthetic IWR1843 ADC cubes + joints (no dataset ships with the
  repo; the real HuPR data is access-gated).  Items carry raw int16 cubes; the FFT chain and the
  loader normalisation run on the GPU inside the training step (config C3).
* ``window_indices`` — the reference's 8-frame window gather with sequence-edge clamping
  (datasets/dataset.py:120-139) as a pure function, used by ``HuPR3D_horivert``.
* ``HuPR3D_horivert`` — reader for the reference's on-disk layout (``single_<n>/{hori,vert}/%09d.npy``
  complex cubes + ``hrnet_annot_<phase>.json``); cubes go to the GPU as complex64 and through
  ``loader_normalize`` (the fused equivalent of the 256 per-slice transform calls of :144-150).
"""
import json
import os

import numpy as np
import torch
import torch.utils.data as data

from .. import synth
from ..preprocessing.process_iwr1843 import loader_normalize


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
                "bbox": torch.tensor([x0, y0, x1 - x0, y1 - y0], dtype=torch.float32)}


class HuPR3D_horivert(data.Dataset):
    """Real-data reader (reference layout).  Returns network-ready tensors on ``device``."""

    def __init__(self, phase, cfg, args, device="cuda"):
        if phase not in ("train", "val", "test"):
            raise ValueError("Invalid phase: {}".format(phase))
        self.phase, self.cfg, self.device = phase, cfg, device
        self.duration = cfg.DATASET.duration
        self.G = cfg.DATASET.numGroupFrames
        self.sampling_ratio = getattr(args, "sampling_ratio", 1)
        self.dirRoot = cfg.DATASET.dataDir
        groups = getattr(cfg.DATASET, phase + "Name")
        with open(os.path.join(self.dirRoot, "hrnet_annot_%s.json" % phase)) as fp:
            annot = json.load(fp)
        self.items = []
        for gi, blocks in enumerate(annot):
            for blk in blocks:
                frame = int(blk["image"][:-4])
                bbox = blk["bbox"]
                self.items.append({"seq": groups[gi], "frame": frame, "imageId": frame + groups[gi] * 100000,
                                   "joints": np.asarray(blk["joints"], dtype=np.float64),
                                   "bbox": [bbox[0], bbox[1], bbox[2] - bbox[0], bbox[3] - bbox[1]]})

    def __len__(self):
        return len(self.items) // self.sampling_ratio

    def _cube(self, item_idx, sensor):
        it = self.items[item_idx]
        path = os.path.join(self.dirRoot, "single_%d/%s/%09d.npy" % (it["seq"], sensor, it["frame"]))
        return torch.from_numpy(np.load(path).astype(np.complex64))

    def __getitem__(self, index):
        index = index * self.sampling_ratio
        idxs = window_indices(index, self.duration, self.G)
        out = {}
        for sensor in ("hori", "vert"):
            cubes = torch.stack([self._cube(i, sensor) for i in idxs]).to(self.device)
            out["VRDAEmap_" + sensor] = loader_normalize(cubes)          # (G, F, 2, R, A, E)
        it = self.items[index]
        out.update({"imageId": it["imageId"], "jointsGroup": torch.LongTensor(it["joints"]),   # truncates like the reference
                    "bbox": torch.FloatTensor(it["bbox"])})
        return out


def getDataset(phase, cfg, args, random=True):
    if str(cfg.DATASET.dataDir).startswith("synthetic") or not os.path.isdir(str(cfg.DATASET.dataDir)):
        return SyntheticHuPR(phase, cfg, args, length=getattr(args, "synthetic_length", 64))
    return HuPR3D_horivert(phase, cfg, args)
