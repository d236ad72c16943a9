"""Runner — train / eval loops with the reference's surface (tools/run.py:13-86):
``Runner(args, cfg)``, ``.loadModelWeight(mode)``, ``.train()``, ``.eval(visualization, epoch) -> AP``.

Differences that are the point of this build: the step runs through ``TrainEngine`` (HIP kernels,
flat gradient buckets, RCCL all-reduce when launched under torchrun), the FFT preprocessing can run
on the GPU inside the step (synthetic/raw-ADC datasets), and losses are not synchronised every step.
"""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.utils.data as data

from ..datasets import getDataset
from ..datasets.dataset import HuPRRawADC, SequenceGroupedSampler
from ..misc.oks_eval import evaluate_keypoints
from ..misc.plot import plotHumanPose
from .base import BaseRunner
from .engine import TrainEngine


def _collate(batch):
    out = {}
    for k in batch[0]:
        v = [b[k] for b in batch]
        out[k] = torch.stack(v) if isinstance(v[0], torch.Tensor) else torch.as_tensor(v)
    return out


class Runner(BaseRunner):
    def __init__(self, args, cfg):
        super().__init__(args, cfg)
        if self.device != "cuda":
            raise RuntimeError("the HIP runner needs a GPU (no CPU fallback)")
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if not args.eval:
            self.trainSet = getDataset("train", cfg, args)
            if isinstance(self.trainSet, HuPRRawADC):
                # raw captures: a cache miss costs a whole sequence (0.9 GB of adc_data.bin + its FFT), so the loader walks
                # shuffled GROUPS of sequences and shuffles the windows inside a group (SequenceGroupedSampler)
                sampler = SequenceGroupedSampler(self.trainSet, group=self.trainSet.cache_sequences, seed=args.seed,
                                                 rank=self.rank, world=self.world)
            else:
                sampler = data.distributed.DistributedSampler(self.trainSet, shuffle=True, drop_last=True) if self.world > 1 else None
            self.trainLoader = data.DataLoader(self.trainSet, cfg.TRAINING.batchSize, shuffle=sampler is None,
                                               sampler=sampler, num_workers=0, collate_fn=_collate,
                                               drop_last=self.world > 1)      # ragged last step only single-GPU
        else:
            self.trainLoader = [0]
        self.testSet = getDataset("test" if args.eval else "val", cfg, args)
        # does the GPU FFT chain feed this run (raw captures), or stored .npy cubes?  (tools/base.py: the preprocess.json sidecar)
        self.uses_gpu_fft_loader = isinstance(self.testSet, HuPRRawADC) or isinstance(getattr(self, "trainSet", None), HuPRRawADC)
        # evaluation is sharded over the ranks (rank r takes samples r, r + world, ...) and gathered on rank 0
        shard = data.Subset(self.testSet, range(self.rank, len(self.testSet), self.world)) if self.world > 1 else self.testSet
        self.testLoader = data.DataLoader(shard, cfg.TEST.batchSize, shuffle=False, num_workers=0, collate_fn=_collate)
        warm = cfg.TRAINING.warmupEpoch
        self.stepSize = len(self.trainLoader) * warm
        LR = cfg.TRAINING.lr if warm == -1 else cfg.TRAINING.lr / (cfg.TRAINING.warmupGrowth ** self.stepSize)
        self.engine = TrainEngine(cfg, device=torch.device("cuda", torch.cuda.current_device()), lr=LR, seed=args.seed)
        self.model, self.optimizer, self.lossComputer = self.engine.model, self.engine.optimizer, self.engine.lossComputer
        for d in (self.dir, self.visDir):
            os.makedirs(d, exist_ok=True)
        if not args.eval:
            print("==========>Train set size:", len(self.trainLoader))
        print("==========>Test set size:", len(self.testLoader))

    def _inputs(self, batch):
        dev = self.engine.device
        if "adc_hori" in batch:          # raw ADC cubes: FFT chain + loader glue on the GPU
            B, G = batch["adc_hori"].shape[:2]
            h = batch["adc_hori"].to(dev).reshape(B * G, 4, 192, 256, 2)
            v = batch["adc_vert"].to(dev).reshape(B * G, 4, 192, 256, 2)
            return self.engine.preprocess(h, v)
        return batch["VRDAEmap_hori"].float().to(dev), batch["VRDAEmap_vert"].float().to(dev)

    def eval(self, visualization=True, epoch=-1):
        """-> AP (reference tools/run.py:35-63).  Predictions are decoded from the GCN head, scaled to image pixels,
        written to ``<phase>_results.json`` by rank 0 and scored with the OKS evaluator against the ground truth the
        reference uses: the float joints of ``<phase>_gt.json`` (not the integer-truncated ``jointsGroup`` the loss sees).
        ``--keypoints`` prints the per-joint APs (``evaluateEach``) instead of the summary line."""
        self.logger.clear(len(self.testLoader.dataset))
        savePreds, gts = [], []
        for batch in self.testLoader:
            keypoints = batch["jointsGroup"]
            hori, vert = self._inputs(batch)
            preds = self.engine.infer(hori, vert)
            with torch.no_grad():
                loss, loss2, pred2d, _ = self.lossComputer.computeLoss(preds, keypoints)
            self.logger.display(loss, loss2, keypoints.size(0), epoch)
            if visualization:                       # --visDir given: one skeleton overlay per sample (run.py:50-52)
                plotHumanPose(pred2d * self.imgHeatmapRatio, self.cfg, self.visDir, batch["imageId"], None)
            self.saveKeypoints(savePreds, pred2d * self.imgHeatmapRatio, batch["bbox"], batch["imageId"])
            gt_joints = batch["jointsFloat"] if "jointsFloat" in batch else keypoints
            for j in range(keypoints.size(0)):
                gts.append({"image_id": int(batch["imageId"][j]), "keypoints": gt_joints[j].numpy().astype(np.float64),
                            "bbox": batch["bbox"][j].numpy().astype(np.float64)})
        if self.world > 1:
            parts = [None] * self.world
            dist.all_gather_object(parts, (savePreds, gts))
            savePreds = [r for p in parts for r in p[0]]
            gts = [g for p in parts for g in p[1]]
        ap = 0.0
        if self.rank == 0:
            self.writeKeypoints(savePreds)
            # ground truth = the whole <phase>_gt.json where the dataset has one (the reference scores against the full file,
            # datasets/dataset.py:68-88: with -sr > 1 the frames the loader never visited count as misses); the synthetic
            # dataset has no file, its ground truth is what the loader visited
            full_gt = getattr(self.testSet, "gt_annotations", None)
            if full_gt is not None:
                gts = full_gt
            names = ["AP", "Ap .5", "AP .75", "AP (M)", "AP (L)", "AR", "AR .5", "AR .75", "AR (M)", "AR (L)"]
            if getattr(self.args, "keypoints", False):            # evaluateEach, THEN the summary (tools/run.py:60-63)
                idx2j = self.cfg.DATASET.idxToJoints
                for k in range(self.numKeypoints):
                    print("%s: %.3f" % (idx2j[k], float(evaluate_keypoints(gts, savePreds, idx_keypoint=k)[0])))
            stats = evaluate_keypoints(gts, savePreds)
            print("  ".join("%s: %.3f" % (n, s) for n, s in zip(names, stats)))
            ap = float(stats[0])
        if self.world > 1:
            box = [ap]
            dist.broadcast_object_list(box, src=0)
            ap = box[0]
        return ap

    def train(self):
        for epoch in range(self.start_epoch, self.cfg.TRAINING.epochs):
            loss_list = []
            self.logger.clear(len(self.trainLoader.dataset))
            if hasattr(self.trainLoader.sampler, "set_epoch"):
                self.trainLoader.sampler.set_epoch(epoch)
            for idxBatch, batch in enumerate(self.trainLoader):
                hori, vert = self._inputs(batch)
                loss, loss2 = self.engine.train_step(hori, vert, batch["jointsGroup"])
                self.logger.display(loss, loss2, batch["jointsGroup"].size(0), epoch)
                if idxBatch % self.cfg.TRAINING.lrDecayIter == 0:
                    self.adjustLR(epoch)
                loss_list.append(loss.detach())
                if getattr(self.args, "max_steps", 0) and idxBatch + 1 >= self.args.max_steps:
                    break
            accAP = self.eval(visualization=False, epoch=epoch)
            if self.rank == 0:
                self.saveModelWeight(epoch, accAP)
                self.saveLosslist(epoch, [float(l) for l in loss_list], "train")
            if getattr(self.args, "max_epochs", 0) and epoch + 1 - self.start_epoch >= self.args.max_epochs:
                break
