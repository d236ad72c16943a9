"""BaseRunner — host-side bookkeeping of the reference's tools/base.py (device choice, seeds, LR
schedule, checkpoint files, keypoint JSON records), kept API-compatible:

  checkpoint dict keys  epoch / model_state_dict / optimizer_state_dict / accuracy   (:76-81)
  files                 logs/<dir>/{checkpoint,model_best,checkpoint_<e>}.pth        (:82-90)
  LR schedule           lr *= lrDecay (or warmupGrowth before warmupEpoch)           (:66-72)
  result records        category_id/center/image_id/scale/score/keypoints           (:124-147)

The reference's resume path is broken as shipped (args.pretrained undefined, Logger.updateBestAcc
missing — SURVEY.md section 0 fact 5); here ``pretrained`` defaults to False and resume works.
"""
import json
import os

import numpy as np
import torch

from ..misc.logger import Logger


class BaseRunner():
    def __init__(self, args, cfg):
        self.device = "cuda" if torch.cuda.is_available() and args.gpuIDs else "cpu"
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(args.seed)
        self.dir = "./logs/" + args.dir
        self.visDir = "./visualization/" + args.visDir
        self.args = args
        self.cfg = cfg
        self.heatmapSize = self.width = self.height = cfg.DATASET.heatmapSize
        self.imgSize = self.imgWidth = self.imgHeight = cfg.DATASET.imgSize
        self.numKeypoints = cfg.DATASET.numKeypoints
        self.dimsWidthHeight = (self.width, self.height)
        self.start_epoch = 0
        self.numFrames = cfg.DATASET.numFrames
        self.F = cfg.DATASET.numGroupFrames
        self.imgHeatmapRatio = cfg.DATASET.imgSize / cfg.DATASET.heatmapSize
        self.aspectRatio = self.imgWidth * 1.0 / self.imgHeight
        self.pixel_std = 200
        self.logger = Logger()

    # ---- geometry / records ------------------------------------------------------------------------
    def _xywh2cs(self, x, y, w, h):
        center = np.array([x + w * 0.5, y + h * 0.5], dtype=np.float32)
        if w > self.aspectRatio * h:
            h = w * 1.0 / self.aspectRatio
        elif w < self.aspectRatio * h:
            w = h * self.aspectRatio
        scale = np.array([w * 1.0 / self.pixel_std, h * 1.0 / self.pixel_std], dtype=np.float32)
        if center[0] != -1:
            scale = scale * 1.25
        return center, scale

    def saveKeypoints(self, savePreds, preds, bbox, image_id, predHeatmap=None):
        """Append one COCO-keypoint detection record per sample (visibility 1, score 1.0)."""
        n = len(preds)
        kp = np.concatenate((np.asarray(preds), np.ones((n, self.numKeypoints, 1))), axis=2)
        for j in range(n):
            center, scale = self._xywh2cs(bbox[j][0], bbox[j][1], bbox[j][2], bbox[j][3])
            iid = image_id[j]
            rec = {"category_id": 1, "center": center.tolist(),
                   "image_id": iid.item() if hasattr(iid, "item") else int(iid), "scale": scale.tolist(),
                   "score": 1.0, "keypoints": kp[j].reshape(self.numKeypoints * 3).tolist()}
            if predHeatmap is not None:
                rec["sigma"] = [float(predHeatmap[j, k].var().item() * self.heatmapSize) for k in range(self.numKeypoints)]
            savePreds.append(rec)
        return savePreds

    def writeKeypoints(self, preds):
        name = "test_results.json" if self.args.eval else "val_results.json"
        with open(os.path.join(self.dir, name), "w") as fp:
            json.dump(preds, fp)

    # ---- schedule / checkpoints --------------------------------------------------------------------
    def adjustLR(self, epoch):
        factor = self.cfg.TRAINING.warmupGrowth if epoch < self.cfg.TRAINING.warmupEpoch else self.cfg.TRAINING.lrDecay
        for group in self.optimizer.param_groups:
            group["lr"] *= factor

    def _checkpoint_dict(self, epoch):
        return {"epoch": epoch, "model_state_dict": self.model.state_dict(),
                "optimizer_state_dict": self.optimizer.state_dict(), "accuracy": self.logger.showBestAP()}

    def saveModelWeight(self, epoch, acc):
        best = self.logger.isBestAccAP(acc)
        group = self._checkpoint_dict(epoch)
        if best:
            print("==========>Save the best model...")
            torch.save(group, os.path.join(self.dir, "model_best.pth"))
        print("==========>Save the latest model...")
        torch.save(group, os.path.join(self.dir, "checkpoint.pth"))
        # Which zero-Doppler convention the on-GPU FFT loader fed these weights (preprocessing.ZERO_DOPPLER).  A sidecar,
        # not a checkpoint key: the checkpoint dict keeps exactly the reference's four keys (tools/base.py:76-81).
        # Written only when the GPU FFT loader actually fed this run (raw-capture dataset): a run on stored .npy cubes never touched
        # the chain, and a sidecar there would claim a convention nobody used.
        if getattr(self, "uses_gpu_fft_loader", False):
            from ..preprocessing import process_iwr1843 as _pre
            with open(os.path.join(self.dir, "preprocess.json"), "w") as fp:
                json.dump({"fft_zero_doppler": _pre.ZERO_DOPPLER}, fp)
        if epoch % 5 == 0:
            torch.save(group, os.path.join(self.dir, "checkpoint_%d.pth" % epoch))

    def saveLosslist(self, epoch, loss_list, mode):
        with open(os.path.join(self.dir, "%s_loss_list_%d.json" % (mode, epoch)), "w") as fp:
            json.dump(loss_list, fp)

    def loadModelWeight(self, mode):
        path = os.path.join(self.dir, "%s.pth" % mode)
        if not (os.path.isdir(self.dir) and os.path.exists(path)):
            print("==========>Train the model from scratch")
            return
        ck = torch.load(path, map_location=self.device)
        self.model.load_state_dict(ck["model_state_dict"])
        side = os.path.join(self.dir, "preprocess.json")
        if os.path.exists(side):
            from ..preprocessing import process_iwr1843 as _pre
            with open(side) as fp:
                trained = json.load(fp).get("fft_zero_doppler")
            if trained and trained != _pre.ZERO_DOPPLER:
                print("==========>WARNING: these weights were trained with HUPR_FFT_ZERO_DOPPLER=%s, this process runs %s" % (trained, _pre.ZERO_DOPPLER))
        elif getattr(self, "uses_gpu_fft_loader", False):
            from ..preprocessing import process_iwr1843 as _pre
            print("==========>NOTE: no preprocess.json beside these weights: the zero-Doppler convention they were trained with is unknown "
                  "(reference-trained or .npy-trained weights: the reference's rounding residue; a round-3 native run: 'exact'); this "
                  "process feeds them HUPR_FFT_ZERO_DOPPLER=%s" % _pre.ZERO_DOPPLER)
        if not self.args.eval and not getattr(self.args, "pretrained", False):
            print("==========>Load the previous optimizer")
            self.optimizer.load_state_dict(ck["optimizer_state_dict"])      # torch.optim.Adam layout either way
            self.start_epoch = ck["epoch"]
            self.logger.updateBestAcc(ck["accuracy"])
        print("==========>Load the model weight from %s, saved at epoch %d" % (self.dir, ck["epoch"]))
