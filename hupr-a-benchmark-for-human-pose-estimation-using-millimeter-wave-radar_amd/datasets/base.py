"""Loader glue mirrored from the reference's datasets/base.py.

``Normalize`` keeps the reference call shape (a (C,H,W) tensor in, same shape out,
datasets/base.py:13-24) for callers that still iterate slice by slice; the batched, fused
form the training loader uses is ``preprocessing.fft_chain_loader`` /
``preprocessing.loader_normalize``.
"""
import json
import os

import numpy as np
import torch

from ..preprocessing.process_iwr1843 import loader_normalize

JOINT_NAMES = ["R_Hip", "R_Knee", "R_Ankle", "L_Hip", "L_Knee", "L_Ankle", "Neck", "Head", "L_Shoulder", "L_Elbow",
               "L_Wrist", "R_Shoulder", "R_Elbow", "R_Wrist"]
SKELETON = [[14, 13], [13, 12], [11, 10], [10, 9], [9, 7], [12, 9], [8, 7], [7, 1], [7, 4], [6, 5], [5, 4], [3, 2], [2, 1]]


def gt_records(cfg, phase):
    """(annotations, images) of ``<dataDir>/hrnet_annot_<phase>.json`` in COCO-keypoint form, one person per frame:
    image_id = frame + 100000 * sequence, visibility 2 on every joint, area = bbox area / 2, bbox as [x, y, w, h]
    (reference datasets/base.py:56-90)."""
    groups = getattr(cfg.DATASET, phase + "Name")
    with open(os.path.join(cfg.DATASET.dataDir, "hrnet_annot_%s.json" % phase)) as fp:
        per_seq = json.load(fp)
    annotations, images = [], []
    for seq, blocks in zip(groups, per_seq):
        for blk in blocks:
            iid = int(blk["image"][:-4]) + seq * 100000
            x0, y0, x1, y1 = blk["bbox"]
            kp = np.concatenate((np.array(blk["joints"]), np.full((14, 1), 2.0)), axis=1).reshape(-1).tolist()
            annotations.append({"num_keypoints": 14, "area": (x1 - x0) * (y1 - y0) / 2, "iscrowd": 0, "keypoints": kp,
                                "image_id": iid, "bbox": [x0, y0, x1 - x0, y1 - y0], "category_id": 1, "id": iid})
            images.append({"license": -1, "file_name": blk["image"], "coco_url": "None", "height": 256, "width": 256,
                           "date_captured": "None", "flickr_url": "None", "id": iid})
    return annotations, images


def generateGTAnnot(cfg, phase="train"):
    """Write ``<dataDir>/<phase>_gt.json``, the COCO ground-truth file the reference evaluates against
    (datasets/base.py:26-92); returns the dictionary."""
    annotations, images = gt_records(cfg, phase)
    annot = {"info": {"description": "HuPR dataset", "url": "", "version": "1.0", "year": 2022,
                      "contributor": "UW-NYCU-AI-Labs", "date_created": "2022/06/23"},
             "licenses": [], "images": images, "annotations": annotations,
             "categories": [{"supercategory": "person", "id": 1, "name": "person", "keypoints": list(JOINT_NAMES),
                             "skeleton": [list(e) for e in SKELETON]}]}
    with open(os.path.join(cfg.DATASET.dataDir, "%s_gt.json" % phase), "w") as fp:
        json.dump(annot, fp)
    return annot


class Normalize(object):
    """Per-slice form of the reference transform.  The batched kernel normalises the 16 planes of a Doppler slot at once; a
    single (8,64,64) slice is placed in the real AND imaginary plane of slot f = 0 of a staging cube that is allocated once per
    device and reused (only that slot is read back)."""
    _staging = {}

    def __call__(self, radarData):
        """radarData: GPU tensor (C=8,H=64,W=64) real -> same shape, per-channel (x-mean)/std."""
        if tuple(radarData.shape) != (8, 64, 64):
            raise ValueError("Normalize expects an (8,64,64) elevation-major slice")
        hwc = radarData.permute(1, 2, 0).to(torch.float32).contiguous()
        key = (radarData.device.type, radarData.device.index)
        cube = Normalize._staging.get(key)
        if cube is None:
            cube = Normalize._staging[key] = torch.zeros((1, 16, 64, 64, 8), dtype=torch.complex64, device=radarData.device)
        # Doppler slot 4 is loader slot f = 0
        cube[0, 4] = torch.complex(hwc, hwc)
        out = loader_normalize(cube)[0, 0, 0]
        return out.permute(2, 0, 1).contiguous()
