"""Skeleton overlay of the decoded joints (reference misc/plot.py:14-80), with PIL instead of cv2 / torchvision.

``plotHumanPose(batch_joints, cfg, visDir, imageIdx, bbox)`` writes one PNG per sample to
``<visDir>/single_<seq>/<frame:09d>.png``: the camera frame ``../frames/<cfg.TEST.plotImgDir>/single_<seq>/processed/
images/<frame:09d>.jpg`` resized to 256 x 256 inside the 2-pixel border torchvision's ``make_grid`` draws (so joint
coordinates are offset by the padding exactly like the reference's), red joint discs (radius 2, thickness 2), the 14
skeleton edges in red and the optional ground-truth box in green.  When the camera frame is absent (the RGB frames are not
part of the radar dataset) the skeleton is drawn on a black canvas instead of failing — the only deviation.
Host-side by design: one small image per evaluated sample, never on the training path.
"""
import os

import numpy as np

# joint index pairs of the 14-joint HuPR skeleton, in the reference's drawing order (misc/plot.py:48-62)
EDGES = [(0, 1), (1, 2), (0, 3), (3, 4), (4, 5), (0, 6), (3, 6), (6, 7), (6, 8), (6, 11), (8, 9), (9, 10), (11, 12), (12, 13)]


def _canvas(cfg, seq, frame, size, padding):
    from PIL import Image  # lazily: Pillow is needed for --visDir only, never by training / inference
    path = os.path.join("../frames", str(cfg.TEST.plotImgDir), "single_%d" % seq, "processed/images", "%09d.jpg" % frame)
    grid = Image.new("RGB", (size[0] + 2 * padding, size[1] + 2 * padding), (0, 0, 0))
    if os.path.exists(path):
        grid.paste(Image.open(path).convert("RGB").resize(size, Image.BILINEAR), (padding, padding))
    return grid


def plotHumanPose(batch_joints, cfg, visDir, imageIdx, bbox=None, upsamplingSize=(256, 256), nrow=8, padding=2):
    """batch_joints: (B, 14, 2) image-pixel coordinates (x, y); imageIdx: (B,) ids = frame + 100000 * sequence;
    bbox: optional (B, 4) [x, y, w, h].  Returns the list of files written."""
    from PIL import ImageDraw
    written = []
    for j in range(len(batch_joints)):
        iid = imageIdx[j]
        name = "%09d" % (iid.item() if hasattr(iid, "item") else int(iid))
        seq, frame = int(name[:4]), int(name[-4:])
        image_dir = os.path.join(visDir, "single_%d" % seq)
        os.makedirs(image_dir, exist_ok=True)
        img = _canvas(cfg, seq, frame, tuple(upsamplingSize), padding)
        draw = ImageDraw.Draw(img)
        pts = [(int(padding + x), int(padding + y)) for x, y in np.asarray(batch_joints[j], dtype=np.float64)]
        for x, y in pts:                                   # cv2.circle(radius 2, thickness 2): a ring reaching radius 3
            draw.ellipse((x - 3, y - 3, x + 3, y + 3), outline=(255, 0, 0), width=2)
        for a, b in EDGES:
            draw.line((pts[a], pts[b]), fill=(255, 0, 0), width=1)
        if bbox is not None:
            x0, y0, w, h = (float(v) for v in bbox[j])
            draw.rectangle((int(x0), int(y0), int(x0 + w), int(y0 + h)), outline=(0, 255, 0), width=1)
        path = os.path.join(image_dir, "%09d.png" % frame)
        img.save(path)
        written.append(path)
    return written
