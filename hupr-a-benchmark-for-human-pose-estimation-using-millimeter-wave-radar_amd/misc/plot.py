"""Skeleton overlay of the decoded joints (reference misc/plot.py:14-80) without cv2 / torchvision.

``plotHumanPose(batch_joints, cfg, visDir, imageIdx, bbox)`` writes one PNG per sample to
``<visDir>/single_<seq>/<frame:09d>.png``.  What the reference's calls produce, restated:
  * canvas (:22-31): the camera frame ``../frames/<cfg.TEST.plotImgDir>/single_<seq>/processed/images/<frame:09d>.jpg`` resized to
    256 x 256, passed ALONE through ``torchvision.utils.make_grid(batch_image, nrow, padding, True)`` — for a single image make_grid
    returns the (min-max normalised) image itself, without a border — so the picture is 256 x 256;
  * joints (:41-47): every joint is nevertheless shifted by ``padding`` (= 2) pixels in x and y before it is drawn
    (``joint[0] = x * width + padding + joint[0]`` with x = y = 0), as a ``cv2.circle(radius 2, thickness 2)``: OpenCV draws
    that as the 4-vertex polygon of ``ellipse2Poly`` (a diamond of L1 radius 2) with a 2-pixel pen, i.e. the pixels at L1
    distance 1..3 from the centre;
  * the 14 skeleton edges (:48-64) between the shifted joints and the optional box (:66-74, NOT shifted): ``cv2.line`` of
    thickness 1 = OpenCV's 8-connected LineIterator (Bresenham with the error term of ``bresenham`` below), red / green;
  * saved as RGB (the reference swaps to BGR for cv2.imwrite, :78).
When the camera frame is absent (the RGB frames are not part of the radar dataset) the skeleton is drawn on a black canvas
instead of failing — the only deviation.  Pinned by tests/golden/plot_fixture.npz (a restatement of the same calls in
tests/golden/make_golden.py; cv2 and torchvision are not installed here, so OpenCV's rasterisation is restated from its
documented algorithm, not executed).  Host-side by design: one small image per evaluated sample, never on the training path.
"""
import os

import numpy as np

# joint index pairs of the 14-joint HuPR skeleton, in the reference's drawing order (misc/plot.py:48-62)
EDGES = [(0, 1), (1, 2), (0, 3), (3, 4), (4, 5), (0, 6), (3, 6), (6, 7), (6, 8), (6, 11), (8, 9), (9, 10), (11, 12), (12, 13)]
RED, GREEN = (255, 0, 0), (0, 255, 0)


def bresenham(p0, p1):
    """Pixels of ``cv2.line(p0, p1, thickness=1)`` (LINE_8): OpenCV's LineIterator — dx + 1 points along the major axis, error term
    ``dx - 2 dy``, a diagonal step whenever it is negative."""
    (x0, y0), (x1, y1) = p0, p1
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    sx, sy = (1 if x1 >= x0 else -1), (1 if y1 >= y0 else -1)
    steep = dy > dx
    if steep:
        dx, dy = dy, dx
    err, x, y = dx - 2 * dy, x0, y0
    out = []
    for _ in range(dx + 1):
        out.append((x, y))
        diag = err < 0
        err += (2 * dx - 2 * dy) if diag else (-2 * dy)
        if steep:
            y += sy
            x += sx if diag else 0
        else:
            x += sx
            y += sy if diag else 0
    return out


def joint_marker(x, y):
    """Pixels of ``cv2.circle((x, y), 2, colour, 2)``: L1 distance 1..3 from the centre (see the module docstring)."""
    return [(x + a, y + b) for a in range(-3, 4) for b in range(-3, 4) if 1 <= abs(a) + abs(b) <= 3]


def _put(arr, pixels, colour):
    h, w = arr.shape[:2]
    for x, y in pixels:
        if 0 <= x < w and 0 <= y < h:                     # OpenCV clips to the image
            arr[y, x] = colour


def _canvas(cfg, seq, frame, size):
    from PIL import Image  # lazily: Pillow is needed for --visDir only, never by training / inference
    path = os.path.join("../frames", str(cfg.TEST.plotImgDir), "single_%d" % seq, "processed/images", "%09d.jpg" % frame)
    if not os.path.exists(path):
        return np.zeros((size[1], size[0], 3), dtype=np.uint8)
    img = np.asarray(Image.open(path).convert("RGB").resize(size, Image.BILINEAR), dtype=np.float32) / 255.0   # Resize + ToTensor
    lo, hi = float(img.min()), float(img.max())
    img = (np.clip(img, lo, hi) - lo) / max(hi - lo, 1e-5)                  # make_grid(normalize=True): min-max over the image
    return np.clip(img * 255.0, 0, 255).astype(np.uint8)                    # grid.mul(255).clamp(0, 255).byte()


def render(joints, bbox=None, canvas=None, size=(256, 256), padding=2):
    """-> uint8 (H, W, 3) RGB array: one sample's overlay on ``canvas`` (black when None)."""
    arr = np.zeros((size[1], size[0], 3), dtype=np.uint8) if canvas is None else np.array(canvas, dtype=np.uint8)
    pts = [(int(padding + x), int(padding + y)) for x, y in np.asarray(joints, dtype=np.float64)]
    for x, y in pts:
        _put(arr, joint_marker(x, y), RED)
    for a, b in EDGES:
        _put(arr, bresenham(pts[a], pts[b]), RED)
    if bbox is not None:
        x0, y0, w, h = (float(v) for v in bbox)
        tl, tr = (int(x0), int(y0)), (int(x0 + w), int(y0))
        bl, br = (int(x0), int(y0 + h)), (int(x0 + w), int(y0 + h))
        for p, q in ((tl, tr), (tl, bl), (tr, br), (bl, br)):
            _put(arr, bresenham(p, q), GREEN)
    return arr


def plotHumanPose(batch_joints, cfg, visDir, imageIdx, bbox=None, upsamplingSize=(256, 256), nrow=8, padding=2):
    """batch_joints: (B, 14, 2) image-pixel coordinates (x, y); imageIdx: (B,) ids = frame + 100000 * sequence;
    bbox: optional (B, 4) [x, y, w, h].  Returns the list of files written."""
    from PIL import Image
    written = []
    for j in range(len(batch_joints)):
        iid = imageIdx[j]
        name = "%09d" % (iid.item() if hasattr(iid, "item") else int(iid))
        seq, frame = int(name[:4]), int(name[-4:])
        image_dir = os.path.join(visDir, "single_%d" % seq)
        os.makedirs(image_dir, exist_ok=True)
        arr = render(batch_joints[j], None if bbox is None else bbox[j], _canvas(cfg, seq, frame, tuple(upsamplingSize)),
                     tuple(upsamplingSize), padding)
        path = os.path.join(image_dir, "%09d.png" % frame)
        Image.fromarray(arr).save(path)
        written.append(path)
    return written
