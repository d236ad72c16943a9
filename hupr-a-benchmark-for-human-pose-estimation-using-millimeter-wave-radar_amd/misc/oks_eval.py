"""COCO-style OKS keypoint AP/AR for the 14-joint HuPR skeleton, in NumPy (no pycocotools).

Restates what the reference obtains from its pycocotools fork for ``iouType='keypoints'``:
  * OKS with the reference's 14 per-joint sigmas            misc/cocoeval.py:192-236, :527
  * IoU thresholds .50:.05:.95, 101 recall points, maxDets 20,
    area ranges all / medium / large                          misc/cocoeval.py:516-528
  * greedy score-ordered matching, ignore handling, accumulate and the 10 summary numbers
    (evaluateImg / accumulate / summarize of COCOeval)
  * ground-truth conventions of generateGTAnnot: area = bbox area / 2, visibility 2,
    one person per image                                       datasets/base.py:59-80
  * detection area = keypoint bounding box area (COCO.loadRes) misc/coco.py:306-367

This runs once per epoch on <= a few 10^4 detections: host-side by design.
"""
import numpy as np

SIGMAS = np.array([1.07, .87, .89, 1.07, .87, .89, 1., 1., .79, .72, .62, .79, .72, .62]) / 10.0
IOU_THRS = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
AREA_RNG = [[0 ** 2, 1e5 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
MAX_DETS = 20


def make_gt(image_id, joints, bbox_xywh):
    """Ground-truth record the way generateGTAnnot writes it (visibility 2, area = w*h/2)."""
    joints = np.asarray(joints, dtype=np.float64).reshape(-1, 2)
    kp = np.concatenate([joints, np.full((len(joints), 1), 2.0)], axis=1).reshape(-1)
    x, y, w, h = [float(v) for v in bbox_xywh]
    return {"image_id": int(image_id), "keypoints": kp, "bbox": [x, y, w, h], "area": w * h / 2.0, "iscrowd": 0}


def _dt_area(kp):
    x, y = kp[0::3], kp[1::3]
    return float((x.max() - x.min()) * (y.max() - y.min()))


def oks_matrix(dts, gts, idx_keypoint=-1):
    """dts, gts: lists of records -> OKS matrix (len(dts), len(gts))."""
    out = np.zeros((len(dts), len(gts)))
    var = (SIGMAS * 2) ** 2
    k = len(SIGMAS)
    for j, gt in enumerate(gts):
        g = np.asarray(gt["keypoints"], dtype=np.float64)
        xg, yg, vg = g[0::3], g[1::3], g[2::3]
        k1 = np.count_nonzero(vg > 0)
        bb = gt["bbox"]
        x0, x1 = bb[0] - bb[2], bb[0] + bb[2] * 2
        y0, y1 = bb[1] - bb[3], bb[1] + bb[3] * 2
        for i, dt in enumerate(dts):
            d = np.asarray(dt["keypoints"], dtype=np.float64)
            xd, yd = d[0::3], d[1::3]
            if k1 > 0:
                dx, dy = xd - xg, yd - yg
            else:
                z = np.zeros(k)
                dx = np.maximum(z, x0 - xd) + np.maximum(z, xd - x1)
                dy = np.maximum(z, y0 - yd) + np.maximum(z, yd - y1)
            e = (dx ** 2 + dy ** 2) / var / (gt["area"] + np.spacing(1)) / 2
            if k1 > 0:
                e = e[vg > 0]
            if idx_keypoint != -1:
                e = e[idx_keypoint:idx_keypoint + 1]
            out[i, j] = np.sum(np.exp(-e)) / e.shape[0]
    return out


def _evaluate_image(gts, dts, ious, a_rng):
    """One image, one area range -> dict or None (COCOeval.evaluateImg for a single category)."""
    if len(gts) == 0 and len(dts) == 0:
        return None
    g_ig = np.array([1 if (g.get("iscrowd", 0) or g["area"] < a_rng[0] or g["area"] > a_rng[1]) else 0 for g in gts], dtype=int)
    gtind = np.argsort(g_ig, kind="mergesort")
    gts = [gts[i] for i in gtind]
    g_ig = g_ig[gtind]
    dtind = np.argsort([-d["score"] for d in dts], kind="mergesort")[:MAX_DETS]
    dts = [dts[i] for i in dtind]
    ious = ious[:, gtind] if len(ious) > 0 else ious
    T, G, D = len(IOU_THRS), len(gts), len(dts)
    gtm = np.zeros((T, G))
    dtm = np.zeros((T, D))
    dt_ig = np.zeros((T, D))
    if len(ious) != 0:
        for ti, t in enumerate(IOU_THRS):
            for di in range(D):
                iou = min(t, 1 - 1e-10)
                m = -1
                for gi in range(G):
                    if gtm[ti, gi] > 0 and not gts[gi].get("iscrowd", 0):
                        continue
                    if m > -1 and g_ig[m] == 0 and g_ig[gi] == 1:
                        break
                    if ious[di, gi] < iou:
                        continue
                    iou = ious[di, gi]
                    m = gi
                if m == -1:
                    continue
                dt_ig[ti, di] = g_ig[m]
                dtm[ti, di] = 1 + m          # any non-zero id
                gtm[ti, m] = 1 + di
    areas = np.array([d["area"] for d in dts])
    outside = ((areas < a_rng[0]) | (areas > a_rng[1])).reshape(1, D)
    dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(outside, T, 0)))
    return {"dtMatches": dtm, "dtScores": np.array([d["score"] for d in dts]), "gtIgnore": g_ig, "dtIgnore": dt_ig}


def evaluate_keypoints(gts, dts, idx_keypoint=-1):
    """gts: records with image_id/keypoints(K,2 or flat K*3)/bbox[x,y,w,h] (+area); dts: records with
    image_id/keypoints (flat K*3)/score.  -> the 10 COCO summary numbers
    [AP, AP.5, AP.75, AP(M), AP(L), AR, AR.5, AR.75, AR(M), AR(L)]."""
    g_by, d_by = {}, {}
    for g in gts:
        if "area" in g:                      # already a full COCO-style record (flat x,y,v triplets)
            rec = dict(g, keypoints=np.asarray(g["keypoints"], dtype=np.float64).reshape(-1))
        else:                                # (K,2) joints + xywh box, as the Runner collects them
            rec = make_gt(g["image_id"], np.asarray(g["keypoints"], dtype=np.float64).reshape(-1, 2), g["bbox"])
        g_by.setdefault(int(rec["image_id"]), []).append(rec)
    for d in dts:
        kp = np.asarray(d["keypoints"], dtype=np.float64).reshape(-1)
        rec = {"image_id": int(d["image_id"]), "keypoints": kp, "score": float(d["score"]), "area": _dt_area(kp)}
        d_by.setdefault(rec["image_id"], []).append(rec)
    img_ids = sorted(set(g_by) | set(d_by))
    T, R, A = len(IOU_THRS), len(REC_THRS), len(AREA_RNG)
    precision = -np.ones((T, R, A))
    recall = -np.ones((T, A))
    per_img = {}
    for iid in img_ids:
        g, d = g_by.get(iid, []), d_by.get(iid, [])
        ds = [d[i] for i in np.argsort([-x["score"] for x in d], kind="mergesort")][:MAX_DETS]
        ious = oks_matrix(ds, g, idx_keypoint) if (len(g) and len(ds)) else []
        per_img[iid] = (g, ds, ious)
    for a, rng in enumerate(AREA_RNG):
        E = [e for e in (_evaluate_image(*per_img[iid], rng) for iid in img_ids) if e is not None]
        if not E:
            continue
        scores = np.concatenate([e["dtScores"] for e in E])
        inds = np.argsort(-scores, kind="mergesort")
        dtm = np.concatenate([e["dtMatches"] for e in E], axis=1)[:, inds]
        dt_ig = np.concatenate([e["dtIgnore"] for e in E], axis=1)[:, inds]
        g_ig = np.concatenate([e["gtIgnore"] for e in E])
        npig = np.count_nonzero(g_ig == 0)
        if npig == 0:
            continue
        tps = np.logical_and(dtm, np.logical_not(dt_ig))
        fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
        tp_sum = np.cumsum(tps, axis=1).astype(np.float64)
        fp_sum = np.cumsum(fps, axis=1).astype(np.float64)
        for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
            nd = len(tp)
            rc = tp / npig
            pr = tp / (fp + tp + np.spacing(1))
            q = np.zeros(R)
            recall[t, a] = rc[-1] if nd else 0
            pr = pr.tolist()
            for i in range(nd - 1, 0, -1):
                if pr[i] > pr[i - 1]:
                    pr[i - 1] = pr[i]
            idx = np.searchsorted(rc, REC_THRS, side="left")
            for ri, pi in enumerate(idx):
                if pi < nd:
                    q[ri] = pr[pi]
            precision[t, :, a] = q

    def _ap(a, thr=None):
        s = precision[:, :, a] if thr is None else precision[np.where(np.isclose(IOU_THRS, thr))[0], :, a]
        s = s[s > -1]
        return float(np.mean(s)) if s.size else -1.0

    def _ar(a, thr=None):
        s = recall[:, a] if thr is None else recall[np.where(np.isclose(IOU_THRS, thr))[0], a]
        s = s[s > -1]
        return float(np.mean(s)) if s.size else -1.0

    return [_ap(0), _ap(0, .5), _ap(0, .75), _ap(1), _ap(2), _ar(0), _ar(0, .5), _ar(0, .75), _ar(1), _ar(2)]
