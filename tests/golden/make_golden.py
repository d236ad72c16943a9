"""Generate the golden fixtures in this directory by running the IMPORTED REFERENCE
(/root/reference, read-only, pure Python) on inputs from the build-owned generator
(hupr_amd.synth).  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures are data (inputs are regenerable from seeds; expected outputs are stored):
  fft_seed{0,1,2}.npz, fft_point.npz   RadarObject.generateHeatmap       (process_iwr1843.py:106-173)
  loader_seed0.npz                     Normalize + Doppler select         (datasets/base.py:13-24, dataset.py:144-150)
  model_eval.npz, model_train.npz      HuPRNet fwd (+ autograd bwd)        (models/*.py)
  loss_seed0.npz                       LossComputer/generateTarget/argmax (misc/losses.py, utils.py, metrics.py)
  model_trained.npz                    TRAIN_STEPS reference Adam steps (tools/run.py:71-79, tools/base.py:47) on a fixed
                                       batch from the seed weights -> peaky heat-maps: loss trajectory, eval outputs on
                                       the training batch and on EXTRA unseen samples, arg-max (`python make_golden.py trained`)
  dataset_tiny.npz                     HuPR3D_horivert items / <phase>_gt.json / evaluate on a miniature on-disk tree
                                       (datasets/dataset.py:17-165, datasets/base.py:26-92; `python make_golden.py dataset`)
  oks_eval.json                        COCOeval('keypoints') stats on a synthetic set (misc/coco.py, misc/cocoeval.py)
  contract.json                        state_dict keys/shapes, YAML dump, Runner helper outputs (tools/base.py)
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from hupr_amd import synth            # noqa: E402
from oracle import ref_import         # noqa: E402

STRIDE = 61
DS_STRIDE = 997
MODEL_SEED, INPUT_SEED, KP_SEED, GAIN = 1, 5, 7, 1.4


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_fft():
    ro = ref_import.radar_object()
    for seed in (0, 1, 2):
        iq = synth.adc_cube_int16(seed)
        out = ro.generateHeatmap(synth.adc_cube_complex(iq)[0])
        np.savez_compressed(
            os.path.join(HERE, "fft_seed%d.npz" % seed),
            seed=seed, stride=STRIDE, sample=out.reshape(-1)[::STRIDE].copy(),
            doppler_l2=np.sqrt((np.abs(out) ** 2).sum(axis=(1, 2, 3))),
            sha256=sha(out), input_sha256=sha(iq))
        print("fft seed", seed, out.shape, np.abs(out).max())
    tg = [dict(range_bin=60, doppler_bin=3, az_bin=10, el_bin=2, amp=500.0),
          dict(range_bin=40, doppler_bin=-5, az_bin=50, el_bin=5, amp=300.0)]
    iq = synth.point_target_cube(tg)
    out = ro.generateHeatmap(synth.adc_cube_complex(iq)[0])
    mag = np.abs(out).sum(axis=3)
    pk = np.unravel_index(mag.argmax(), mag.shape)
    np.savez_compressed(os.path.join(HERE, "fft_point.npz"), stride=STRIDE,
                        sample=out.reshape(-1)[::STRIDE].copy(), peak=np.array(pk),
                        doppler_l2=np.sqrt((np.abs(out) ** 2).sum(axis=(1, 2, 3))),
                        sha256=sha(out), input_sha256=sha(iq),
                        targets=json.dumps(tg))
    print("point peak (i, r, a) =", pk)
    return ro


def make_loader(ro):
    _, _, _, Normalize = ref_import.misc_parts()
    import torchvision.transforms as T      # the stub installed by ref_import
    tf = T.Compose([T.ToTensor(), Normalize()])
    iq = synth.adc_cube_int16(0)
    cube = ro.generateHeatmap(synth.adc_cube_complex(iq)[0])
    # reference dataset.py:144-150, one sensor
    full = torch.zeros((8, 2, 64, 64, 8))
    k = 0
    for d in range(16 // 2 - 8 // 2, 16 // 2 + 8 // 2):
        full[k, 0] = tf(cube[d].real).permute(1, 2, 0)
        full[k, 1] = tf(cube[d].imag).permute(1, 2, 0)
        k += 1
    full = full.numpy()
    slice_in = cube[5].real.copy()                       # a healthy Doppler bin
    slice_out = tf(slice_in).permute(1, 2, 0).numpy()
    zero_in = cube[8].imag.copy()                        # the clutter-nulled bin (hazard D.2)
    np.savez_compressed(os.path.join(HERE, "loader_seed0.npz"),
                        slice_in=slice_in, slice_out=slice_out.astype(np.float32),
                        zero_doppler_absmax=np.abs(zero_in).max(), all_absmax=np.abs(cube).max(),
                        stride=STRIDE, full_sample=full.reshape(-1)[::STRIDE].copy(),
                        full_sha256=sha(full))
    print("loader ok; zero-doppler absmax %.3e vs %.3e" % (np.abs(zero_in).max(), np.abs(cube).max()))


def _sample(t, n=64):
    f = t.reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].clone()


def make_model():
    cfg = ref_import.load_cfg()
    models = ref_import.model_module()
    LossComputer, generateTarget, get_max_preds, _ = ref_import.misc_parts()
    net = models.HuPRNet(cfg)
    st = synth.hupr_state(MODEL_SEED, gain=GAIN)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in st.items()}
    hn, vn = synth.model_inputs(2, INPUT_SEED)
    h, v = torch.from_numpy(hn), torch.from_numpy(vn)
    gt = torch.from_numpy(synth.keypoints(2, KP_SEED))
    lc = LossComputer(cfg, "cpu")

    for mode in ("eval", "train"):
        net.load_state_dict(sd)
        net.train(mode == "train")
        for p in net.parameters():
            p.grad = None
        p1, p2 = net(h, v)
        loss, loss2, pred2d, gt2d = lc.computeLoss((p1, p2), gt)
        loss.backward()
        names, gnorm, gsamp = [], [], []
        for n, p in net.named_parameters():
            names.append(n)
            gnorm.append(p.grad.double().norm().item())
            s = _sample(p.grad)
            gsamp.append(np.pad(s.numpy(), (0, 64 - s.numel())))
        am1 = p1.detach().reshape(2, 14, -1).argmax(-1).numpy()
        am2 = p2.detach().reshape(2, 14, -1).argmax(-1).numpy()
        extra = {}
        if mode == "train":   # BN running statistics after one train-mode forward
            post = net.state_dict()
            for k in post:
                if k.endswith("running_mean") or k.endswith("running_var"):
                    extra["stat:" + k] = post[k].numpy().copy()
        np.savez_compressed(
            os.path.join(HERE, "model_%s.npz" % mode),
            model_seed=MODEL_SEED, input_seed=INPUT_SEED, kp_seed=KP_SEED, gain=GAIN,
            heatmap=p1.detach().numpy(), gcn_heatmap=p2.detach().numpy(),
            argmax1=am1, argmax2=am2, loss=loss.item(), loss2=loss2.item(),
            pred2d=pred2d, gt2d=gt2d, grad_names=np.array(names), grad_l2=np.array(gnorm),
            grad_sample=np.stack(gsamp), **extra)
        print(mode, "loss", loss.item(), loss2.item(), "ranges",
              p1.min().item(), p1.max().item(), p2.min().item(), p2.max().item())
        print("   argmax2[0]", am2[0].tolist())

    # loss / targets / argmax in isolation
    tg = np.stack([generateTarget(g, 14, 64, 256)[0] for g in gt])
    nz = np.argwhere(tg > 0)
    np.savez_compressed(os.path.join(HERE, "loss_seed0.npz"), kp_seed=KP_SEED, gt=gt.numpy(),
                        target_nz_index=nz.astype(np.int16), target_nz_value=tg[tg > 0],
                        target_sha256=sha(tg))
    return net, cfg


TRAIN_STEPS, TRAIN_LR, TRAIN_WD, EXTRA, EXTRA_SEED = 120, 1e-4, 1e-4, 8, 9


def make_trained():
    """The random-weight fixture has flat heat-maps (max/mean 1.06: every arg-max is a near-tie, useless for gating a
    reduced-precision path).  Train the reference on CPU the way tools/run.py does (Adam lr 1e-4 / wd 1e-4 = the YAML's
    values, BCE on both heads) for TRAIN_STEPS steps on the fixed B=2 synthetic batch: the eval-mode maps become peaky
    (max/mean 50-100).  Weights are NOT stored (142 MB): the GPU tests regenerate them by running the same steps on the
    fp32 parity path, whose per-step agreement with the reference is what the loss trajectory pins."""
    import time
    cfg = ref_import.load_cfg()
    models = ref_import.model_module()
    LossComputer, _, _, _ = ref_import.misc_parts()
    torch.manual_seed(0)
    net = models.HuPRNet(cfg)
    st = synth.hupr_state(MODEL_SEED, gain=GAIN)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
    hn, vn = synth.model_inputs(2, INPUT_SEED)
    h, v = torch.from_numpy(hn), torch.from_numpy(vn)
    gt = torch.from_numpy(synth.keypoints(2, KP_SEED))
    lc = LossComputer(cfg, "cpu")
    opt = torch.optim.Adam(net.parameters(), lr=TRAIN_LR, betas=(0.9, 0.999), weight_decay=TRAIN_WD)
    losses = []
    t0 = time.time()
    net.train()
    for it in range(TRAIN_STEPS):
        opt.zero_grad()
        p1, p2 = net(h, v)
        loss, loss2, _, _ = lc.computeLoss((p1, p2), gt)
        loss.backward()
        opt.step()
        losses.append((loss.item(), loss2.item()))
        if it % 10 == 0:
            print("  step %d loss %.5f %.5f  (%.0fs)" % (it, loss.item(), loss2.item(), time.time() - t0), flush=True)
    net.eval()
    xh, xv = synth.model_inputs(EXTRA, EXTRA_SEED)
    with torch.no_grad():
        p1, p2 = net(h, v)
        q1, q2 = net(torch.from_numpy(xh), torch.from_numpy(xv))

    def peak(t):
        f = t.reshape(-1, 64 * 64)
        return (f.max(1)[0] / f.mean(1)).numpy()
    names, pnorm = [], []
    for n, p in net.named_parameters():
        names.append(n)
        pnorm.append(p.detach().double().norm().item())
    np.savez_compressed(
        os.path.join(HERE, "model_trained.npz"),
        model_seed=MODEL_SEED, input_seed=INPUT_SEED, kp_seed=KP_SEED, gain=GAIN, steps=TRAIN_STEPS, lr=TRAIN_LR, wd=TRAIN_WD,
        extra=EXTRA, extra_seed=EXTRA_SEED, losses=np.array(losses),
        heatmap=p1.numpy().astype(np.float16), gcn_heatmap=p2.numpy().astype(np.float16),
        argmax1=p1.reshape(2, 14, -1).argmax(-1).numpy(), argmax2=p2.reshape(2, 14, -1).argmax(-1).numpy(),
        max1=p1.reshape(2, 14, -1).max(-1)[0].numpy(), max2=p2.reshape(2, 14, -1).max(-1)[0].numpy(),
        x_argmax1=q1.reshape(EXTRA, 14, -1).argmax(-1).numpy(), x_argmax2=q2.reshape(EXTRA, 14, -1).argmax(-1).numpy(),
        x_max1=q1.reshape(EXTRA, 14, -1).max(-1)[0].numpy(), x_max2=q2.reshape(EXTRA, 14, -1).max(-1)[0].numpy(),
        x_top2=q2.reshape(EXTRA, 14, -1).topk(2, dim=-1)[0].numpy(),
        peak1=peak(p1), peak2=peak(p2), x_peak1=peak(q1), x_peak2=peak(q2),
        param_names=np.array(names), param_l2=np.array(pnorm))
    print("trained fixture: final loss %.5f; median max/mean train-batch %.1f / %.1f, unseen %.1f / %.1f" %
          (losses[-1][0], np.median(peak(p1)), np.median(peak(p2)), np.median(peak(q1)), np.median(peak(q2))))


def make_trained_spread():
    """how far does the REFERENCE drift from ITSELF over the same TRAIN_STEPS fit when only the rounding
    changes?  Re-runs make_trained()'s loop on the imported reference (a) with torch.set_num_threads(1) (different reduction
    order inside ATen's convolutions / BatchNorm), (b) in fp64 (the un-rounded trajectory), (c) with the input batch
    perturbed by one fp32 ulp-scale relative noise (1e-7).  Stores the per-step losses next to the default run's
    (model_trained.npz `losses`) in model_trained_spread.npz; tests/test_trained_gpu.py gates the HIP fp32 path's deviation
    against this spread instead of a flat tolerance.  (`python make_golden.py spread`; ~1 h on 8 cores.)"""
    import time
    cfg = ref_import.load_cfg()
    models = ref_import.model_module()
    LossComputer, _, _, _ = ref_import.misc_parts()
    st = synth.hupr_state(MODEL_SEED, gain=GAIN)
    hn, vn = synth.model_inputs(2, INPUT_SEED)
    gtn = synth.keypoints(2, KP_SEED)
    out = {}

    def fit(tag, dtype, threads, eps):
        if threads:
            torch.set_num_threads(threads)
        torch.manual_seed(0)
        net = models.HuPRNet(cfg)
        net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
        net = net.to(dtype)
        net.radarDecoder.gcn.A = net.radarDecoder.gcn.A.to(dtype)     # plain attribute, not a buffer (gcn_networks.py:44)
        h, v = torch.from_numpy(hn).to(dtype), torch.from_numpy(vn).to(dtype)
        if eps:
            g = torch.Generator().manual_seed(1234)
            h = h * (1 + eps * torch.randn(h.shape, generator=g).to(dtype))
            v = v * (1 + eps * torch.randn(v.shape, generator=g).to(dtype))
        gt = torch.from_numpy(gtn)
        lc = LossComputer(cfg, "cpu")
        if dtype == torch.float64:                      # keep the whole loss in fp64 (the targets are built as fp32)
            bce = lc.bce
            lc.bce = lambda p, t: bce(p, t.to(p.dtype))
        opt = torch.optim.Adam(net.parameters(), lr=TRAIN_LR, betas=(0.9, 0.999), weight_decay=TRAIN_WD)
        losses, t0 = [], time.time()
        net.train()
        for it in range(TRAIN_STEPS):
            opt.zero_grad()
            p1, p2 = net(h, v)
            loss, loss2, _, _ = lc.computeLoss((p1, p2), gt)
            loss.backward()
            opt.step()
            losses.append((loss.item(), loss2.item()))
            if it % 10 == 0:
                print("  [%s] step %d loss %.6f %.6f (%.0fs)" % (tag, it, loss.item(), loss2.item(), time.time() - t0), flush=True)
        out["losses_" + tag] = np.array(losses)
        np.savez_compressed(os.path.join(HERE, "model_trained_spread.npz"), steps=TRAIN_STEPS, **out)

    nt = torch.get_num_threads()
    fit("f64", torch.float64, nt, 0.0)
    fit("eps", torch.float32, nt, 1e-7)
    fit("t1", torch.float32, 1, 0.0)
    base = np.load(os.path.join(HERE, "model_trained.npz"))["losses"]
    for k, v in out.items():
        rel = np.abs(v - base) / base
        print("reference vs reference (%s): rel loss deviation at steps 3/10/30/120: %.2e %.2e %.2e %.2e" %
              (k, rel[:3].max(), rel[:10].max(), rel[:30].max(), rel.max()))


def make_dataset():
    """The reference's HuPR3D_horivert (datasets/dataset.py:17-165) reading the miniature on-disk tree of
    synth.write_tiny_dataset: item dictionaries (network inputs as strided samples + moments, labels), the
    ``train_gt.json`` it generates, and COCOeval on a result file built from perturbed ground truth."""
    import copy
    import tempfile
    cfg = copy.deepcopy(ref_import.load_cfg())
    ds_mod = ref_import.dataset_module()

    class Args:
        sampling_ratio = 1
    out = {}
    with tempfile.TemporaryDirectory() as root:
        synth.write_tiny_dataset(root)
        cfg.DATASET.dataDir = root
        cfg.DATASET.duration = synth.TINY["duration"]
        cfg.DATASET.trainName, cfg.DATASET.valName, cfg.DATASET.testName = (synth.TINY[k] for k in ("trainName", "valName", "testName"))
        ds = ds_mod.getDataset("train", cfg, Args(), random=False)
        assert len(ds) == 12
        out["train_gt_json"] = open(os.path.join(root, "train_gt.json")).read()
        for idx in (0, 1, 4, 5, 6, 11):
            it = ds[idx]
            for sensor in ("hori", "vert"):
                m = it["VRDAEmap_" + sensor].numpy()
                out["i%d_%s_sample" % (idx, sensor)] = m.reshape(-1)[::DS_STRIDE].copy()
                out["i%d_%s_moments" % (idx, sensor)] = np.array([m.astype(np.float64).sum(), (m.astype(np.float64) ** 2).sum()])
                # which source frame fed each window slot: slot means identify the frame (window clamping, :125-139)
                out["i%d_%s_slot_l1" % (idx, sensor)] = np.abs(m.astype(np.float64)).reshape(8, -1).sum(1)
            out["i%d_joints" % idx] = it["jointsGroup"].numpy()
            out["i%d_bbox" % idx] = it["bbox"].numpy()
            out["i%d_imageId" % idx] = np.array(it["imageId"])
        # sampling_ratio arithmetic (:121-124,161-162) with random=False
        Args.sampling_ratio = 3
        ds3 = ds_mod.getDataset("train", cfg, Args(), random=False)
        out["sr3_len"] = np.array(len(ds3))
        out["sr3_imageIds"] = np.array([ds3[i]["imageId"] for i in range(len(ds3))])
        # evaluation protocol: results = ground truth shifted by a deterministic offset -> COCOeval stats (+ per-joint APs)
        Args.sampling_ratio = 1
        dsv = ds_mod.getDataset("val", cfg, Args(), random=False)
        recs = []
        for i in range(len(dsv)):
            ann = dsv.annots[i]
            j = ann["joints"] + synth.uniform((14, 2), -9.0, 9.0, "tiny_shift", i, dtype=np.float64)
            kp = np.concatenate([j, np.ones((14, 1))], axis=1).reshape(-1).tolist()
            recs.append({"category_id": 1, "image_id": int(ann["imageId"]), "score": 1.0, "keypoints": kp})
        logd = os.path.join(root, "logs")
        os.makedirs(logd)
        json.dump(recs, open(os.path.join(logd, "val_results.json"), "w"))
        out["val_results_json"] = json.dumps(recs)
        out["val_ap"] = np.array(dsv.evaluate(logd))
        aps = []
        coco_eval = ds_mod.COCOeval(dsv.coco, dsv.coco.loadRes(os.path.join(logd, "val_results.json")), "keypoints")
        coco_eval.params.useSegm = None
        for k in range(14):
            coco_eval.evaluate(k)
            coco_eval.accumulate()
            coco_eval.summarize()
            aps.append(float(coco_eval.stats[0]))
        out["val_ap_each"] = np.array(aps)
    np.savez_compressed(os.path.join(HERE, "dataset_tiny.npz"), stride=DS_STRIDE, **out)
    print("dataset fixture ok: val AP %.4f, per-joint %s" % (float(out["val_ap"]), np.round(out["val_ap_each"], 3).tolist()))


def make_contract(net, cfg):
    import yaml
    sd = net.state_dict()
    contract = {"state_dict": [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()]}
    with open(os.path.join(ref_import.REF, "config", "mscsa_prgcn.yaml")) as f:
        contract["yaml"] = yaml.safe_load(f)

    # Runner helpers (tools/base.py:49-72,124-147) exercised on a bare BaseRunner
    def go():
        import importlib
        return importlib.import_module("tools.base")
    base = ref_import._with_ref_path(go)

    class Args:
        gpuIDs, seed, dir, visDir, eval = [], 0, "x", "none", True
    r = base.BaseRunner(Args(), cfg)
    preds = synth.uniform((2, 14, 2), 0, 64, "preds").astype(np.float32) * 4.0
    bbox = torch.tensor([[50.0, 60.0, 100.0, 150.0], [10.0, 20.0, 200.0, 90.0]])
    recs = r.saveKeypoints([], preds, bbox, torch.tensor([100123, 1500042]))
    contract["saveKeypoints"] = {"preds": preds.tolist(), "bbox": bbox.tolist(), "records": recs}
    # LR schedule: adjustLR at idxBatch % 2000 == 0 for 3 epochs of 5790 iterations
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=cfg.TRAINING.lr)
    r.optimizer = opt
    lrs = []
    for epoch in range(3):
        for it in range(5790):
            if it % cfg.TRAINING.lrDecayIter == 0:
                r.adjustLR(epoch)
            if it % 1000 == 0:
                lrs.append(opt.param_groups[0]["lr"])
    contract["lr_schedule"] = lrs
    with open(os.path.join(HERE, "contract.json"), "w") as f:
        json.dump(contract, f, indent=0)
    print("contract ok", len(contract["state_dict"]), "keys")


def make_plot():
    """plotHumanPose (misc/plot.py:14-80) on a black camera frame: cv2 and torchvision are not installed here, so this is a
    RESTATEMENT of the reference's call sequence with OpenCV's rasterisation written out from its documented algorithm —
      make_grid(single image, padding 2, normalize=True) -> the image itself, 256 x 256, no border        (:31; torchvision.utils)
      joint += padding; cv2.circle(ndarr, joint, 2, [255, 0, 0], 2)                                        (:41-46)
        thickness > 1 -> EllipseEx: ellipse2Poly with delta 90 for radius 2 = the diamond (x+-2, y), (x, y+-2), drawn as a
        closed poly-line with a 2-pixel pen: the pixels at L1 distance 1..3 from the centre
      cv2.line(ndarr, a, b, [255, 0, 0], 1) for the 14 edges, [0, 255, 0] for the four box sides          (:48-74)
        LINE_8 LineIterator: dx + 1 points along the major axis, err = dx - 2 dy, diagonal step while err < 0
    independent of hupr_amd/misc/plot.py (which is what the fixture pins).  Stored: joints, box, and the red / green pixel sets."""
    import numpy as np
    from hupr_amd import synth

    def line(p0, p1):
        (x0, y0), (x1, y1) = p0, p1
        dx, dy = abs(x1 - x0), abs(y1 - y0)
        sx, sy = (1 if x1 >= x0 else -1), (1 if y1 >= y0 else -1)
        swap = dy > dx
        if swap:
            dx, dy = dy, dx
        err, plus, minus = dx - (dy + dy), dx + dx, -(dy + dy)
        x, y, pts = x0, y0, []
        for _ in range(dx + 1):
            pts.append((x, y))
            mask = err < 0
            err += minus + (plus if mask else 0)
            if swap:
                y += sy
                x += sx if mask else 0
            else:
                x += sx
                y += sy if mask else 0
        return pts

    joints = synth.pose_joints(synth.uniform01(2 * 31, "plot_fixture").reshape(2, 31)).astype(np.float64)
    bbox = np.array([[20., 30., 200., 180.], [60.5, 41.2, 120.7, 190.9]])
    edges = [(0, 1), (1, 2), (0, 3), (3, 4), (4, 5), (0, 6), (3, 6), (6, 7), (6, 8), (6, 11), (8, 9), (9, 10), (11, 12), (12, 13)]
    out = {"joints": joints, "bbox": bbox}
    for b in range(2):
        img = np.zeros((256, 256, 3), np.uint8)
        def put(pts, col):
            for x, y in pts:
                if 0 <= x < 256 and 0 <= y < 256:
                    img[y, x] = col
        jj = [(int(2 + x), int(2 + y)) for x, y in joints[b]]
        for x, y in jj:
            put([(x + u, y + v) for u in range(-3, 4) for v in range(-3, 4) if 1 <= abs(u) + abs(v) <= 3], (255, 0, 0))
        for i, k in edges:
            put(line(jj[i], jj[k]), (255, 0, 0))
        x0, y0, w, h = bbox[b]
        tl, tr, bl, br = (int(x0), int(y0)), (int(x0 + w), int(y0)), (int(x0), int(y0 + h)), (int(x0 + w), int(y0 + h))
        for p, q in ((tl, tr), (tl, bl), (tr, br), (bl, br)):
            put(line(p, q), (0, 255, 0))
        out["red_%d" % b] = np.argwhere((img == (255, 0, 0)).all(-1)).astype(np.int16)
        out["green_%d" % b] = np.argwhere((img == (0, 255, 0)).all(-1)).astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "plot_fixture.npz"), **out)
    print("plot fixture: %d / %d red pixels, %d / %d green" % (len(out["red_0"]), len(out["red_1"]), len(out["green_0"]), len(out["green_1"])))


def make_oks():
    """200-image synthetic GT + detections through the reference's pycocotools fork
    (misc/coco.py + misc/cocoeval.py loaded as a package with an empty ``mask`` submodule)."""
    import importlib.util
    import tempfile
    import types
    np.float = float                     # misc/cocoeval.py:381-382 uses the removed alias
    pkg = types.ModuleType("_refcoco")
    pkg.__path__ = [os.path.join(ref_import.REF, "misc")]
    sys.modules["_refcoco"] = pkg
    sys.modules["_refcoco.mask"] = types.ModuleType("_refcoco.mask")
    mods = {}
    for name in ("coco", "cocoeval"):
        spec = importlib.util.spec_from_file_location("_refcoco." + name, os.path.join(ref_import.REF, "misc", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules["_refcoco." + name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    n_img = 200
    joints = synth.uniform((n_img, 14, 2), 30, 226, "oks_gt").astype(np.float64).round()
    noise = synth.normal((n_img, 14, 2), "oks_noise", dtype=np.float64) * synth.uniform((n_img, 1, 1), 1.0, 14.0, "oks_lvl").astype(np.float64)
    det = np.round((joints + noise) / 4.0) * 4.0                       # detections live on the 4-px heat-map grid
    images, anns, dts, gts = [], [], [], []
    for i in range(n_img):
        iid = 100000 * (1 + i // 50) + i
        x0, y0 = joints[i].min(0)
        x1, y1 = joints[i].max(0)
        bbox = [float(x0), float(y0), float(x1 - x0), float(y1 - y0)]
        kp = np.concatenate([joints[i], np.full((14, 1), 2.0)], 1).reshape(-1).tolist()
        anns.append({"num_keypoints": 14, "area": bbox[2] * bbox[3] / 2, "iscrowd": 0, "keypoints": kp, "image_id": iid,
                     "bbox": bbox, "category_id": 1, "id": iid})
        images.append({"file_name": "%09d.jpg" % i, "height": 256, "width": 256, "id": iid})
        dk = np.concatenate([det[i], np.ones((14, 1))], 1).reshape(-1).tolist()
        dts.append({"category_id": 1, "image_id": iid, "score": 1.0, "keypoints": dk})
        gts.append({"image_id": iid, "keypoints": joints[i].tolist(), "bbox": bbox})
    gtj = {"info": {}, "licenses": [], "images": images, "annotations": anns,
           "categories": [{"supercategory": "person", "id": 1, "name": "person", "keypoints": ["k%d" % k for k in range(14)], "skeleton": []}]}
    with tempfile.TemporaryDirectory() as td:
        gp, dp = os.path.join(td, "gt.json"), os.path.join(td, "dt.json")
        json.dump(gtj, open(gp, "w"))
        json.dump(dts, open(dp, "w"))
        coco = mods["coco"].COCO(gp)
        cdt = coco.loadRes(dp)
        ev = mods["cocoeval"].COCOeval(coco, cdt, "keypoints")
        ev.params.useSegm = None
        ev.evaluate()
        ev.accumulate()
        ev.summarize()
        stats = [float(x) for x in ev.stats]
        ev.evaluate(3)                   # per-keypoint variant used by evaluateEach (dataset.py:48-66)
        ev.accumulate()
        ev.summarize()
        stats_k3 = [float(x) for x in ev.stats]
    with open(os.path.join(HERE, "oks_eval.json"), "w") as f:
        json.dump({"gts": gts, "dts": dts, "stats": stats, "stats_keypoint3": stats_k3}, f)
    print("oks stats", stats[:5])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "plot":
        make_plot()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "oks":
        make_oks()
        sys.exit(0)
    assert ref_import.available(), "reference tree not found"
    if len(sys.argv) > 1 and sys.argv[1] == "trained":
        make_trained()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "spread":
        make_trained_spread()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dataset":
        make_dataset()
        sys.exit(0)
    ro = make_fft()
    make_loader(ro)
    net, cfg = make_model()
    make_contract(net, cfg)
    make_oks()
    make_dataset()
    make_trained()
