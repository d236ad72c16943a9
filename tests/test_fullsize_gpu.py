"""GPU (-m gpu): properties that hold at ANY size, checked at BASELINE.json's full sizes (B = 32 samples = 512
sensor-frames per step) where the CPU oracle is too slow to run."""
import numpy as np
import pytest
import torch

from hupr_amd import synth

pytestmark = pytest.mark.gpu


def _net(math="f32", seed=1):
    from hupr_amd import functional as F_
    from hupr_amd.config_tree import load_config
    from hupr_amd.models import HuPRNet
    F_.set_math(math)
    cfg = load_config()
    net = HuPRNet(cfg).cuda()
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.hupr_state(seed, gain=1.4).items()})
    return cfg, net


def test_fft_chain_512_frames_independent_and_linear():
    from hupr_amd import preprocessing
    a = torch.from_numpy(synth.adc_cube_int16(31, nframes=16) // 2).cuda().repeat(32, 1, 1, 1, 1)
    a[7] = torch.from_numpy(synth.adc_cube_int16(32)[0] // 2).cuda()           # one distinct frame inside the batch
    ya = preprocessing.fft_chain(a)
    assert ya.shape == (512, 16, 64, 64, 8)
    assert torch.equal(ya[16], ya[0]) and torch.equal(ya[511], ya[15])          # replicas are bit-identical
    single = preprocessing.fft_chain(a[7:8])
    assert torch.equal(single[0], ya[7])                                        # position in the batch is irrelevant
    b = torch.from_numpy(synth.adc_cube_int16(33, nframes=16) // 2).cuda().repeat(32, 1, 1, 1, 1)
    yab = preprocessing.fft_chain(a + b)
    yb = preprocessing.fft_chain(b)
    scale = torch.view_as_real(yab).abs().max()
    assert torch.view_as_real(yab - ya - yb).abs().max() <= 2e-5 * scale
    ld = preprocessing.fft_chain_loader(a)
    flat = ld.reshape(512, 8, 2, 4096, 8)
    keep = [0, 1, 2, 3, 5, 6, 7]                                                # slot 4 = clutter-nulled Doppler bin
    assert flat[:, keep].mean(dim=3).abs().max() < 1e-4
    assert (flat[:, keep].std(dim=3, unbiased=True) - 1).abs().max() < 1e-4
    assert torch.isfinite(ld[:, keep]).all()


@pytest.mark.parametrize("math,tol", [("f32", 2e-5), ("bf16", 5e-3)])
def test_model_batch32_eval_is_per_sample(math, tol):
    """eval-mode outputs of sample i do not depend on the rest of the batch nor on its position (BN uses running
    statistics; every kernel is row-independent) — checked at B = 32 against B = 1 and against a permuted batch.
    In bf16 mode the convolution kernel chosen (128- vs 256-voxel tiles, different fp32 summation order) depends on the
    batch size and encoder activations are stored as bf16, so a last-bit difference can flip a bf16 rounding: the gate
    there is bf16-rounding class and arg-max agreement is required for >= 99 % of the joints."""
    from hupr_amd import functional as F_
    try:
        _, net = _net(math)
        net.eval()
        h, v = (torch.from_numpy(t).cuda() for t in synth.model_inputs(32, 77))
        with torch.no_grad():
            p1, p2 = net(h, v)
            q1, q2 = net(h[5:6].contiguous(), v[5:6].contiguous())
            perm = torch.randperm(32, generator=torch.Generator().manual_seed(3)).cuda()
            r1, r2 = net(h[perm].contiguous(), v[perm].contiguous())
        assert p1.shape == (32, 14, 1, 64, 64) and p2.shape == (32, 1, 14, 64, 64)
        assert torch.isfinite(p1).all() and torch.isfinite(p2).all()
        assert (p1[5] - q1[0]).abs().max() <= tol and (p2[5] - q2[0]).abs().max() <= tol
        assert (p1[perm] - r1).abs().max() <= tol and (p2[perm] - r2).abs().max() <= tol
        am = p2.reshape(32, 14, -1).argmax(-1)
        same = (am[perm] == r2.reshape(32, 14, -1).argmax(-1)).float().mean().item()
        assert same == 1.0 if math == "f32" else same >= 0.99, same
    finally:
        F_.set_math("f32")


def test_model_batch32_train_step_properties():
    """train-mode at the bench batch: finite loss/gradients, BatchNorm running stats move, and the analytic gradient
    agrees with a central finite difference of the loss along a random direction (fp32 pipe)."""
    from hupr_amd.misc import LossComputer
    cfg, net = _net("f32")
    net.train()
    h, v = (torch.from_numpy(t).cuda() for t in synth.model_inputs(32, 78))
    gt = torch.from_numpy(synth.keypoints(32, 79))
    lc = LossComputer(cfg, "cuda")
    rm0 = net.RAradarEncoder.layer1[1].main[1].running_mean.clone()
    loss, *_ = lc.computeLoss(net(h, v), gt, decode=False)
    loss.backward()
    assert torch.isfinite(loss) and 0.5 < loss.item() < 3.0
    assert not torch.equal(rm0, net.RAradarEncoder.layer1[1].main[1].running_mean)
    params = [p for p in net.parameters()]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
    # directional derivative on a subset of tensors (decoder head + one encoder conv + GCN bias)
    names = ["radarDecoder.decoderLayer1.2.weight", "REradarEncoder.layer2.1.main.0.weight", "radarDecoder.gcn.L2.bias"]
    named = dict(net.named_parameters())
    g = torch.Generator().manual_seed(5)
    dirs = {n: torch.randn(named[n].shape, generator=g).cuda() for n in names}
    analytic = sum((named[n].grad * dirs[n]).sum().item() for n in names)
    eps = 2e-3

    def loss_at(sign):
        with torch.no_grad():
            for n in names:
                named[n].add_(sign * eps * dirs[n])
            val = lc.computeLoss(net(h, v), gt, decode=False)[0].item()
            for n in names:
                named[n].sub_(sign * eps * dirs[n])
        return val
    numeric = (loss_at(+1) - loss_at(-1)) / (2 * eps)
    print("directional derivative: analytic %.6f numeric %.6f" % (analytic, numeric))
    assert abs(analytic - numeric) <= 0.05 * abs(numeric) + 2e-3


def test_bf16_and_f32_pipes_agree_at_batch32():
    from hupr_amd import functional as F_
    h, v = (torch.from_numpy(t).cuda() for t in synth.model_inputs(32, 80))
    outs = {}
    try:
        for math in ("f32", "bf16"):
            _, net = _net(math)
            net.eval()
            with torch.no_grad():
                outs[math] = net(h, v)
    finally:
        F_.set_math("f32")
    e1 = (outs["f32"][0] - outs["bf16"][0]).abs().max().item()
    e2 = (outs["f32"][1] - outs["bf16"][1]).abs().max().item()
    agree = (outs["f32"][1].reshape(32, 14, -1).argmax(-1) == outs["bf16"][1].reshape(32, 14, -1).argmax(-1)).float().mean().item()
    print("bf16 vs f32 at B=32: max-abs %.3e / %.3e, argmax agreement %.4f" % (e1, e2, agree))
    assert e1 <= 2e-2 and e2 <= 2e-2 and agree >= 0.9


def test_graph_replay_matches_eager_steps():
    """TrainEngine.capture(): the whole step (FFT loader, forward, loss, backward, Adam with device-side step count) as
    one hipGraph.  Five optimisation steps taken eagerly and taken as 2 eager + 1 capture warm-up + 2 replays must land
    on the same parameters (same kernels, same order; only the Adam bias corrections are evaluated on the device)."""
    from hupr_amd import functional as F_
    from hupr_amd.config_tree import load_config
    from hupr_amd.tools.engine import TrainEngine
    try:
        F_.set_math("bf16")
        cfg = load_config()
        dev = torch.device("cuda", 0)
        B, G = 4, cfg.DATASET.numGroupFrames
        adc_h = torch.from_numpy(synth.adc_cube_int16(31, sensor=0, nframes=B * G)).to(dev)
        adc_v = torch.from_numpy(synth.adc_cube_int16(31, sensor=1, nframes=B * G)).to(dev)
        joints = torch.from_numpy(synth.keypoints(B, 32)).to(dev)
        e1 = TrainEngine(cfg, device=dev, seed=0)
        for _ in range(5):
            l1, _ = e1.train_step_from_adc(adc_h, adc_v, joints)
        e2 = TrainEngine(cfg, device=dev, seed=0)
        for _ in range(2):
            e2.train_step_from_adc(adc_h, adc_v, joints)
        e2.capture(adc_h, adc_v, joints, warmup=1)
        for _ in range(2):
            l2, _ = e2.train_step_from_adc(adc_h, adc_v, joints)
        torch.cuda.synchronize()
        p1 = torch.cat([p.detach().flatten() for p in e1.model.parameters()])
        p2 = torch.cat([p.detach().flatten() for p in e2.model.parameters()])
        assert torch.isfinite(p2).all()
        rel = ((p1 - p2).norm() / p1.norm()).item()
        assert rel <= 1e-5, rel
        assert abs(float(l1.detach()) - float(l2.detach())) <= 1e-4 * abs(float(l1.detach()))
        # the BatchNorm step counters (bumped by one launch per step, also inside the graph) count all five steps
        for eng in (e1, e2):
            nbt = [v for k, v in eng.model.state_dict().items() if k.endswith("num_batches_tracked")]
            assert len(nbt) == 30 and all(int(v) == 5 for v in nbt), [int(v) for v in nbt]
    finally:
        F_.set_math("f32")


def test_two_stream_branches_match_one_stream():
    """The vertical branch on a side stream (forward and backward) must not change a single bit: three optimisation steps
    with functional.TWO_STREAMS on and off land on identical parameters, BatchNorm statistics and losses."""
    from hupr_amd import functional as F_
    from hupr_amd.config_tree import load_config
    from hupr_amd.tools.engine import TrainEngine
    saved = F_.TWO_STREAMS
    try:
        F_.set_math("bf16")
        cfg = load_config()
        dev = torch.device("cuda", 0)
        B, G = 4, cfg.DATASET.numGroupFrames
        adc_h = torch.from_numpy(synth.adc_cube_int16(41, sensor=0, nframes=B * G)).to(dev)
        adc_v = torch.from_numpy(synth.adc_cube_int16(41, sensor=1, nframes=B * G)).to(dev)
        joints = torch.from_numpy(synth.keypoints(B, 42)).to(dev)
        states, losses = [], []
        for two in (True, False, True):
            F_.TWO_STREAMS = two
            eng = TrainEngine(cfg, device=dev, seed=0)
            for _ in range(3):
                loss, _ = eng.train_step_from_adc(adc_h, adc_v, joints)
            torch.cuda.synchronize()
            states.append({k: v.detach().clone() for k, v in eng.model.state_dict().items()})
            losses.append(float(loss.detach()))
        for other in (1, 2):
            assert losses[0] == losses[other], losses
            for k in states[0]:
                assert torch.equal(states[0][k], states[other][k]), k
    finally:
        F_.TWO_STREAMS = saved
        F_.set_math("f32")


def test_two_stream_branches_match_one_stream_at_batch32():
    """The same at the benched batch, where the two branches' kernels really share the chip (at B = 4 most launches are alone on it):
    two optimisation steps on two streams, on one stream, and on two streams again land on identical losses, parameters and
    BatchNorm statistics.  Round 3 regression: the resampling forward, in the packed-fp32 form hipcc gave it, returned wrong sums
    in ~1e-4 of its elements whenever it shared the chip with the other branch's level-3 convolution — a two-stream step was not
    reproducible, the pose fit of test_trained_gpu.py landed somewhere else on every run (scripts/interp_race.py,
    scripts/step_determinism.py)."""
    import pose_fit
    from hupr_amd import functional as F_
    from hupr_amd.config_tree import load_config
    from hupr_amd.tools.engine import TrainEngine
    saved = F_.TWO_STREAMS
    try:
        F_.set_math("bf16")
        cfg = load_config()
        dev = torch.device("cuda", 0)
        h, v, joints = pose_fit.scene_batch(32, np.random.default_rng(5), torch.Generator(device=dev).manual_seed(6), dev)
        states, losses = [], []
        for two in (True, False, True, True):
            F_.TWO_STREAMS = two
            eng = TrainEngine(cfg, device=dev, seed=0, lr=2e-4)
            for _ in range(2):
                loss, _ = eng.train_step(h, v, joints)
            torch.cuda.synchronize()
            states.append({k: t.detach().clone() for k, t in eng.model.state_dict().items()})
            losses.append(float(loss.detach()))
            eng.close()
        for other in (1, 2, 3):
            assert losses[0] == losses[other], losses
            bad = [k for k in states[0] if not torch.equal(states[0][k], states[other][k])]
            assert not bad, (other, bad[:8])
    finally:
        F_.TWO_STREAMS = saved
        F_.set_math("f32")


def test_bf16_path_argmax_agreement_b32():
    """SURVEY 8(d) bf16 gate at scale (2 heads x 448 joints, eval, B = 32): the bf16 path (bf16 matrix pipe + bf16-stored
    activations) against the fp32 parity path on the same weights/inputs.  The random-weight fixture has flat heat-maps
    (values within a few 1e-3 of each other around the peak), so some arg-max flips are unavoidable: every flip must be a
    proven near-tie (the fp32 map at the bf16 arg-max is within 2e-3 of its own maximum, i.e. inside the bf16 heat-map
    error), the agreement rate must stay >= 95 %, and the heat-maps themselves within 1e-2."""
    from hupr_amd import functional as F_
    try:
        _, net = _net("f32")
        net.eval()
        h, v = (torch.from_numpy(t).cuda() for t in synth.model_inputs(32, 77))
        out = {}
        for m in ("f32", "bf16"):
            F_.set_math(m)
            with torch.no_grad():
                p1, p2 = net(h, v)
            out[m] = (p1.reshape(32, 14, -1).float(), p2.reshape(32, 14, -1).float())
        for hd in (0, 1):
            a, b = out["f32"][hd], out["bf16"][hd]
            assert (a - b).abs().max().item() <= 1e-2
            ia, ib = a.argmax(-1), b.argmax(-1)
            same = ia == ib
            assert same.float().mean().item() >= 0.95, same.float().mean().item()
            gap = (a.max(-1).values - a.gather(-1, ib[..., None])[..., 0])[~same]
            assert gap.numel() == 0 or gap.max().item() <= 2e-3, gap.max().item()
    finally:
        F_.set_math("f32")


def test_benched_step_bf16_batch32_from_adc_properties():
    """The exact step bench.py times (bf16 pipe + bf16 activations, B = 32, int16 ADC cubes -> FFT loader -> forward -> BCE x2
    + device arg-max decode -> backward -> Adam): loss within 1 % of the fp32 parity path on the same data and weights, finite
    gradients for every parameter, BatchNorm running statistics moved, every parameter updated, decode tensors well-formed."""
    from hupr_amd import functional as F_
    from hupr_amd.config_tree import load_config
    from hupr_amd.tools.engine import TrainEngine
    cfg = load_config()
    dev = torch.device("cuda", 0)
    B, G = 32, cfg.DATASET.numGroupFrames
    base_h = torch.from_numpy(synth.adc_cube_int16(10, sensor=0, nframes=16)).to(dev)
    base_v = torch.from_numpy(synth.adc_cube_int16(10, sensor=1, nframes=16)).to(dev)
    adc_h = base_h.repeat(B * G // 16, 1, 1, 1, 1).contiguous()
    adc_v = base_v.repeat(B * G // 16, 1, 1, 1, 1).contiguous()
    joints = torch.from_numpy(synth.keypoints(B, 20)).to(dev)
    out = {}
    try:
        for math in ("f32", "bf16"):
            F_.set_math(math)
            eng = TrainEngine(cfg, device=dev, seed=0)
            p0 = torch.cat([p.detach().flatten() for p in eng.model.parameters()]).clone()
            rm0 = eng.model.RAradarEncoder.layer1[1].main[1].running_mean.clone()
            loss, loss2 = eng.train_step_from_adc(adc_h, adc_v, joints, decode="device")
            torch.cuda.synchronize()
            grads = torch.cat([b.flat_grad for b in eng.buckets.buckets])
            p1 = torch.cat([p.detach().flatten() for p in eng.model.parameters()])
            (pi, pm), (gi, gm) = eng.last_decode
            out[math] = dict(loss=float(loss.detach()), loss2=float(loss2.detach()), grads=grads.clone(), moved=(p1 != p0).float().mean().item(),
                             rm=(eng.model.RAradarEncoder.layer1[1].main[1].running_mean - rm0).abs().max().item(), pi=pi.clone(), gi=gi.clone(), gm=gm.clone())
            assert torch.isfinite(grads).all() and torch.isfinite(p1).all() and torch.isfinite(loss)
            assert grads.abs().max() > 0 and out[math]["moved"] > 0.99 and out[math]["rm"] > 0
            assert pi.shape == (B * 14,) and pi.dtype == torch.int32 and int(pi.min()) >= 0 and int(pi.max()) < 4096
            del eng
    finally:
        F_.set_math("f32")
    f, b = out["f32"], out["bf16"]
    rel = abs(b["loss"] - f["loss"]) / f["loss"]
    gcos = torch.nn.functional.cosine_similarity(f["grads"], b["grads"], dim=0).item()
    print("B=32 step from ADC cubes: loss f32 %.5f bf16 %.5f (rel %.2e), gradient cosine %.5f, |g| ratio %.4f" %
          (f["loss"], b["loss"], rel, gcos, (b["grads"].norm() / f["grads"].norm()).item()))
    assert rel <= 1e-2 and abs(b["loss2"] - f["loss2"]) / f["loss2"] <= 1e-2
    assert gcos >= 0.98 and 0.9 <= (b["grads"].norm() / f["grads"].norm()).item() <= 1.1
    # the ground-truth decode does not depend on the pipe: target maxima sit at int(x / 4 + 0.5) (misc/utils.py:37-38)
    assert torch.equal(f["gi"], b["gi"]) and bool((f["gm"] == 1.0).all())
    mu = (joints.float() / 4 + 0.5).long()
    assert torch.equal(f["gi"].long(), (mu[..., 1] * 64 + mu[..., 0]).reshape(-1))


@pytest.mark.parametrize("B", [1, 3, 5])
def test_ragged_batches_train_and_eval(B):
    """the reference's DataLoader has no drop_last, so the last step of an epoch is ragged (115 800 mod 256 = 88;
    odd sizes on small sets).  Any batch size must run through both pipes: finite loss / gradients in train mode, and eval
    outputs that do not depend on which other samples share the batch."""
    from hupr_amd import functional as F_
    from hupr_amd.config_tree import load_config
    from hupr_amd.tools.engine import TrainEngine
    cfg = load_config()
    dev = torch.device("cuda", 0)
    G = cfg.DATASET.numGroupFrames
    adc_h = torch.from_numpy(synth.adc_cube_int16(55, sensor=0, nframes=B * G)).to(dev)
    adc_v = torch.from_numpy(synth.adc_cube_int16(55, sensor=1, nframes=B * G)).to(dev)
    joints = torch.from_numpy(synth.keypoints(B, 56)).to(dev)
    try:
        for math in ("f32", "bf16"):
            F_.set_math(math)
            eng = TrainEngine(cfg, device=dev, seed=0)
            loss, _ = eng.train_step_from_adc(adc_h, adc_v, joints, decode="device")
            torch.cuda.synchronize()
            assert torch.isfinite(loss) and all(torch.isfinite(b.flat_grad).all() for b in eng.buckets.buckets)
            h, v = eng.preprocess(adc_h, adc_v)
            p1, p2 = eng.infer(h, v)
            q1, q2 = eng.infer(h[:1].contiguous(), v[:1].contiguous())
            assert p1.shape == (B, 14, 1, 64, 64) and p2.shape == (B, 1, 14, 64, 64)
            tol = 2e-5 if math == "f32" else 5e-3
            assert (p1[:1] - q1).abs().max().item() <= tol and (p2[:1] - q2).abs().max().item() <= tol
    finally:
        F_.set_math("f32")


def test_benched_forward_batch32_from_adc_matches_the_oracle():
    """VERDICT r2 "weak 4": every reference-derived fixture is B = 2.  Here the benched shape itself — 32 samples whose 2 x 256
    sensor-frames come from int16 ADC cubes through the on-GPU FFT loader, TRAIN-mode forward (BatchNorm on the statistics of the
    32-sample batch) — is compared with the ORACLE (bit-faithful restatement of the reference, pinned by the B = 2 golden files)
    run on the host from the same loader tensors: fp32 path within north_star's 1e-3 with identical arg-max on both heads; the
    bf16 path within its tolerance, and the loss of both within 1e-4 / 1e-2 relative of the oracle's."""
    from hupr_amd import functional as F_, preprocessing
    from hupr_amd.config_tree import load_config
    from hupr_amd.misc import LossComputer
    from hupr_amd.models import HuPRNet
    from oracle import loss as oloss, model as omodel
    cfg = load_config()
    B, G = 32, cfg.DATASET.numGroupFrames
    adc = [torch.from_numpy(np.concatenate([synth.adc_cube_int16(40 + b, sensor=s, nframes=G) for b in range(B)])).cuda() for s in (0, 1)]
    h, v = (preprocessing.fft_chain_loader(a).view(B, G, 8, 2, 64, 64, 8) for a in adc)
    gt = synth.keypoints(B, 41)
    sd = {k: torch.from_numpy(np.array(t)) for k, t in synth.hupr_state(3, gain=1.4).items()}
    with torch.no_grad():
        o1, o2 = omodel.forward(sd, h.cpu(), v.cpu(), train=True)
    ol = oloss.compute_loss((o1, o2), gt)[0].item()
    am = [o.reshape(B, 14, -1).argmax(-1) for o in (o1, o2)]
    try:
        for math, tol, ltol in (("f32", 1e-3, 1e-4), ("bf16", 2e-2, 1e-2)):
            F_.set_math(math)
            net = HuPRNet(cfg).cuda()
            net.load_state_dict(sd)
            F_.invalidate_packed()
            net.train()
            with torch.no_grad():
                p = net(h, v)
                loss = LossComputer(cfg, "cuda").computeLoss(p, torch.from_numpy(gt), decode=False)[0].item()
            e = [(p[i].cpu().float() - (o1, o2)[i]).abs().max().item() for i in (0, 1)]
            same = [(p[i].reshape(B, 14, -1).argmax(-1).cpu() == am[i]).float().mean().item() for i in (0, 1)]
            print("%s path vs oracle, B = 32 train-mode forward from ADC cubes: max-abs %.2e / %.2e, arg-max identical %.4f / %.4f, "
                  "loss %.6f vs %.6f" % (math, e[0], e[1], same[0], same[1], loss, ol))
            assert max(e) <= tol and abs(loss - ol) <= ltol * ol
            if math == "f32":      # identical arg-max; a flip is only acceptable as an exact tie of the oracle's own (flat, random-weight) map
                for i in (0, 1):
                    po, oo = p[i].reshape(B, 14, -1).cpu().float(), (o1, o2)[i].reshape(B, 14, -1)
                    for b, k in zip(*np.nonzero((po.argmax(-1) != oo.argmax(-1)).numpy())):
                        assert (oo[b, k].max() - oo[b, k, po[b, k].argmax()]).item() <= 1e-5, (i, b, k)
                assert min(same) >= 0.99
    finally:
        F_.set_math("f32")
        F_.invalidate_packed()


def test_single_sample_inference_tails_are_bit_identical_and_slicing_is_within_rounding():
    """Config C2 (B = 1 eval, bf16 path): (1) the inference tails that sum the K-sliced convolutions' partial sums themselves
    (functional.infer_tail) give the SAME bits as convolution -> reduce launch -> BatchNorm / PReLU launch; (2) slicing the
    reductions at all changes only fp32 summation order: heat-maps within 2e-3 of the unsliced forward, arg-max identical up to ties inside that tolerance."""
    from hupr_amd import functional as F_
    try:
        _, net = _net("bf16")
        net.eval()
        h, v = (torch.from_numpy(t).cuda() for t in synth.model_inputs(1, 91))
        outs = {}
        for tag, tails, split in (("tails", True, 1), ("reduce", False, 1), ("unsliced", False, 0)):
            F_.INFER_TAILS = tails
            F_.rt.lib().hupr_debug_halo_split_k(split)
            with torch.no_grad():
                outs[tag] = tuple(t.float().clone() for t in net(h, v))
        for i in (0, 1):
            assert torch.equal(outs["tails"][i], outs["reduce"][i])
            assert (outs["tails"][i] - outs["unsliced"][i]).abs().max().item() <= 2e-3
            a, b = outs["tails"][i].reshape(14, -1), outs["unsliced"][i].reshape(14, -1)      # flat random-weight maps: a flip must be a tie
            gap = b.max(-1).values - b.gather(-1, a.argmax(-1)[:, None])[:, 0]
            assert gap.max().item() <= 2e-3
    finally:
        F_.INFER_TAILS = True
        F_.rt.lib().hupr_debug_halo_split_k(1)
        F_.set_math("f32")


def test_inference_constants_follow_weight_updates_eager_and_captured():
    """the host-side inference constants — packed convolution layouts, concatenated MSCSA projection weights,
    the zero-padded head filter — must follow parameter updates made behind torch's version counters (FusedAdam, a replayed
    training graph), in eager calls AND in an inference hipGraph captured earlier: its nodes read the cached buffers (no pack
    launches inside the graph), and ``functional.refresh_packed`` refills exactly those buffers in place."""
    from hupr_amd import functional as F_
    from hupr_amd.models import HuPRNet
    try:
        cfg, net = _net("bf16")
        net.eval()
        h, v = (torch.from_numpy(t).cuda() for t in synth.model_inputs(1, 17))
        with torch.no_grad():
            a0 = tuple(t.clone() for t in net(h, v))                 # fills every cache
            n_const = lambda: len(F_._wc_cache) + len(F_._proj_cache) + len(F_._head_cache)
            n_wc = n_const()
            # 6 concatenations of projection weights (two maps x three levels) + the zero-padded head filter: pack-table entries since
            # round 5 (functional._proj_cat / _head_w16_cached), refreshed in place by the same table launch as the packed layouts
            assert len(F_._proj_cache) >= 6 and len(F_._head_cache) >= 1
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                net(h, v)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = net(h, v)
            g.replay()
            torch.cuda.synchronize()
            assert all(torch.equal(o, a) for o, a in zip(out, a0))
            assert n_const() == n_wc                                 # the capture created nothing
            for p in net.parameters():                               # an update torch's version counters do not see
                p.data.mul_(1.02)
            F_.invalidate_packed()
            F_.refresh_packed(h.device)                              # what keeps a captured inference graph current
            g.replay()
            torch.cuda.synchronize()
            got_graph = tuple(t.clone() for t in out)
            got_eager = net(h, v)
            fresh = HuPRNet(cfg).cuda().eval()
            fresh.load_state_dict(net.state_dict())
            F_.invalidate_packed()
            want = fresh(h, v)
            assert not torch.equal(want[1], a0[1])                   # the update is visible at all
            for i in (0, 1):
                assert torch.equal(got_eager[i], want[i])
                assert torch.equal(got_graph[i], want[i])
        # TrainEngine._replay bumps the epoch itself (its Adam node runs inside the graph): covered by
        # tests/test_engine_dp_gpu.py::test_graph_captures_the_exchange_step + the eval that follows it there
    finally:
        F_.set_math("f32")
        F_.invalidate_packed()
