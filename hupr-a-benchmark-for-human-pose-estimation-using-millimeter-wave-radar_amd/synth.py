"""Build-owned, library-version-independent synthetic data generator.

Everything synthetic in this repo (ADC cubes, model inputs, weights, labels)
comes from one counter-based generator: ``u64 = splitmix64(key + index)`` so the
GPU box, the CPU oracle and the golden-fixture script regenerate *identical*
bits without shipping large files and without depending on NumPy/PyTorch RNG
stream stability.  (SURVEY.md section 8(d) "Synthetic inputs".)
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def _key(*parts):
    """Fold integers/strings into one 64-bit key (order-sensitive)."""
    k = np.uint64(0x243F6A8885A308D3)
    for p in parts:
        if isinstance(p, str):
            v = 0
            for ch in p.encode():
                v = (v * 131 + ch) & 0xFFFFFFFFFFFFFFFF
        else:
            v = int(p) & 0xFFFFFFFFFFFFFFFF
        with np.errstate(over="ignore"):
            k = _splitmix64(np.array([(int(k) ^ v) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
    return k


def raw_u64(n, *key):
    """n uint64 words for the stream identified by ``key``."""
    base = _key(*key)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _splitmix64((idx * np.uint64(0xD1342543DE82EF95) + base) & _MASK)


def uniform01(n, *key):
    """float64 uniform in [0, 1) with 53 random bits."""
    return (raw_u64(n, *key) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(shape, lo, hi, *key, dtype=np.float32):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * uniform01(n, *key)).astype(dtype).reshape(shape)


def normal(shape, *key, dtype=np.float32):
    """Standard normal via Box-Muller on two independent streams."""
    n = int(np.prod(shape))
    u1 = uniform01(n, *key, "bm1")
    u2 = uniform01(n, *key, "bm2")
    z = np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(dtype).reshape(shape)


def randint(shape, lo, hi, *key):
    """int64 uniform in [lo, hi)."""
    n = int(np.prod(shape))
    r = raw_u64(n, *key) % np.uint64(hi - lo)
    return (r.astype(np.int64) + lo).reshape(shape)


# ----------------------------------------------------------------------------
# domain objects
# ----------------------------------------------------------------------------
NUM_RX, NUM_CHIRP, NUM_SAMPLE = 4, 192, 256


def adc_cube_int16(seed, seq=0, frame=0, sensor=0, nframes=1):
    """IWR1843 ADC cube(s): int16 I/Q uniform in [-2048, 2047] (12-bit ADC range).

    Returns ``(nframes, 4 rx, 192 chirp, 256 sample, 2 [I,Q])`` int16 — the device
    input layout of ``hupr_fft_chain_*`` (786 432 B per sensor-frame).
    """
    shape = (nframes, NUM_RX, NUM_CHIRP, NUM_SAMPLE, 2)
    return randint(shape, -2048, 2048, "adc", seed, seq, frame, sensor).astype(np.int16)


def adc_cube_complex(iq):
    """int16 I/Q (..., 2) -> complex128 (what the reference's generateHeatmap consumes)."""
    return iq[..., 0].astype(np.float64) + 1j * iq[..., 1].astype(np.float64)


def point_target_cube(targets, noise=0.0, seed=0):
    """Known-answer scene: sum of complex exponentials over (sample, chirp, virtual antenna).

    ``targets`` = list of dict(range_bin, doppler_bin, az_bin, el_bin, amp).  Phase
    progressions follow the reference's demux: chirp c = 3*cc + tx; tx0 -> az
    antennas 0..3, tx2 -> az antennas 4..7, tx1 -> elevation row (az offset 2).
    Returns int16 I/Q ``(1, 4, 192, 256, 2)``.
    """
    s = np.arange(NUM_SAMPLE)[None, None, :]
    cc = np.arange(64)[None, :, None]
    rx = np.arange(NUM_RX)[:, None, None]
    out = np.zeros((NUM_RX, NUM_CHIRP, NUM_SAMPLE), dtype=np.complex128)
    for t in targets:
        fr, fd = t["range_bin"] / 256.0, t["doppler_bin"] / 64.0
        fa, fe = t["az_bin"] / 64.0, t["el_bin"] / 8.0
        base = t.get("amp", 400.0) * np.exp(2j * np.pi * (fr * s + fd * cc))
        for tx in range(3):
            if tx == 0:
                ph = np.exp(2j * np.pi * fa * rx)
            elif tx == 2:
                ph = np.exp(2j * np.pi * fa * (rx + 4))
            else:
                ph = np.exp(2j * np.pi * (fa * (rx + 2) + fe))
            out[:, tx::3, :] += base * ph
    if noise > 0:
        out += noise * (normal(out.shape, "ptn_r", seed, dtype=np.float64)
                        + 1j * normal(out.shape, "ptn_i", seed, dtype=np.float64))
    iq = np.stack([np.clip(np.rint(out.real), -32768, 32767),
                   np.clip(np.rint(out.imag), -32768, 32767)], axis=-1).astype(np.int16)
    return iq[None]


def dca1000_encode(frames):
    """int16 (n_frames, 4, 192, 256, 2) cube -> the flat int16 stream a DCA1000 capture of those frames would hold
    (inverse of the layout getadcDataFromDCA1000 parses, process_iwr1843.py:54-83): per chirp the complex samples run
    [rx0 x 256][rx1 x 256][rx2 x 256][rx3 x 256], and every two of them are stored as [I0, I1, Q0, Q1]."""
    fr = np.asarray(frames, dtype=np.int16)
    n = fr.shape[0]
    z = fr.transpose(0, 2, 1, 3, 4).reshape(n * NUM_CHIRP * NUM_RX * NUM_SAMPLE, 2)      # stream order: frame, chirp, rx, sample
    pairs = z.reshape(-1, 2, 2)                                                           # (pair, sample-in-pair, I/Q)
    return np.ascontiguousarray(pairs.transpose(0, 2, 1)).reshape(-1)                     # I0 I1 Q0 Q1


TINY = {"duration": 6, "trainName": [3, 12], "valName": [7], "testName": [7]}


def tiny_cube(seq, frame, sensor):
    """Deterministic complex64 (16, 64, 64, 8) stand-in for one pre-processed ``.npy`` cube of the reference layout."""
    k = (seq * 1000 + frame) * 2 + sensor
    return (100.0 * (normal((16, 64, 64, 8), "tiny_re", k) + 1j * normal((16, 64, 64, 8), "tiny_im", k))).astype(np.complex64)


def tiny_annotations(seqs, duration):
    """hrnet_annot_<phase>.json content: per sequence a list of {image, joints (14 x [x, y] floats), bbox [x0, y0, x1, y1]}."""
    out = []
    for seq in seqs:
        blocks = []
        for fr in range(duration):
            j = uniform((14, 2), 40.0, 216.0, "tiny_joints", seq, fr, dtype=np.float64)
            x0, y0 = j.min(0) - 6.5
            x1, y1 = j.max(0) + 7.25
            blocks.append({"image": "%09d.jpg" % fr, "joints": j.tolist(), "bbox": [float(x0), float(y0), float(x1), float(y1)]})
        out.append(blocks)
    return out


def write_tiny_dataset(root, cubes=True, raw=False, phases=("train", "val")):
    """A miniature HuPR tree under ``root`` (2 + 1 sequences of TINY["duration"] frames): ``hrnet_annot_<phase>.json`` and,
    per frame and sensor, either the reference's pre-processed ``single_<n>/<sensor>/%09d.npy`` cube (``cubes``) or per
    sequence the raw ``single_<n>/<sensor>/adc_data.bin`` capture (``raw``).  Used by tests/golden/make_golden.py (the
    reference reads it) and by the dataset tests (the product reads the identical files)."""
    import json
    import os
    os.makedirs(root, exist_ok=True)
    for phase in phases:
        seqs = TINY[phase + "Name"]
        with open(os.path.join(root, "hrnet_annot_%s.json" % phase), "w") as fp:
            json.dump(tiny_annotations(seqs, TINY["duration"]), fp)
        for seq in seqs:
            for si, sensor in enumerate(("hori", "vert")):
                d = os.path.join(root, "single_%d" % seq, sensor)
                os.makedirs(d, exist_ok=True)
                if cubes:
                    for fr in range(TINY["duration"]):
                        np.save(os.path.join(d, "%09d.npy" % fr), tiny_cube(seq, fr, si))
                if raw:
                    frames = np.concatenate([adc_cube_int16(77, seq=seq, frame=fr, sensor=si) for fr in range(TINY["duration"])])
                    dca1000_encode(frames).tofile(os.path.join(d, "adc_data.bin"))
    return root


def model_inputs(batch, seed, G=8, F=8, R=64, A=64, E=8):
    """Two (B,G,F,2,R,A,E) fp32 standard-normal cubes (what Normalize emits statistically)."""
    shape = (batch, G, F, 2, R, A, E)
    return normal(shape, "hori", seed), normal(shape, "vert", seed)


def keypoints(batch, seed, K=14, lo=40, hi=216):
    """(B,K,2) int64 joints in 256-px image coordinates."""
    return randint((batch, K, 2), lo, hi, "joints", seed)


# ----------------------------------------------------------------------------
# pose scenes: a LEARNABLE synthetic task (the uniform-noise inputs above carry no information about the labels, so a
# network fitted to them memorises its batch and answers unseen samples with multi-modal maps — useless for gating a
# reduced-precision path on arg-max agreement).  A scene puts one reflector per joint into the (range, azimuth) plane of
# both sensors, at the cell the joint's target heat-map is centred on; the joint's identity is carried by WHEN and in
# WHICH Doppler half the reflector shows: joint k lives in frame k % G of the window and in Doppler slots
# 4 (k // G) .. 4 (k // G) + 3 (the two halves are the two "channels" MNet sees after the reference's .view,
# models/networks.py:26-27).  On top of unit Gaussian noise, like real Normalize output.
# ----------------------------------------------------------------------------
POSE_AMP, POSE_SIGMA = 3.0, 1.25


def pose_scene_blobs(joints, G=8, R=64, A=64, img=256):
    """joints (B,K,2) integer (x, y) image coordinates -> reflector planes (B, G, 2, R, A) fp32 (unit-height Gaussians).
    The exponent of a cell depends on its integer squared distance to the centre only, so the Gaussians are gathered from a table of
    exp(-d2 / 2 sigma^2) over d2 — the very fp64 values the cell-by-cell evaluation produced (the pose fits of tests/ are chaotic: one
    differing bit is another trained network), 25 ms -> 3 ms per batch of 32 on the host, which paced those fits."""
    joints = np.asarray(joints)
    B, K, _ = joints.shape
    mu = (joints.astype(np.int64).astype(np.float64) * (R / img) + 0.5).astype(np.int64)      # the target centres (misc/utils.py:37-38)
    span = max(R, A) + int(np.abs(mu).max()) + 1
    table = np.exp(-np.arange(2 * span * span + 1, dtype=np.float64) / (2.0 * POSE_SIGMA ** 2))
    dy2 = (np.arange(R, dtype=np.int64)[None, None, :] - mu[:, :, 1, None]) ** 2                 # (B, K, R)
    dx2 = (np.arange(A, dtype=np.int64)[None, None, :] - mu[:, :, 0, None]) ** 2                 # (B, K, A)
    out = np.zeros((B, G, 2, R, A), dtype=np.float64)
    for k in range(K):                                   # joints sharing a plane are added in joint order, as always
        out[:, k % G, (k // G) % 2] += table[dy2[:, k, :, None] + dx2[:, k, None, :]]
    return out.astype(np.float32)


def pose_scene_inputs(blobs, noise_h, noise_v, amp=POSE_AMP):
    """noise_* (B,G,F,2,R,A,E) unit normal (numpy arrays or torch tensors), blobs (B,G,2,R,A) of the same kind ->
    the two network inputs: noise + amp * reflectors broadcast over re/im, elevation and the 4 Doppler slots of their half."""
    B, G, F = noise_h.shape[:3]
    half = F // 2
    b = blobs[:, :, :, None, None, :, :, None]                         # (B,G,2,1,1,R,A,1)
    sh = (B, G, 2, half) + tuple(noise_h.shape[3:])
    return ((noise_h.reshape(sh) + amp * b).reshape(noise_h.shape),
            (noise_v.reshape(sh) + amp * b).reshape(noise_v.shape))


# a standing person in 256-px image coordinates, joints in cfg.DATASET.idxToJoints order (R_Hip, R_Knee, R_Ankle, L_Hip, L_Knee,
# L_Ankle, Neck, Head, L_Shoulder, L_Elbow, L_Wrist, R_Shoulder, R_Elbow, R_Wrist), and how far each joint swings around it
POSE_TEMPLATE = np.array([[118, 140], [116, 178], [115, 212], [138, 140], [140, 178], [141, 212], [128, 82], [128, 58],
                          [147, 88], [156, 116], [159, 142], [109, 88], [100, 116], [97, 142]], dtype=np.float64)
POSE_SWING = np.array([3, 7, 11, 3, 7, 11, 3, 4, 4, 9, 14, 4, 9, 14], dtype=np.float64)


def pose_joints(u):
    """u: (B, 3 + 28) uniforms in [0, 1) -> (B, 14, 2) int64 joints: the template skeleton, scaled (0.8 .. 1.15) about its hip
    centre, shifted (+-45 px in x, +-14 px in y) and with every joint swung by up to POSE_SWING px — adjacent joints stay
    adjacent, which is what the PRGCN's skeleton adjacency assumes (models/layers.py:97-112); clipped to the [40, 216) box
    the uniform generator uses."""
    u = np.asarray(u, dtype=np.float64)
    B = u.shape[0]
    centre = np.array([128.0, 140.0])
    scale = 0.8 + 0.35 * u[:, 0]
    shift = np.stack([(u[:, 1] - 0.5) * 90.0, (u[:, 2] - 0.5) * 28.0], 1)
    swing = (u[:, 3:].reshape(B, 14, 2) - 0.5) * 2.0 * POSE_SWING[None, :, None]
    j = centre + (POSE_TEMPLATE[None] - centre) * scale[:, None, None] + shift[:, None, :] + swing
    return np.clip(np.floor(j), 40, 215).astype(np.int64)


def pose_scenes(batch, seed, zero_doppler=None):
    """Deterministic held-out scenes: (hori, vert, joints) numpy, regenerable anywhere from the seed.
    zero_doppler: what Doppler slot f = 4 of both inputs holds — None: noise + the reflectors of its Doppler half (rounds 2-3);
    "noise": unit noise and no reflector, like the real chain, whose clutter-nulled zero-Doppler bin carries no target energy
    (reference Normalize of the rounding residue / this chain's dither); "zero": zeros (round 3's exact clutter removal)."""
    key = 7919 + 104729 * int(seed)
    joints = pose_joints(uniform01(batch * 31, "pose", key).reshape(batch, 31))
    nh, nv = model_inputs(batch, key)
    h, v = pose_scene_inputs(pose_scene_blobs(joints), nh, nv)
    if zero_doppler == "noise":
        h[:, :, 4], v[:, :, 4] = nh[:, :, 4], nv[:, :, 4]
    elif zero_doppler == "zero":
        h[:, :, 4], v[:, :, 4] = 0.0, 0.0
    elif zero_doppler is not None:
        raise ValueError(zero_doppler)
    return h, v, joints


# ----------------------------------------------------------------------------
# HuPRNet parameter inventory (reference state_dict contract)
# ----------------------------------------------------------------------------
def _bn_specs(pre, c):
    return [(pre + ".weight", (c,), "bn_w"), (pre + ".bias", (c,), "bn_b"),
            (pre + ".running_mean", (c,), "bn_rm"), (pre + ".running_var", (c,), "bn_rv"),
            (pre + ".num_batches_tracked", (), "bn_nbt")]


def hupr_param_specs(nf=32, G=8, K=14, H=64):
    """[(state_dict key, shape, kind)] in the reference's registration order.

    Mirrors models/networks.py:17-21, models/chirp_networks.py:15, models/layers.py:81-123,
    190-210 and models/gcn_networks.py:41-43 (names/shapes only)."""
    specs = []
    for s in ("RA", "RE"):
        specs += [(s + "chirpNet.temporalConvWx1x1.weight", (nf, 2, 2, 1, 1), "conv"),
                  (s + "chirpNet.temporalConvWx1x1.bias", (nf,), "bias:4")]
    for s in ("RA", "RE"):
        pre = s + "radarEncoder."
        specs += [(pre + "layer1.0.weight", (2 * nf, nf, 3, 3, 3), "conv"),
                  (pre + "layer1.0.bias", (2 * nf,), "bias:%d" % (nf * 27))]
        for blk, ci, co in (("layer1.1", 2 * nf, 2 * nf), ("layer2.1", 2 * nf, 4 * nf),
                            ("layer2.2", 4 * nf, 4 * nf), ("layer3.1", 4 * nf, 8 * nf),
                            ("layer3.2", 8 * nf, 8 * nf)):
            b = pre + blk
            specs += [(b + ".main.0.weight", (co, ci, 3, 3, 3), "conv")]
            specs += _bn_specs(b + ".main.1", co)
            specs += [(b + ".main.3.weight", (co, co, 3, 3, 3), "conv")]
            specs += _bn_specs(b + ".main.4", co)
            specs += [(b + ".downsample.0.weight", (co, ci, 3, 3, 3), "conv")]
            specs += _bn_specs(b + ".downsample.1", co)
        specs += [(pre + "l1temporalMerge.weight", (2 * nf, 2 * nf, G, 1, 1), "conv"),
                  (pre + "l2temporalMerge.weight", (4 * nf, 4 * nf, G // 2, 1, 1), "conv"),
                  (pre + "temporalMerge.weight", (8 * nf, 8 * nf, G // 4, 1, 1), "conv")]
    pre = "radarDecoder."
    for layer, blocks in (("decoderLayer3", ((32 * nf, 8 * nf), (8 * nf, 4 * nf))),
                          ("decoderLayer2", ((20 * nf, 4 * nf), (4 * nf, 2 * nf))),
                          ("decoderLayer1", ((10 * nf, 2 * nf), (2 * nf, nf)))):
        for i, (ci, co) in enumerate(blocks):
            b = "%s%s.%d" % (pre, layer, i)
            specs += [(b + ".main.0.weight", (co, ci, 3, 3), "conv"),
                      (b + ".main.1.weight", (1,), "prelu"),
                      (b + ".main.2.weight", (co, co, 3, 3), "conv"),
                      (b + ".downsample.0.weight", (co, ci, 3, 3), "conv"),
                      (b + ".relu.weight", (1,), "prelu")]
    specs += [(pre + "decoderLayer1.2.weight", (K, nf, 1, 1), "conv")]
    feat = (H // 2) * (H // 2)
    for l in ("L1", "L2", "L3"):
        specs += [(pre + "gcn.%s.weight" % l, (feat, feat), "gcn"),
                  (pre + "gcn.%s.bias" % l, (feat, K), "gcn")]
    for name in ("phi_cross_hori", "theta_cross_hori", "phi_cross_vert", "theta_cross_vert",
                 "phi_self_hori", "theta_self_hori", "phi_self_vert", "theta_self_vert"):
        for i, c in enumerate((8 * nf, 4 * nf, 2 * nf)):
            specs += [("%s%s.%d.weight" % (pre, name, i), (c, c, 1, 1), "conv")]
    return specs


def hupr_state(seed, gain=1.0, nontrivial_bn=True, **kw):
    """Deterministic parameter dictionary {key: ndarray} with PyTorch-default bounds
    (kaiming-uniform a=sqrt(5) => U(+-1/sqrt(fan_in)); GCN U(+-1/32); PReLU 0.25) times
    ``gain``.  ``nontrivial_bn`` perturbs BN affine/running stats so eval mode is exercised."""
    out = {}
    for name, shape, kind in hupr_param_specs(**kw):
        if kind == "conv":
            fan_in = int(np.prod(shape[1:]))
            b = gain / np.sqrt(fan_in)
            out[name] = uniform(shape, -b, b, "w", seed, name)
        elif kind.startswith("bias:"):
            b = 1.0 / np.sqrt(int(kind.split(":")[1]))
            out[name] = uniform(shape, -b, b, "w", seed, name)
        elif kind == "gcn":
            b = gain / np.sqrt(1024.0)
            out[name] = uniform(shape, -b, b, "w", seed, name)
        elif kind == "prelu":
            out[name] = np.full(shape, 0.25, dtype=np.float32)
        elif kind == "bn_w":
            out[name] = uniform(shape, 0.6, 1.4, "w", seed, name) if nontrivial_bn \
                else np.ones(shape, np.float32)
        elif kind == "bn_b":
            out[name] = uniform(shape, -0.2, 0.2, "w", seed, name) if nontrivial_bn \
                else np.zeros(shape, np.float32)
        elif kind == "bn_rm":
            out[name] = uniform(shape, -0.1, 0.1, "w", seed, name) if nontrivial_bn \
                else np.zeros(shape, np.float32)
        elif kind == "bn_rv":
            out[name] = uniform(shape, 0.5, 1.5, "w", seed, name) if nontrivial_bn \
                else np.ones(shape, np.float32)
        elif kind == "bn_nbt":
            out[name] = np.zeros((), dtype=np.int64)
    return out
