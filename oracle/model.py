"""ORACLE (test infrastructure, not product): functional torch-CPU restatement of
``HuPRNet`` forward (autograd supplies backward), keyed by the reference's ``state_dict``
names so the same parameter dictionary drives the oracle and the HIP product.

Follows reference:
  models/networks.py:23-41          forward_chirp (.view reinterpretation!) + forward
  models/chirp_networks.py:11-21    MNet: Conv3d(2->32,(2,1,1),s(2,1,1)) + MaxPool3d((4,1,1))
  models/layers.py:40-70            BasicBlock3D (BN + ReLU)
  models/layers.py:8-38             BasicBlock2D (no BN, PReLU) as used by the decoder
  models/layers.py:186-217          Encoder3D
  models/layers.py:72-184           MultiScaleCrossSelfAttentionPRGCN (+ attention :126-133)
  models/gcn_networks.py:6-64       GCN_layers / PRGCN
Parity pinned by tests/golden/model_*.npz generated from the imported reference.
"""
import torch
import torch.nn.functional as F

# unnormalised skeleton adjacency with self loops (models/layers.py:97-112)
_EDGES = [(0, 1), (0, 3), (1, 2), (3, 4), (4, 5), (6, 7), (6, 8), (8, 9), (9, 10),
          (6, 11), (11, 12), (12, 13)]


def adjacency(dtype=torch.float32):
    A = torch.eye(14, dtype=dtype)
    for i, j in _EDGES:
        A[i, j] = 1.0
        A[j, i] = 1.0
    # the reference matrix is NOT symmetric in two places: row 8 has col 6 but row 6 lacks 8,
    # row 11 has col 6 but row 6 lacks 11  (layers.py:104,106,109)
    A[6, 8] = 0.0
    A[6, 11] = 0.0
    return A


class Ctx:
    """Carries parameters, train/eval flag and (optionally) collects intermediates."""

    def __init__(self, params, train=False, taps=None, update_stats=False):
        self.p = params
        self.train = train
        self.taps = taps          # dict or None: name -> tensor, for bisecting
        self.update_stats = update_stats

    def tap(self, name, t):
        if self.taps is not None:
            self.taps[name] = t.detach().clone()


def _bn(ctx, pre, x):
    p = ctx.p
    if ctx.train:
        rm = p[pre + ".running_mean"] if ctx.update_stats else None
        rv = p[pre + ".running_var"] if ctx.update_stats else None
        return F.batch_norm(x, rm, rv, p[pre + ".weight"], p[pre + ".bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, p[pre + ".running_mean"], p[pre + ".running_var"],
                        p[pre + ".weight"], p[pre + ".bias"], False, 0.1, 1e-5)


def basic_block3d(ctx, pre, x):
    p = ctx.p
    res = _bn(ctx, pre + ".downsample.1", F.conv3d(x, p[pre + ".downsample.0.weight"], None, 1, 1))
    y = F.conv3d(x, p[pre + ".main.0.weight"], None, 1, 1)
    y = F.relu(_bn(ctx, pre + ".main.1", y))
    y = _bn(ctx, pre + ".main.4", F.conv3d(y, p[pre + ".main.3.weight"], None, 1, 1))
    return F.relu(y + res)


def basic_block2d(ctx, pre, x):
    p = ctx.p
    res = F.conv2d(x, p[pre + ".downsample.0.weight"], None, 1, 1)
    y = F.conv2d(x, p[pre + ".main.0.weight"], None, 1, 1)
    y = F.prelu(y, p[pre + ".main.1.weight"])
    y = F.conv2d(y, p[pre + ".main.2.weight"], None, 1, 1)
    return F.prelu(y + res, p[pre + ".relu.weight"])


def mnet(ctx, pre, x):
    """x: (B*G, 2, F, R, A) -> (B*G, 32, 1, R, A)"""
    p = ctx.p
    y = F.conv3d(x, p[pre + ".temporalConvWx1x1.weight"], p[pre + ".temporalConvWx1x1.bias"],
                 (2, 1, 1), 0)
    return F.max_pool3d(y, (x.shape[2] // 2, 1, 1), (x.shape[2] // 2, 1, 1))


def forward_chirp(ctx, hori, vert):
    B, G, Fr, _, R, A, _ = hori.shape
    outs = []
    for name, x in (("RAchirpNet", hori), ("REchirpNet", vert)):
        m = x.mean(dim=6)                                   # (B,G,F,2,R,A)
        v = m.reshape(B * G, 2, Fr, R, A)                    # == .view: (F,2) memory seen as (2,F)
        y = mnet(ctx, name, v).squeeze(2)                    # (B*G,32,R,A)
        y = y.reshape(B, G, -1, R, A).permute(0, 2, 1, 3, 4)  # (B,32,G,R,A)
        ctx.tap(name, y)
        outs.append(y)
    return outs


def encoder3d(ctx, pre, x):
    p = ctx.p
    l1 = F.conv3d(x, p[pre + ".layer1.0.weight"], p[pre + ".layer1.0.bias"], 1, 1)
    l1 = basic_block3d(ctx, pre + ".layer1.1", l1)
    l2 = F.interpolate(l1, scale_factor=0.5, mode="trilinear", align_corners=True)
    l2 = basic_block3d(ctx, pre + ".layer2.1", l2)
    l2 = basic_block3d(ctx, pre + ".layer2.2", l2)
    l3 = F.interpolate(l2, scale_factor=0.5, mode="trilinear", align_corners=True)
    l3 = basic_block3d(ctx, pre + ".layer3.1", l3)
    l3 = basic_block3d(ctx, pre + ".layer3.2", l3)
    m1 = F.conv3d(l1, p[pre + ".l1temporalMerge.weight"]).squeeze(2)
    m2 = F.conv3d(l2, p[pre + ".l2temporalMerge.weight"]).squeeze(2)
    m3 = F.conv3d(l3, p[pre + ".temporalMerge.weight"]).squeeze(2)
    ctx.tap(pre + ".l1", m1)
    ctx.tap(pre + ".l2", m2)
    ctx.tap(pre + ".l3", m3)
    return m1, m2, m3


def attention(k, q, v):
    """S[j,k]=sum_c K[c,j]Q[c,k]; P=softmax over j (keys); out[c,k]=sum_j V[c,j]P[j,k]."""
    b, c, h, w = v.shape
    k = k.reshape(b, c, h * w)
    q = q.reshape(b, c, h * w)
    s = torch.einsum("bij,bik->bjk", k, q)
    out = torch.einsum("bci,bik->bck", v.reshape(b, c, h * w), F.softmax(s, 1))
    return out.reshape(b, c, h, w)


def _level(ctx, lvl, ra, re):
    p = ctx.p
    pre = "radarDecoder."

    def cv(name, x):
        return F.conv2d(x, p[pre + name + "." + str(lvl) + ".weight"])
    ra_cross = attention(cv("phi_cross_hori", ra), cv("theta_cross_vert", re), ra) + ra
    ra_self = attention(cv("phi_self_hori", ra), cv("theta_self_hori", ra), ra)
    re_cross = attention(cv("phi_cross_vert", re), cv("theta_cross_hori", ra), re) + re
    re_self = attention(cv("phi_self_vert", re), cv("theta_self_vert", re), re)
    return [ra_cross, ra_self, re_cross, re_self]


def prgcn(ctx, maps):
    p = ctx.p
    B, K, H, W = maps.shape
    A = adjacency(maps.dtype)
    x = F.interpolate(maps, scale_factor=0.5, mode="bilinear", align_corners=True)
    x = x.reshape(-1, K, (H // 2) * (W // 2)).permute(0, 2, 1)
    for i, name in enumerate(("L1", "L2", "L3")):
        x = torch.matmul(p["radarDecoder.gcn." + name + ".weight"], torch.matmul(x, A)) \
            + p["radarDecoder.gcn." + name + ".bias"]
        if i < 2:
            x = F.relu(x)
    hm = x.permute(0, 2, 1).reshape(-1, K, H // 2, W // 2)
    hm = F.interpolate(hm, scale_factor=2.0, mode="bilinear", align_corners=True)
    return torch.sigmoid(hm).unsqueeze(1)


def decoder(ctx, ra1, ra2, ra3, re1, re2, re3):
    up = lambda t: F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=True)
    pre = "radarDecoder."
    x = torch.cat(_level(ctx, 0, ra3, re3), 1)
    x = basic_block2d(ctx, pre + "decoderLayer3.0", x)
    x = up(basic_block2d(ctx, pre + "decoderLayer3.1", x))
    ctx.tap("dec3", x)
    x = torch.cat([x] + _level(ctx, 1, ra2, re2), 1)
    x = basic_block2d(ctx, pre + "decoderLayer2.0", x)
    x = up(basic_block2d(ctx, pre + "decoderLayer2.1", x))
    ctx.tap("dec2", x)
    x = torch.cat([x] + _level(ctx, 2, ra1, re1), 1)
    x = basic_block2d(ctx, pre + "decoderLayer1.0", x)
    x = basic_block2d(ctx, pre + "decoderLayer1.1", x)
    maps = F.conv2d(x, ctx.p[pre + "decoderLayer1.2.weight"])
    ctx.tap("maps", maps)
    return maps, prgcn(ctx, maps)


def forward(params, hori, vert, train=False, taps=None, update_stats=False):
    """-> (heatmap (B,K,1,H,W), gcn_heatmap (B,1,K,H,W))   [models/networks.py:35-41]"""
    ctx = Ctx(params, train, taps, update_stats)
    ra, re = forward_chirp(ctx, hori, vert)
    ra1, ra2, ra3 = encoder3d(ctx, "RAradarEncoder", ra)
    re1, re2, re3 = encoder3d(ctx, "REradarEncoder", re)
    maps, gcn = decoder(ctx, ra1, ra2, ra3, re1, re2, re3)
    return torch.sigmoid(maps).unsqueeze(2), gcn
