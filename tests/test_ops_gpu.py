"""GPU (-m gpu): every HuPRNet operator kernel, through the C ABI, against the same op in
torch-CPU fp32/fp64 (the oracle's building blocks).  Tolerances are fp32-roundoff class."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 9973)
    return (torch.randn(*shape, generator=g) * scale)


def cl(x):      # NCDHW -> channels-last (B,D,H,W,C) contiguous
    return x.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(x):   # channels-last -> NCDHW
    return x.permute(0, 4, 1, 2, 3).contiguous()


def close(got, ref, tol, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-30
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (rel %.3e)" % (what, err, scale, err / scale)


CONV_CASES = [
    # B, Ci, Co, D, H, W, k, pad, bias
    (2, 32, 64, 4, 16, 16, (3, 3, 3), (1, 1, 1), True),
    (1, 64, 64, 8, 12, 20, (3, 3, 3), (1, 1, 1), False),
    (2, 64, 128, 2, 8, 8, (3, 3, 3), (1, 1, 1), False),
    (1, 128, 256, 2, 16, 16, (3, 3, 3), (1, 1, 1), False),
    (2, 64, 64, 8, 16, 16, (8, 1, 1), (0, 0, 0), False),      # temporal merge
    (2, 128, 128, 4, 8, 8, (4, 1, 1), (0, 0, 0), False),
    (3, 320, 64, 1, 16, 16, (1, 3, 3), (0, 1, 1), False),     # decoder 2-D conv, odd batch
    (2, 64, 32, 1, 16, 16, (1, 3, 3), (0, 1, 1), False),
    (2, 256, 256, 1, 16, 16, (1, 1, 1), (0, 0, 0), False),    # q/k projection
    (2, 32, 16, 1, 16, 16, (1, 1, 1), (0, 0, 0), False),      # padded head
    (1, 32, 32, 1, 7, 9, (1, 3, 3), (0, 1, 1), False),        # ragged M (63 voxels)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case):
    from hupr_amd import functional as F_
    B, Ci, Co, D, H, W, k, pad, has_bias = case
    x = rnd(B, Ci, D, H, W, seed=1)
    w = rnd(Co, Ci, *k, seed=2, scale=(Ci * np.prod(k)) ** -0.5)
    b = rnd(Co, seed=3) if has_bias else None
    xr, wr = x.clone().double().requires_grad_(True), w.clone().double().requires_grad_(True)
    br = b.clone().double().requires_grad_(True) if has_bias else None
    yr = F.conv3d(xr, wr, br, 1, pad)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy.double())
    xg = cl(x).cuda().requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if has_bias else None
    if len(k) == 3 and k[0] == 1 and D == 1:
        wg_in = wg.reshape(Co, Ci, k[1], k[2])     # exercise the Conv2d weight shape
    else:
        wg_in = wg
    y = F_.conv(xg, wg_in, bg, None, pad)
    close(ncdhw(y), yr, 2e-5, "conv fwd")
    y.backward(cl(gy).cuda())
    close(ncdhw(xg.grad), xr.grad, 2e-5, "conv dgrad")
    close(wg.grad, wr.grad, 5e-5, "conv wgrad")
    if has_bias:
        close(bg.grad, br.grad, 2e-5, "conv dbias")


def test_conv_residual_epilogue():
    from hupr_amd import functional as F_
    x, w, r = rnd(2, 64, 1, 16, 16, seed=5), rnd(64, 64, 3, 3, seed=6, scale=0.05), rnd(2, 64, 1, 16, 16, seed=7)
    ref = F.conv2d(x[:, :, 0], w, None, 1, 1) + r[:, :, 0]
    xg, rg = cl(x).cuda().requires_grad_(True), cl(r).cuda().requires_grad_(True)
    y = F_.conv(xg, w.cuda(), None, rg, (0, 1, 1))
    close(ncdhw(y)[:, :, 0], ref, 2e-5, "conv+res")
    y.sum().backward()
    assert torch.equal(rg.grad, torch.ones_like(rg))


GEMM_CASES = [(0, 1, 3, 200, 136, 64), (0, 0, 2, 256, 64, 256), (1, 0, 2, 130, 64, 200), (0, 1, 1, 1024, 1024, 448),
              (0, 0, 2, 1024, 16, 1024), (1, 0, 1, 64, 100, 33), (0, 1, 2, 14, 30, 10)]


@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm_modes(case):
    from hupr_amd import functional as F_
    ta, tb, batch, M, N, K = case
    A = rnd(batch, *((K, M) if ta else (M, K)), seed=8)
    Bm = rnd(batch, *((N, K) if tb else (K, N)), seed=9)
    ref = torch.matmul(A.double().transpose(1, 2) if ta else A.double(), Bm.double().transpose(1, 2) if tb else Bm.double())
    lda, ldb = (M if ta else K), (K if tb else N)
    out = F_.gemm(ta, tb, A.cuda(), Bm.cuda(), M, N, K, lda, ldb, batch, A[0].numel(), Bm[0].numel())
    close(out, ref, 2e-5, "gemm %r" % (case,))
    res = rnd(batch, M, N, seed=10)
    out2 = F_.gemm(ta, tb, A.cuda(), Bm.cuda(), M, N, K, lda, ldb, batch, A[0].numel(), Bm[0].numel(), res=res.cuda())
    close(out2, ref + res.double(), 2e-5, "gemm+res")
    out3 = F_.gemm(ta, tb, A.cuda(), Bm.cuda(), M, N, K, lda, ldb, batch, A[0].numel(), Bm[0].numel(),
                   out=out2.clone(), accumulate=True)
    close(out3, 2 * ref + res.double(), 2e-5, "gemm accumulate")


class _BN:
    def __init__(self, C, seed):
        self.weight = (1 + 0.3 * rnd(C, seed=seed)).cuda().requires_grad_(True)
        self.bias = (0.2 * rnd(C, seed=seed + 1)).cuda().requires_grad_(True)
        self.running_mean = (0.1 * rnd(C, seed=seed + 2)).cuda()
        self.running_var = (1 + 0.2 * rnd(C, seed=seed + 3).abs()).cuda()
        self.num_batches_tracked = torch.zeros((), dtype=torch.long).cuda()
        self.momentum, self.eps = 0.1, 1e-5


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("C,vox", [(64, (2, 4, 8, 8)), (128, (3, 2, 5, 7)), (256, (1, 2, 4, 4))])
def test_bn_relu_and_block_tail(training, C, vox):
    from hupr_amd import functional as F_
    B, D, H, W = vox
    x1, x2 = rnd(B, C, D, H, W, seed=11) * 2 + 0.5, rnd(B, C, D, H, W, seed=12)
    bn1, bn2 = _BN(C, 20), _BN(C, 30)

    def ref_bn(x, bn, rm, rv):
        return F.batch_norm(x, rm, rv, bn.weight.detach().cpu().double().requires_grad_(True) if False else None, None,
                            training, 0.1, 1e-5)
    # reference in fp64 with autograd
    params = []
    for bn in (bn1, bn2):
        params.append((bn.weight.detach().cpu().double().requires_grad_(True), bn.bias.detach().cpu().double().requires_grad_(True),
                       bn.running_mean.cpu().double().clone(), bn.running_var.cpu().double().clone()))
    x1r, x2r = x1.double().requires_grad_(True), x2.double().requires_grad_(True)
    (w1, b1, rm1, rv1), (w2, b2, rm2, rv2) = params
    y1r = F.relu(F.batch_norm(x1r, rm1, rv1, w1, b1, training, 0.1, 1e-5))
    gy = rnd(B, C, D, H, W, seed=13)
    x1g = cl(x1).cuda().requires_grad_(True)
    y1 = F_.BNActFn.apply(x1g, bn1.weight, bn1.bias, bn1, training, True)
    close(ncdhw(y1), y1r, 1e-5, "bn+relu fwd")
    y1r.backward(gy.double())
    y1.backward(cl(gy).cuda())
    close(ncdhw(x1g.grad), x1r.grad, 5e-5, "bn+relu dx")
    close(bn1.weight.grad, w1.grad, 5e-5, "bn dgamma")
    close(bn1.bias.grad, b1.grad, 5e-5, "bn dbeta")
    if training:
        close(bn1.running_mean, rm1, 1e-5, "running_mean")
        close(bn1.running_var, rv1, 1e-5, "running_var")
        assert int(bn1.num_batches_tracked) == 1
    # block tail relu(bn_a(x1) + bn_b(x2))
    for t in (x1r, x2r, w1, b1, w2, b2):
        t.grad = None
    bn1.weight.grad = bn1.bias.grad = None
    (_, _, rm1b, rv1b) = (None, None, bn1.running_mean.cpu().double().clone(), bn1.running_var.cpu().double().clone())
    yr = F.relu(F.batch_norm(x1r, rm1b, rv1b, w1, b1, training, 0.1, 1e-5) + F.batch_norm(x2r, rm2, rv2, w2, b2, training, 0.1, 1e-5))
    x1g2, x2g = cl(x1).cuda().requires_grad_(True), cl(x2).cuda().requires_grad_(True)
    y = F_.BNAddBNReLUFn.apply(x1g2, bn1.weight, bn1.bias, bn1, x2g, bn2.weight, bn2.bias, bn2, training)
    close(ncdhw(y), yr, 1e-5, "tail fwd")
    yr.backward(gy.double())
    y.backward(cl(gy).cuda())
    close(ncdhw(x1g2.grad), x1r.grad, 5e-5, "tail dx1")
    close(ncdhw(x2g.grad), x2r.grad, 5e-5, "tail dx2")
    close(bn2.weight.grad, w2.grad, 5e-5, "tail dgamma2")
    close(bn2.bias.grad, b2.grad, 5e-5, "tail dbeta2")


@pytest.mark.parametrize("ab", [(0.3, 0.7), (1.0, 1.0), (1e-3, 0.999)])
def test_both_losses_and_their_weighted_sum_as_one_node(ab):
    """PairBCEFn: loss = alpha BCE(p1, t) + beta BCE(p2, t) and loss2 in two launches forward / one backward — the same floats as two
    BCEFn nodes combined by torch's scalar arithmetic (what rounds 1-4 ran: four + two launches and three torch-native ones)."""
    from hupr_amd import functional as F_
    alpha, beta = ab
    p1 = torch.sigmoid(rnd(4, 14, 64, 64, seed=610)).cuda()
    p2 = torch.sigmoid(rnd(4, 14, 64, 64, seed=611) * 3).cuda()
    t = torch.sigmoid(rnd(4, 14, 64, 64, seed=612) * 4).cuda()
    a1, a2 = p1.clone().requires_grad_(True), p2.clone().requires_grad_(True)
    l1, l2 = F_.BCEFn.apply(a1, t), F_.BCEFn.apply(a2, t)
    ref = alpha * l1 + beta * l2
    (ref * 1.7).backward()
    b1, b2 = p1.clone().requires_grad_(True), p2.clone().requires_grad_(True)
    n0 = F_.rt.lib().hupr_launch_count()
    loss, loss2 = F_.PairBCEFn.apply(b1, b2, t, alpha, beta)
    (loss * 1.7).backward()
    assert F_.rt.lib().hupr_launch_count() - n0 == 3
    assert torch.equal(loss, ref.detach()) and torch.equal(loss2, l2.detach())
    assert torch.equal(b1.grad, a1.grad) and torch.equal(b2.grad, a2.grad)
    # a gradient through the second output too (loss2 used on its own)
    c1, c2 = p1.clone().requires_grad_(True), p2.clone().requires_grad_(True)
    loss, loss2 = F_.PairBCEFn.apply(c1, c2, t, alpha, beta)
    (loss + 0.5 * loss2).backward()
    d1, d2 = p1.clone().requires_grad_(True), p2.clone().requires_grad_(True)
    l1, l2 = F_.BCEFn.apply(d1, t), F_.BCEFn.apply(d2, t)
    ((alpha * l1 + beta * l2) + 0.5 * l2).backward()
    close(c2.grad, d2.grad, 1e-6, "gradient through both outputs")
    assert torch.equal(c1.grad, d1.grad)


def test_prelu():
    from hupr_amd import functional as F_
    x, a = rnd(2, 1, 16, 16, 64, seed=14), torch.tensor([0.25])
    xr, ar = x.double().requires_grad_(True), a.double().requires_grad_(True)
    yr = F.prelu(xr, ar)
    g = rnd(*x.shape, seed=15)
    yr.backward(g.double())
    xg, ag = x.cuda().requires_grad_(True), a.cuda().requires_grad_(True)
    y = F_.PReLUFn.apply(xg, ag)
    close(y, yr, 1e-6, "prelu fwd")
    y.backward(g.cuda())
    close(xg.grad, xr.grad, 1e-6, "prelu dx")
    close(ag.grad, ar.grad, 1e-5, "prelu dalpha")


@pytest.mark.parametrize("shape,size", [((2, 8, 16, 16, 64), (4, 8, 8)), ((1, 4, 32, 32, 128), (2, 16, 16)),
                                        ((2, 1, 16, 16, 128), (1, 32, 32)), ((2, 1, 64, 64, 16), (1, 32, 32)),
                                        ((2, 1, 32, 32, 16), (1, 64, 64))])
def test_interp_align_corners(shape, size):
    from hupr_amd import functional as F_
    B, D, H, W, C = shape
    x = rnd(B, C, D, H, W, seed=16)
    xr = x.clone().requires_grad_(True)
    if D == 1:
        yr = F.interpolate(xr[:, :, 0], size=size[1:], mode="bilinear", align_corners=True).unsqueeze(2)
    else:
        yr = F.interpolate(xr, size=size, mode="trilinear", align_corners=True)
    g = rnd(*yr.shape, seed=17)
    yr.backward(g)
    xg = cl(x).cuda().requires_grad_(True)
    y = F_.interp(xg, size)
    close(ncdhw(y), yr, 2e-6, "interp fwd")
    y.backward(cl(g).cuda())
    close(ncdhw(xg.grad), xr.grad, 1e-5, "interp bwd")


@pytest.mark.parametrize("B,G,H,C", [(32, 4, 32, 128), (32, 2, 16, 256), (3, 4, 16, 128), (1, 2, 16, 256), (2, 1, 16, 128)])
def test_streaming_merge_weight_gradient_on_wider_maps(B, G, H, C, bf16_math):
    """hupr_tmerge_wgrad_stream_bf16 on the level-2 / level-3 merges (C = 128 / 256: virtual 64-channel frames, output-channel blocks
    spread over the workgroups, partials scattered to (Co, Ci, G) by the reduce kernel) against fp64 on the bf16-rounded operands and
    against the generic kernel; run twice for determinism."""
    from hupr_amd import functional as F_
    L = F_.rt.lib()
    if not L.hupr_tmerge_wgrad_stream_supported(G, H * H, C, C):
        pytest.skip("geometry not covered by the streaming kernel")
    x = rnd(B, G, H, H, C, seed=510).cuda().bfloat16().requires_grad_(True)
    w = rnd(C, C, G, 1, 1, seed=511, scale=(C * G) ** -0.5).cuda().requires_grad_(True)
    dy = rnd(B, 1, H, H, C, seed=512).cuda()

    def dw():
        w.grad = None
        F_.TemporalMergeFn.apply(x, w).backward(dy)
        return w.grad.detach().clone()
    d1, d2 = dw(), dw()
    F_.TMERGE_STREAM = False
    try:
        d0 = dw()
    finally:
        F_.TMERGE_STREAM = True
    ref = torch.einsum("bhwo,bghwc->ocg", dy.to(torch.bfloat16).double().cpu().reshape(B, H, H, C), x.detach().double().cpu()).reshape(C, C, G, 1, 1)
    close(d1, ref, 5e-5, "streaming wgrad vs fp64 (bf16-rounded operands)")
    close(d1, d0, 5e-5, "streaming vs generic wgrad")
    assert torch.equal(d1, d2)


@pytest.mark.parametrize("H,C", [(64, 64), (32, 128), (16, 256)])
@pytest.mark.parametrize("cat", [1, 0])
def test_single_sample_attentions_of_a_level_in_one_launch(H, C, cat, bf16_math):
    """hupr_attn_fwd_bf16in_ld_ws_batch: the four attentions of an MSCSA level (reference models/layers.py:150-163) at B = 1 as one
    split launch + one merge launch — bit-identical to the four launch pairs it replaces (same key shares, same merge), at the three
    levels' shapes, with and without the bf16 concatenation output."""
    from hupr_amd import functional as F_
    if not F_.rt.lib().hupr_attn_fwd_split_ws_bytes(1, H * H, C):
        pytest.skip("the split form does not apply to this shape")
    ra, re = rnd(1, 1, H, H, C, seed=160).cuda(), rnd(1, 1, H, H, C, seed=161).cuda()
    ws = [rnd(C, C, 1, 1, seed=162 + i, scale=C ** -0.5).cuda().requires_grad_(True) for i in range(8)]

    def run():
        with torch.no_grad():
            return [o.clone() for o in F_.MSCSALevelFn.apply(ra, re, cat | 2, *ws)]
    assert F_.ATTN_BATCH
    y1 = run()
    F_.ATTN_BATCH = False
    try:
        y0 = run()
    finally:
        F_.ATTN_BATCH = True
    assert len(y1) == len(y0) == (1 if cat else 4)
    for a, b in zip(y1, y0):
        assert torch.equal(a, b)
    with torch.no_grad():                                     # and against the training form of the node (no key split)
        ref = F_.MSCSALevelFn.apply(ra, re, cat, *ws)
    for a, b in zip(y1, ref):
        close(a.float(), b.float(), 1e-2 if cat else 2e-5, "split + batched vs one-pass attention")


@pytest.fixture(params=["hupr_k_conv_halo_bf16<64, 64>", "hupr_k_conv_halo256m_bf16<2, 16>"])
def level3_aggressor(request):
    """The level-3 convolution at B = 32 that shares the chip with a victim: round 3's aggressor (the 128-voxel kernel, which these
    layers ran on until the end of round 4 and smaller batches still do: hupr_debug_halo_tiles(1) keeps depth-2 layers on it) and the
    kernel they run on now (the 16 x 16 x 32 kernel's 2 x 8 x 16 tile)."""
    from hupr_amd import functional as F_
    L = F_.rt.lib()
    L.hupr_debug_halo_tiles(1 if request.param.startswith("hupr_k_conv_halo_bf16") else 31)
    yield request.param
    L.hupr_debug_halo_tiles(31)


def test_resampling_forward_is_unaffected_by_a_convolution_on_another_stream(level3_aggressor, bf16_math):
    """Round 3 regression (scripts/interp_race.py): the build of hupr_k_interp_fwd that hipcc's SLP vectoriser
    produced returned wrong sums in >90 % of the launches that shared the chip with the level-3 convolution kernel
    (hupr_k_conv_halo_bf16<64, 64>, B = 32) running on another stream — and in none alone.  The library's kernel (scalar FMAs,
    -fno-slp-vectorize) must give the bits of the launch alone in every one of 600 launches beside that convolution."""
    from hupr_amd import functional as F_
    L, rt = F_.rt.lib(), F_.rt
    dev = torch.device("cuda")
    x = torch.randn(32, 4, 32, 32, 128, device=dev).relu().bfloat16()
    B, G, H, W, C = x.shape

    def interp():
        y = torch.empty((B, 2, 16, 16, C), dtype=x.dtype, device=dev)
        rt.check(L.hupr_interp_linear_fwd_bf16act(rt.ptr(x), rt.ptr(y), B, G, H, W, 2, 16, 16, C, C, C, rt.stream()))
        return y
    ref = interp()
    torch.cuda.synchronize()
    x3 = torch.randn(32, 2, 16, 16, 256, device=dev).bfloat16()
    w3 = (torch.randn(256, 256, 3, 3, 3, device=dev) * 0.02).requires_grad_(True)
    side = F_.side_stream(dev)
    bad = 0
    for _ in range(3):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(100):
                F_.conv(x3, w3, None, None, (1, 1, 1))
        outs = [interp() for _ in range(200)]
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, ref)) for o in outs)
    assert bad == 0, "%d of 600 launches differ from the launch alone" % bad


@pytest.mark.parametrize("victim", ["fft_chain", "halo_conv_epilogue", "attention_forward", "attention_backward"])
def test_packed_fp32_kernels_are_unaffected_by_co_resident_kernels(victim, level3_aggressor, bf16_math):
    """VERDICT r3 item 7.  Round 3's corruption needed one compiler-formed packed-fp32 sequence and one
    co-resident kernel, and its mechanism inside the chip was never established.  The library still carries HAND-WRITTEN packed
    fp32 arithmetic — the FFT butterflies (hupr_k_doppler_range / hupr_k_angle: 6 000 v_pk_*_f32), the 4-vector adds of the halo
    convolution epilogue, and since round 4 the score pairs of the attention soft-max (v_pk_fma_f32 / v_pk_add_f32) — two compute
    streams are the library default and data parallel adds an RCCL kernel on a third.  Each of those kernels, on a FIXED input,
    must return the bits of its launch alone in every one of 2 400 launches that share the chip with round 3's aggressor
    (hupr_k_conv_halo_bf16<64, 64>: the level-3 convolution at B = 32) on a second stream AND an all-reduce through
    hupr_allreduce_bucket (the C ABI's RCCL communicator, one rank) on a third."""
    from hupr_amd import functional as F_, preprocessing, synth
    from hupr_amd.tools.distributed import RcclTransport
    L, rt = F_.rt.lib(), F_.rt
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(5)
    if victim == "fft_chain":
        iq = torch.from_numpy(np.concatenate([synth.adc_cube_int16(31, frame=f) for f in range(32)])).cuda()
        ws = torch.empty(L.hupr_fft_chain_ws_bytes(32), dtype=torch.uint8, device=dev)
        run = lambda: preprocessing.fft_chain_loader_means(iq, ws=ws)
    elif victim == "halo_conv_epilogue":
        xa = torch.randn(4, 8, 64, 64, 64, device=dev, generator=gen).bfloat16()
        wa = (torch.randn(64, 64, 3, 3, 3, device=dev, generator=gen) * 0.02).requires_grad_(True)
        def run():
            with torch.no_grad():
                return F_.conv(xa, wa, None, None, (1, 1, 1))
    elif victim == "attention_backward":
        # the dS arithmetic of hupr_k_attn_bwd_dq / hupr_k_attn_bwd_dkv512 on accumulator pairs (compiler-formed v_pk_fma_f32,
        # v_pk_add_f32 with neg modifiers, v_pk_mul_f32 from two-element vectors: round 4, second half)
        k, q, v = (torch.randn(2, 1024, 64, device=dev, generator=gen) for _ in range(3))
        kb, qb, vb = (k * 0.5).bfloat16(), (q * 0.5).bfloat16(), v.bfloat16()
        g32 = torch.randn(2, 1024, 64, device=dev, generator=gen)
        gb = g32.bfloat16()
        o0, l0 = torch.empty(2, 1024, 64, device=dev), torch.empty(2, 1024, device=dev)
        rt.check(L.hupr_attn_fwd_bf16in(rt.ptr(kb), rt.ptr(qb), rt.ptr(vb), rt.ptr(v), rt.ptr(o0), rt.ptr(l0), 2, 1024, 64, rt.stream()))
        def run():
            d = torch.empty(3, 2, 1024, 64, device=dev)
            scr = torch.empty(2, 1024, device=dev)
            rt.check(L.hupr_attn_bwd_bf16in(rt.ptr(kb), rt.ptr(qb), rt.ptr(vb), rt.ptr(gb), rt.ptr(v), rt.ptr(o0), rt.ptr(g32), rt.ptr(l0),
                                            rt.ptr(d[0]), rt.ptr(d[1]), rt.ptr(d[2]), rt.ptr(scr), 2, 1024, 64, 1, rt.stream()))
            return d
    else:
        k, q, v = (torch.randn(2, 1024, 64, device=dev, generator=gen) for _ in range(3))
        kb, qb, vb = (k * 0.5).bfloat16(), (q * 0.5).bfloat16(), v.bfloat16()
        def run():
            out, lse = torch.empty(2, 1024, 64, device=dev), torch.empty(2, 1024, device=dev)
            rt.check(L.hupr_attn_fwd_bf16in(rt.ptr(kb), rt.ptr(qb), rt.ptr(vb), rt.ptr(v), rt.ptr(out), rt.ptr(lse), 2, 1024, 64, rt.stream()))
            return out
    ref = run().clone()
    torch.cuda.synchronize()
    x3 = torch.randn(32, 2, 16, 16, 256, device=dev, generator=gen).bfloat16()
    w3 = (torch.randn(256, 256, 3, 3, 3, device=dev, generator=gen) * 0.02).requires_grad_(True)
    side, third = F_.side_stream(dev), torch.cuda.Stream(device=dev)
    comm = RcclTransport(dev)
    bucket = torch.randn(12 << 20, device=dev, generator=gen)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    try:
        for _ in range(12):
            side.wait_stream(torch.cuda.current_stream())
            third.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(100):
                    F_.conv(x3, w3, None, None, (1, 1, 1))
            for _ in range(4):
                comm.all_reduce(bucket, stream=third)
            for _ in range(200):
                bad += (run() != ref).any().to(torch.int64)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.current_stream().wait_stream(third)
        torch.cuda.synchronize()
    finally:
        comm.close()
    assert bad.item() == 0, "%d of 2 400 launches of %s differ from the launch alone" % (bad.item(), victim)


def test_mnet_front_end():
    from hupr_amd import functional as F_
    B, G = 2, 3
    x = rnd(B, G, 8, 2, 16, 16, 8, seed=18)
    w, b = rnd(32, 2, 2, 1, 1, seed=19, scale=0.5), rnd(32, seed=20, scale=0.5)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    m = x.double().mean(dim=6).reshape(B * G, 2, 8, 16, 16)            # the reference's .view
    yr = F.max_pool3d(F.conv3d(m, wr, br, (2, 1, 1)), (4, 1, 1), (4, 1, 1)).squeeze(2)   # (BG,32,R,A)
    g = rnd(*yr.shape, seed=21)
    yr.backward(g.double())
    wg, bg = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = F_.MNetFn.apply(x.cuda(), wg, bg)                               # (B,G,R,A,32)
    close(y.reshape(B * G, 16, 16, 32).permute(0, 3, 1, 2), yr, 1e-5, "mnet fwd")
    y.backward(g.permute(0, 2, 3, 1).reshape(B, G, 16, 16, 32).contiguous().cuda())
    close(wg.grad, wr.grad, 5e-5, "mnet dW")
    close(bg.grad, br.grad, 5e-5, "mnet dbias")


@pytest.mark.parametrize("N,C,residual", [(256, 256, True), (1024, 128, False), (320, 64, True)])
def test_attention(N, C, residual):
    from hupr_amd import functional as F_
    B = 2
    k, q, v = rnd(B, N, C, seed=22, scale=C ** -0.25), rnd(B, N, C, seed=23, scale=C ** -0.25), rnd(B, N, C, seed=24)
    kr, qr, vr = (t.double().requires_grad_(True) for t in (k, q, v))
    s = torch.einsum("bjc,bkc->bjk", kr, qr)                           # S[j,k]
    outr = torch.einsum("bjc,bjk->bkc", vr, F.softmax(s, 1))           # softmax over keys j
    if residual:
        outr = outr + vr
    g = rnd(B, N, C, seed=25)
    outr.backward(g.double())
    kg, qg, vg = (t.cuda().requires_grad_(True) for t in (k, q, v))
    out = F_.AttentionFn.apply(kg, qg, vg, residual)
    close(out, outr, 2e-5, "attention fwd")
    out.backward(g.cuda())
    close(vg.grad, vr.grad, 5e-5, "attention dV")
    close(qg.grad, qr.grad, 1e-4, "attention dQ")
    close(kg.grad, kr.grad, 1e-4, "attention dK")


def test_attention_large_logits_stable():
    from hupr_amd import functional as F_
    k, q, v = rnd(1, 256, 64, seed=26) * 6, rnd(1, 256, 64, seed=27) * 6, rnd(1, 256, 64, seed=28)
    s = torch.einsum("bjc,bkc->bjk", k.double(), q.double())
    ref = torch.einsum("bjc,bjk->bkc", v.double(), F.softmax(s, 1))
    out = F_.AttentionFn.apply(k.cuda(), q.cuda(), v.cuda(), False)
    assert torch.isfinite(out).all()
    close(out, ref, 1e-4, "attention big logits")


@pytest.mark.parametrize("B,products", [(2, True), (3, True), (32, True), (2, False)])
def test_gcn_layer(B, products, monkeypatch):
    """PRGCN layer forward / backward vs fp64 torch; products: the dedicated kernels of csrc/gcn_products.hip (even, odd and the
    bench's batch) or the generic fp32 engine they replace."""
    from hupr_amd import functional as F_
    from oracle.model import adjacency
    monkeypatch.setattr(F_, "GCN_PRODUCTS", products)
    Fdim, K = 1024, 14
    x = torch.zeros(B, Fdim, 16)
    x[..., :K] = rnd(B, Fdim, K, seed=29)
    w, b = rnd(Fdim, Fdim, seed=30, scale=1 / 32), rnd(Fdim, K, seed=31, scale=1 / 32)
    A = adjacency()
    xr, wr, br = x[..., :K].double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.relu(torch.matmul(wr, torch.matmul(xr, A.double())) + br)
    g = rnd(B, Fdim, K, seed=32)
    yr.backward(g.double())
    xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = F_.GCNLayerFn.apply(xg, wg, bg, A.cuda(), True)
    close(y[..., :K], yr, 2e-5, "gcn fwd")
    assert (y[..., K:] == 0).all()
    g16 = torch.zeros(B, Fdim, 16)
    g16[..., :K] = g
    y.backward(g16.cuda())
    close(xg.grad[..., :K], xr.grad, 5e-5, "gcn dx")
    close(wg.grad, wr.grad, 5e-5, "gcn dW")
    close(bg.grad, br.grad, 5e-5, "gcn dbias")


def test_heads_loss_targets_argmax():
    from hupr_amd import functional as F_, synth
    from oracle import loss as oloss
    B, K, H = 2, 14, 64
    x = torch.zeros(B, H * H, 16)
    x[..., :K] = rnd(B, H * H, K, seed=33) * 2
    xr = x[..., :K].double().requires_grad_(True)
    pr = torch.sigmoid(xr).permute(0, 2, 1)                      # (B,K,HW)
    gt = synth.keypoints(B, 7)
    tg, _ = oloss.batch_targets(gt)
    tgt = torch.from_numpy(tg).reshape(B, K, H * H)
    lr = F.binary_cross_entropy(pr, tgt.double())
    lr.backward()
    xg = x.cuda().requires_grad_(True)
    p = F_.SigmoidHeadFn.apply(xg, K)
    close(p, pr, 1e-6, "sigmoid head")
    t_dev = F_.gaussian_targets(torch.from_numpy(gt).cuda())
    assert torch.equal(t_dev.cpu(), torch.from_numpy(tg)), "targets must be bit-exact"
    loss = F_.BCEFn.apply(p.reshape(B, K, H, H), t_dev)
    assert abs(loss.item() - lr.item()) < 1e-5
    loss.backward()
    close(xg.grad[..., :K], xr.grad, 1e-5, "bce+sigmoid grad")
    assert (xg.grad[..., K:] == 0).all()
    # argmax with ties: first maximum wins; compare with the oracle decode
    hm = rnd(B * K, H * H, seed=34)
    hm[0, 100] = hm[0, 3000] = 50.0
    hm[1] = -1.0
    idx, mx = F_.argmax_rows(hm.cuda())
    ref = hm.numpy().argmax(axis=1)
    assert np.array_equal(idx.cpu().numpy(), ref) and idx[0] == 100 and idx[1] == 0
    from hupr_amd.misc import get_max_preds
    pred, _ = get_max_preds(hm.reshape(B, K, H, H).cuda())
    pred_ref, _ = oloss.argmax_decode(hm.reshape(B, K, H, H).numpy())
    assert np.array_equal(pred, pred_ref)
    # edge joints: clipped / fully outside patches
    edge = np.array([[[0, 0], [255, 255], [-40, 100], [400, 400]] + [[128, 128]] * 10])
    te, _ = oloss.batch_targets(edge)
    assert torch.equal(F_.gaussian_targets(torch.from_numpy(edge).cuda()).cpu(), torch.from_numpy(te))


def test_adam_step_matches_torch():
    from hupr_amd import runtime as rt
    n = 10007
    p, g = rnd(n, seed=35), rnd(n, seed=36)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)
    pd, m, v = p.cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    for step in range(1, 4):
        pr.grad = g.clone() * step
        opt.step()
        gd = (g * step).cuda()
        rt.check(rt.lib().hupr_adam_step_f32(rt.ptr(pd), rt.ptr(gd), rt.ptr(m), rt.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 1e-4,
                                            step, 1.0, rt.stream()))
    close(pd, pr, 1e-6, "adam")


# ---- bf16 matrix pipe (fp32 tensors, bf16 operands, fp32 accumulate): looser, rounding-level gates --------
@pytest.fixture
def bf16_math():
    from hupr_amd import functional as F_
    F_.set_math("bf16")
    yield
    F_.set_math("f32")


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bf16(case, bf16_math):
    """Against an fp64 convolution of the bf16-ROUNDED operands the kernel must agree to fp32-accumulation
    accuracy (this checks indexing/transposes exactly); against unrounded operands to bf16 accuracy."""
    from hupr_amd import functional as F_
    B, Ci, Co, D, H, W, k, pad, has_bias = case
    x = rnd(B, Ci, D, H, W, seed=1)
    w = rnd(Co, Ci, *k, seed=2, scale=(Ci * np.prod(k)) ** -0.5)
    b = rnd(Co, seed=3) if has_bias else None
    gy = rnd(B, Co, D + 2 * pad[0] - k[0] + 1, H + 2 * pad[1] - k[1] + 1, W + 2 * pad[2] - k[2] + 1, seed=4)
    xq, wq, gq = _bf16_round(x), _bf16_round(w), _bf16_round(gy)
    yr = F.conv3d(xq, wq, b.double() if has_bias else None, 1, pad)
    # gradients w.r.t. x and w with rounded co-operands
    xr, wr = xq.clone().requires_grad_(True), wq.clone().requires_grad_(True)
    F.conv3d(xr, wq, None, 1, pad).backward(gq)
    F.conv3d(xq, wr, None, 1, pad).backward(gq)
    xg, wg = cl(x).cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if has_bias else None
    y = F_.conv(xg, wg, bg, None, pad)
    close(ncdhw(y), yr, 2e-5, "bf16 conv fwd")
    y.backward(cl(gy).cuda())
    close(ncdhw(xg.grad), xr.grad, 2e-5, "bf16 conv dgrad")
    close(wg.grad, wr.grad, 5e-5, "bf16 conv wgrad")
    full = F.conv3d(x.double(), w.double(), b.double() if has_bias else None, 1, pad)
    close(ncdhw(y), full, 2e-2, "bf16 conv vs unrounded")


@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm_bf16(case, bf16_math):
    from hupr_amd import functional as F_
    ta, tb, batch, M, N, K = case
    A = rnd(batch, *((K, M) if ta else (M, K)), seed=8)
    Bm = rnd(batch, *((N, K) if tb else (K, N)), seed=9)
    Aq, Bq = _bf16_round(A), _bf16_round(Bm)
    ref = torch.matmul(Aq.transpose(1, 2) if ta else Aq, Bq.transpose(1, 2) if tb else Bq)
    lda, ldb = (M if ta else K), (K if tb else N)
    res = rnd(batch, M, N, seed=10)
    out = F_.gemm(ta, tb, A.cuda(), Bm.cuda(), M, N, K, lda, ldb, batch, A[0].numel(), Bm[0].numel(), res=res.cuda())
    close(out, ref + res.double(), 2e-5, "bf16 gemm %r" % (case,))


def test_attention_bf16(bf16_math):
    from hupr_amd import functional as F_
    B, N, C = 2, 256, 128
    k, q, v = rnd(B, N, C, seed=22, scale=C ** -0.25), rnd(B, N, C, seed=23, scale=C ** -0.25), rnd(B, N, C, seed=24)
    s = torch.einsum("bjc,bkc->bjk", k.double(), q.double())
    ref = torch.einsum("bjc,bjk->bkc", v.double(), F.softmax(s, 1)) + v.double()
    out = F_.AttentionFn.apply(k.cuda(), q.cuda(), v.cuda(), True)
    close(out, ref, 2e-2, "bf16 attention")


@pytest.mark.parametrize("N,C,residual", [(256, 64, True), (512, 64, False), (256, 128, True), (4096, 64, False), (256, 256, True),
                                          (384, 256, False)])
def test_flash_attention_bf16(N, C, residual, bf16_math):
    """Fused attention kernels (bf16 operands): forward and all three gradients vs an fp64 reference."""
    from hupr_amd import functional as F_
    assert F_.rt.lib().hupr_attn_flash_supported(N, C)
    B = 2
    k, q, v = rnd(B, N, C, seed=42, scale=C ** -0.25), rnd(B, N, C, seed=43, scale=C ** -0.25), rnd(B, N, C, seed=44)
    kr, qr, vr = (t.double().requires_grad_(True) for t in (k, q, v))
    s = torch.einsum("bjc,bkc->bjk", kr, qr)
    outr = torch.einsum("bjc,bjk->bkc", vr, F.softmax(s, 1))
    if residual:
        outr = outr + vr
    g = rnd(B, N, C, seed=45)
    outr.backward(g.double())
    kg, qg, vg = (t.cuda().requires_grad_(True) for t in (k, q, v))
    out = F_.AttentionFn.apply(kg, qg, vg, residual)
    close(out, outr, 2e-2, "flash fwd")
    out.backward(g.cuda())
    close(vg.grad, vr.grad, 2e-2, "flash dV")
    close(qg.grad, qr.grad, 3e-2, "flash dQ")
    close(kg.grad, kr.grad, 3e-2, "flash dK")
    # and against the materialised bf16 path (same operand rounding): tighter
    F_.USE_FLASH = False
    try:
        k2, q2, v2 = (t.cuda().requires_grad_(True) for t in (k, q, v))
        out2 = F_.AttentionFn.apply(k2, q2, v2, residual)
        out2.backward(g.cuda())
    finally:
        F_.USE_FLASH = True
    close(out, out2, 1e-2, "flash vs materialised fwd")
    close(vg.grad, v2.grad, 1e-2, "flash vs materialised dV")


@pytest.mark.parametrize("B,N,C", [(1, 4096, 64), (3, 1024, 128), (1, 256, 256), (2, 128, 64)])
def test_attention_split_keys_matches_plain_forward(B, N, C, bf16_math):
    """Small batches walk the keys in shares (blockIdx.z) and merge them in a second launch; against the plain kernel on the same
    bf16 operands (output, bf16 copy, log-sum-exp) — the shares round P relative to their own running maxima, hence a tolerance —
    and against fp64."""
    from hupr_amd import functional as F_
    L = F_.rt.lib()
    k, q, v = (rnd(B, N, C, seed=80 + i, scale=(C ** -0.25 if i < 2 else 1.0)).cuda() for i in range(3))
    kb, qb, vb = (t.bfloat16() for t in (k, q, v))
    assert (L.hupr_attn_fwd_split_ws_bytes(B, N, C) > 0) == (B == 1)          # default policy: single-sample calls only
    res = {}
    try:
        L.hupr_debug_attn_split(1)                                          # every grid below 128 workgroups
        nbytes = L.hupr_attn_fwd_split_ws_bytes(B, N, C)
        assert nbytes > 0 and L.hupr_attn_fwd_split_ws_bytes(32, 4096, 64) == 0
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        for name, w in (("plain", None), ("split", ws)):
            out, lse = torch.full((B, N, C), float("nan"), device="cuda"), torch.full((B, N), float("nan"), device="cuda")
            o16 = torch.zeros((B, N, 2 * C), dtype=torch.bfloat16, device="cuda")
            F_.rt.check(L.hupr_attn_fwd_bf16in_ld_ws(F_.rt.ptr(kb), C, F_.rt.ptr(qb), C, F_.rt.ptr(vb), F_.rt.ptr(v), F_.rt.ptr(out),
                                                     F_.rt.ptr(lse), o16.data_ptr() + C * 2, 2 * C, B, N, C,
                                                     F_.rt.ptr(w) if w is not None else None, nbytes if w is not None else 0,
                                                     F_.rt.stream()))
            res[name] = (out, lse, o16)
    finally:
        L.hupr_debug_attn_split(0)
    sref = torch.einsum("bjc,bkc->bjk", kb.double(), qb.double())
    ref = torch.einsum("bjc,bjk->bkc", vb.double(), F.softmax(sref, 1)) + v.double()
    for name in res:
        close(res[name][0], ref, 1e-2, name + " vs fp64")
        close(res[name][1], torch.logsumexp(sref, 1), 1e-5, name + " lse")
        assert torch.equal(res[name][2][..., C:], res[name][0].bfloat16()) and not res[name][2][..., :C].any()
    close(res["split"][0], res["plain"][0], 4e-3, "split vs plain")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eval_batchnorm_single_launch_matches_two_launches(dtype):
    """Inference (eval mode under no_grad): coefficients + apply in one launch (hupr_bn_eval_act_*) against the coefficient
    launch followed by the apply launch — same arithmetic, bit for bit; both block tails of BasicBlock3D."""
    from hupr_amd import functional as F_
    C = 64
    bns = []
    for i in range(2):
        bn = torch.nn.BatchNorm3d(C).cuda().eval()
        with torch.no_grad():
            bn.weight.copy_(rnd(C, seed=90 + i) + 1.0); bn.bias.copy_(rnd(C, seed=92 + i))
            bn.running_mean.copy_(rnd(C, seed=94 + i)); bn.running_var.copy_(rnd(C, seed=96 + i).abs() + 0.3)
        bns.append(bn)
    x1, x2 = (rnd(3, 2, 8, 8, C, seed=98 + i).cuda().to(dtype) for i in range(2))
    with torch.no_grad():
        a = F_.BNActFn.apply(x1, bns[0].weight, bns[0].bias, bns[0], False, True, True)
        b = F_.BNActFn.apply(x1, bns[0].weight, bns[0].bias, bns[0], False, True, False)
        c = F_.BNAddBNReLUFn.apply(x1, bns[0].weight, bns[0].bias, bns[0], x2, bns[1].weight, bns[1].bias, bns[1], False, True)
        d = F_.BNAddBNReLUFn.apply(x1, bns[0].weight, bns[0].bias, bns[0], x2, bns[1].weight, bns[1].bias, bns[1], False, False)
    assert torch.equal(a, b) and torch.equal(c, d) and a.dtype == dtype
    ref = torch.relu(bns[0](x1.float().permute(0, 4, 1, 2, 3))).permute(0, 2, 3, 4, 1)
    close(a.float(), ref, 1e-5 if dtype == torch.float32 else 1e-2, "eval bn + relu vs torch")


def test_flash_attention_large_logits(bf16_math):
    from hupr_amd import functional as F_
    k, q, v = rnd(1, 256, 64, seed=46) * 5, rnd(1, 256, 64, seed=47) * 5, rnd(1, 256, 64, seed=48)
    kq, qq = k.to(torch.bfloat16).double(), q.to(torch.bfloat16).double()
    s = torch.einsum("bjc,bkc->bjk", kq, qq)
    ref = torch.einsum("bjc,bjk->bkc", v.to(torch.bfloat16).double(), F.softmax(s, 1))
    out = F_.AttentionFn.apply(k.cuda(), q.cuda(), v.cuda(), False)
    assert torch.isfinite(out).all()
    close(out, ref, 2e-2, "flash big logits")


def test_conv_halo256_persistent_kernel_matches_128_voxel_kernel(bf16_math):
    """The 256-voxel persistent kernel only engages for >= 256 tiles; check it against the 128-voxel kernel (same
    operand rounding, same accumulation order) and against fp64 on rounded operands at such a size."""
    from hupr_amd import functional as F_
    L = F_.rt.lib()
    B, C, D, H, W = 4, 64, 8, 64, 64
    x = rnd(B, D, H, W, C, seed=50).cuda()
    w = rnd(128, C, 3, 3, 3, seed=51, scale=0.03).cuda()
    bias = rnd(128, seed=52).cuda()
    res = rnd(B, D, H, W, 128, seed=53).cuda()
    try:
        L.hupr_debug_halo_variant(1)
        y128 = F_._conv_raw(x, w, 0, bias, res, 128, (3, 3, 3), (1, 1, 1), (D, H, W))
        L.hupr_debug_halo_variant(0)
        y256 = F_._conv_raw(x, w, 0, bias, res, 128, (3, 3, 3), (1, 1, 1), (D, H, W))
    finally:
        L.hupr_debug_halo_variant(0)
    close(y256, y128, 1e-6, "halo256 vs halo128")
    ref = F.conv3d(_bf16_round(ncdhw(x.cpu()))[:1], _bf16_round(w.cpu()), bias.cpu().double(), 1, 1)
    close(ncdhw(y256.cpu())[:1] - ncdhw(res.cpu())[:1].double(), ref, 2e-5, "halo256 vs fp64")


@pytest.mark.parametrize("shape", [(8, 64, 64, 8, 32, 32, 3), (5, 128, 128, 4, 32, 32, 3), (16, 64, 64, 1, 64, 64, 1), (9, 320, 64, 1, 64, 64, 1),
                                   (32, 256, 128, 2, 16, 16, 3)])
def test_conv_residual_prefetched_under_the_last_stage_equals_the_immediate_epilogue(shape, bf16_math):
    """256-voxel 16 x 16 x 32 convolution with a residual (the second input gradient of a block's pair, the decoder's conv + residual):
    the residual elements are fetched into registers under the tile's last stage and the tile is parked like any other, instead of
    being read in an immediate epilogue — the same fp32 sum rounded once: identical bits; also with the output written over the
    residual (``out`` is ``res``), against fp64, and on the 2 x 8 x 16 tile, which keeps the immediate epilogue."""
    from hupr_amd import functional as F_
    L = F_.rt.lib()
    B, Ci, Co, D, H, W, kd = shape
    x = rnd(B, D, H, W, Ci, seed=56).cuda().bfloat16()
    w = rnd(Co, Ci, kd, 3, 3, seed=57, scale=(Ci * 9 * kd) ** -0.5).cuda()
    res = rnd(B, D, H, W, Co, seed=58).cuda().bfloat16()
    k, pad = (kd, 3, 3), (kd // 2, 1, 1)
    out = {}
    try:
        for on in (1, 0):
            L.hupr_debug_halo_res_prefetch(on)
            y = F_._conv_raw(x, w, 0, None, res, Co, k, pad, (D, H, W))
            inplace = res.clone()
            y2 = F_._conv_raw(x, w, 0, None, inplace, Co, k, pad, (D, H, W), out=inplace)
            out[on] = (y, y2)
    finally:
        L.hupr_debug_halo_res_prefetch(1)
    assert torch.equal(out[1][0], out[0][0]) and torch.equal(out[1][1], out[0][1]) and torch.equal(out[1][0], out[1][1])
    y0 = F_._conv_raw(x, w, 0, None, None, Co, k, pad, (D, H, W))
    if kd == 3:
        ref = F.conv3d(ncdhw(x.float().cpu())[:1].double(), _bf16_round(w.cpu()), None, 1, 1)
    else:
        ref = F.conv2d(ncdhw(x.float().cpu())[:1, :, 0].double(), _bf16_round(w.cpu())[:, :, 0], None, 1, 1).unsqueeze(2)
    close(ncdhw(out[1][0].float().cpu())[:1], ref + ncdhw(res.float().cpu())[:1].double(), 1e-2, "conv + residual vs fp64")
    assert not torch.equal(y0, out[1][0])


@pytest.mark.parametrize("shape", [(4, 64, 64, 8, 64, 64), (9, 128, 128, 4, 32, 32), (3, 64, 128, 8, 32, 64)])
def test_conv_halo256m_4x8x8_tile_matches_the_128_voxel_kernel(shape, bf16_math):
    """The 256-voxel kernel's 4 x 8 x 8 tile (v_mfma_f32_16x16x32_bf16, persistent workgroups, fragment pipeline across stage and item
    boundaries) against the 128-voxel kernel (hupr_debug_halo_tiles(0): 32 x 32 x 16, another fp32 order inside a 32-channel group —
    a handful of outputs one bf16 step apart) and against fp64: one and two channel chunks, one and two output-channel tiles, a tile
    count that does not divide evenly over the 256 workgroups; deterministic from launch to launch."""
    from hupr_amd import functional as F_
    L = F_.rt.lib()
    B, Ci, Co, D, H, W = shape
    x = rnd(B, D, H, W, Ci, seed=54).cuda().bfloat16()
    w = rnd(Co, Ci, 3, 3, 3, seed=55, scale=(Ci * 27) ** -0.5).cuda()
    try:
        L.hupr_debug_halo_tiles(0)
        y_128 = F_._conv_raw(x, w, 0, None, None, Co, (3, 3, 3), (1, 1, 1), (D, H, W))
        L.hupr_debug_halo_tiles(31)
        y_m16 = F_._conv_raw(x, w, 0, None, None, Co, (3, 3, 3), (1, 1, 1), (D, H, W))
        y_m16b = F_._conv_raw(x, w, 0, None, None, Co, (3, 3, 3), (1, 1, 1), (D, H, W))
    finally:
        L.hupr_debug_halo_tiles(31)
    ref = F.conv3d(ncdhw(x.float().cpu())[:1].double(), _bf16_round(w.cpu()), None, 1, 1)
    close(ncdhw(y_128.float().cpu())[:1], ref, 1e-2, "128-voxel kernel (bf16 store) vs fp64")
    assert y_m16.dtype == torch.bfloat16 and torch.equal(y_m16, y_m16b)
    d = (y_m16.float() - y_128.float()).abs()
    assert (d > 0).float().mean().item() < 2e-3 and (d <= torch.maximum(y_128.float().abs(), y_m16.float().abs()) * 2 ** -7 + 1e-5).all()
    close(ncdhw(y_m16.float().cpu())[:1], ref, 1e-2, "halo256m (bf16 store) vs fp64")


@pytest.mark.parametrize("shape", [(4, 64, 8, 64, 64), (5, 128, 8, 32, 64), (7, 64, 8, 64, 40)])
def test_conv_halo256m_32_output_channels_match_the_128_voxel_kernel(shape, bf16_math):
    """32 output channels (the input gradient of the encoders' first convolution, `layers.py:194` Conv3d(32 -> 64) seen from its output):
    the 16 x 16 x 32 kernel's 8 x 8 x 8 tile — eight waves = eight depth slices, every wave with all the channels, the halo's padding
    planes zeroed once — against the 128-voxel kernel these launches ran on before (hupr_debug_halo_tiles(7)) and against fp64; one and two
    channel chunks, with and without the residual epilogue, an uneven tile count over the 256 workgroups; deterministic."""
    from hupr_amd import functional as F_
    L = F_.rt.lib()
    B, Ci, D, H, W = shape
    Co = 32
    x = rnd(B, D, H, W, Ci, seed=154).cuda().bfloat16()
    w = rnd(Co, Ci, 3, 3, 3, seed=155, scale=(Ci * 27) ** -0.5).cuda()
    res = rnd(B, D, H, W, Co, seed=156).cuda().bfloat16()
    out = {}
    try:
        for mode in (7, 31):
            L.hupr_debug_halo_tiles(mode)
            out[mode] = (F_._conv_raw(x, w, 0, None, None, Co, (3, 3, 3), (1, 1, 1), (D, H, W)),
                         F_._conv_raw(x, w, 0, None, res, Co, (3, 3, 3), (1, 1, 1), (D, H, W)))
        again = F_._conv_raw(x, w, 0, None, None, Co, (3, 3, 3), (1, 1, 1), (D, H, W))
    finally:
        L.hupr_debug_halo_tiles(31)
    assert torch.equal(again, out[31][0])
    for a, b in zip(out[31], out[7]):
        d = (a.float() - b.float()).abs()
        assert (d > 0).float().mean().item() < 2e-3 and (d <= torch.maximum(a.float().abs(), b.float().abs()) * 2 ** -7 + 1e-5).all()
    assert not torch.equal(out[31][0], out[31][1])
    ref = F.conv3d(ncdhw(x.float().cpu())[:2].double(), _bf16_round(w.cpu()), None, 1, 1)
    close(ncdhw(out[31][0].float().cpu())[:2], ref, 1e-2, "halo256m 8x8x8, Co = 32 (bf16 store) vs fp64")
    close(ncdhw(out[31][1].float().cpu())[:2], ref + ncdhw(res.float().cpu())[:2].double(), 1e-2, "halo256m 8x8x8, Co = 32 + residual vs fp64")


@pytest.mark.parametrize("shape", [(32, 64, 256, 2, 16, 16), (17, 128, 128, 2, 16, 32), (8, 320, 64, 1, 64, 64), (9, 64, 192, 1, 32, 48)])
def test_conv_halo256m_two_slice_tile_matches_the_128_voxel_kernel(shape, bf16_math):
    """Depth 2 (encoder level 3) and depth 1 with 1 x 3 x 3 taps (decoder): the 16 x 16 x 32 kernel's 2 x 8 x 16 / 1 x 16 x 16 tiles against the 128-voxel kernel these layers ran on before
    (hupr_debug_halo_tiles(1)) — same products, another fp32 order: a few outputs one bf16 step apart — and against fp64; with and
    without the residual epilogue; even and uneven tile counts over the 256 workgroups."""
    from hupr_amd import functional as F_
    L = F_.rt.lib()
    B, Ci, Co, D, H, W = shape
    k3, pad = ((3, 3, 3), (1, 1, 1)) if D > 1 else ((1, 3, 3), (0, 1, 1))      # D = 1: the decoder's 1 x 3 x 3 taps on the 1 x 16 x 16 tile
    x = rnd(B, D, H, W, Ci, seed=56).cuda().bfloat16()
    w = rnd(Co, Ci, *k3, seed=57, scale=(Ci * 9 * k3[0]) ** -0.5).cuda()
    res = rnd(B, D, H, W, Co, seed=58).cuda().bfloat16()
    out = {}
    try:
        for mode in (1, 31):                # 7: all tiles, 1: the 4 x 8 x 8 tile only
            L.hupr_debug_halo_tiles(mode)
            out[mode] = (F_._conv_raw(x, w, 0, None, None, Co, k3, pad, (D, H, W)),
                         F_._conv_raw(x, w, 0, None, res, Co, k3, pad, (D, H, W)))
    finally:
        L.hupr_debug_halo_tiles(31)
    for a, b in zip(out[31], out[1]):
        d = (a.float() - b.float()).abs()
        assert (d > 0).float().mean().item() < 2e-3 and (d <= torch.maximum(a.float().abs(), b.float().abs()) * 2 ** -7 + 1e-5).all()
    ref = F.conv3d(ncdhw(x.float().cpu())[:1].double(), _bf16_round(w.cpu()), None, 1, pad)
    close(ncdhw(out[31][0].float().cpu())[:1], ref, 1e-2, "halo256m 2x8x16 / 1x16x16 (bf16 store) vs fp64")


# ---- bf16-stored activations ("bf16act" kernels of the encoder island) ------------------------------------------------
# The fp32-activation kernels round x to bf16 while staging, so on bf16-representable inputs both variants perform the
# SAME arithmetic; the bf16act result must equal the fp32 result rounded once to bf16 (store rounding only).
def _q(t):
    return t.to(torch.bfloat16).to(torch.float32)


HALO_ACT_CASES = [
    # B, Ci, Co, D, H, W, kd, bias, res
    (2, 32, 64, 4, 16, 16, 3, True, False),      # stem (KC = 32 kernel, bias)
    (4, 64, 128, 8, 64, 64, 3, False, True),     # engages the 256-voxel persistent kernel
    (2, 128, 256, 2, 16, 16, 3, False, False),
    (3, 320, 64, 1, 16, 16, 1, False, True),     # 2-D tile
]


@pytest.mark.parametrize("case", HALO_ACT_CASES)
def test_conv_halo_bf16_activations(case, bf16_math):
    from hupr_amd import functional as F_
    B, Ci, Co, D, H, W, kd, has_bias, has_res = case
    k, pad = (kd, 3, 3), (kd // 2, 1, 1)
    x = _q(rnd(B, D, H, W, Ci, seed=60)).cuda()
    w = rnd(Co, Ci, *k, seed=61, scale=(Ci * 9 * kd) ** -0.5).cuda()
    bias = rnd(Co, seed=62).cuda() if has_bias else None
    res = _q(rnd(B, D, H, W, Co, seed=63)).cuda() if has_res else None
    y32 = F_._conv_raw(x, w, 0, bias, res, Co, k, pad, (D, H, W))
    y16 = F_._conv_raw(x.bfloat16(), w, 0, bias, res.bfloat16() if has_res else None, Co, k, pad, (D, H, W))
    assert y16.dtype == torch.bfloat16
    d16 = (y16.float() - _q(y32)).abs()
    if not torch.equal(y16.float(), _q(y32)):
        # only where the 256-voxel kernel engages (bf16-stored activations only): it multiplies on v_mfma_f32_16x16x32_bf16 (another fp32
        # order inside a 32-channel group than the 128-voxel kernel's 32 x 32 x 16): a few outputs land one bf16 step away
        assert (d16 > 0).float().mean().item() < 2e-3 and (d16 <= torch.maximum(_q(y32).abs(), y16.float().abs()) * 2 ** -7 + 1e-5).all(), d16.max().item()      # (+ the fp32 order noise where residual and sum cancel)
        L_ = F_.rt.lib()
        try:
            L_.hupr_debug_halo_tiles(0)
            y16b = F_._conv_raw(x.bfloat16(), w, 0, bias, res.bfloat16() if has_res else None, Co, k, pad, (D, H, W))
        finally:
            L_.hupr_debug_halo_tiles(31)
        assert torch.equal(y16b.float(), _q(y32))              # the 32 x 32 x 16 form: store rounding only
    # weight gradient: fp32 output, identical products; only the fp32 summation order over voxel slices differs
    # (LDS-DMA kernel: two K halves per workgroup, other slice count)
    if Ci % 32 == 0 and Co % 8 == 0:
        dy = _q(rnd(B, D, H, W, Co, seed=64)).cuda()
        L = F_.rt.lib()
        ws = F_.workspace(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd), x.device)
        dw32, dw16 = torch.empty_like(w), torch.empty_like(w)
        F_.rt.check(L.hupr_conv3x3_wgrad_halo_bf16(F_.rt.ptr(x), F_.rt.ptr(dy), F_.rt.ptr(dw32), B, D, H, W, Ci, Ci, Co, Co,
                                                   kd, F_.rt.ptr(ws), ws.numel(), F_.rt.stream()))
        xb, dyb = x.bfloat16(), dy.bfloat16()
        F_.rt.check(L.hupr_conv3x3_wgrad_halo_bf16act(F_.rt.ptr(xb), F_.rt.ptr(dyb), F_.rt.ptr(dw16), B, D, H, W, Ci, Ci, Co,
                                                      Co, kd, F_.rt.ptr(ws), ws.numel(), F_.rt.stream()))
        close(dw16, dw32, 2e-6, "bf16act wgrad vs fp32-stored wgrad")


@pytest.mark.parametrize("shape", [(3, 32, 64, 2, 16, 16, 3), (2, 24, 40, 1, 16, 32, 1), (2, 32, 72, 4, 8, 24, 3)])
def test_wgrad_k_quarter_mode_for_narrow_inputs(shape, bf16_math):
    """Ci <= 32: the LDS-DMA weight-gradient kernels split K four ways instead of leaving the waves of the missing ci blocks idle
    (3-D taps: hupr_k_wgrad_halo_m16<true, true> since round 6 — a wave = one K-step x one 16-wide ci block).
    Both modes against fp64 autograd on the same bf16 operands, and against each other (summation order only)."""
    from hupr_amd import functional as F_
    B, Ci, Co, D, H, W, kd = shape
    L = F_.rt.lib()
    x = _q(rnd(B, D, H, W, Ci, seed=70)).cuda().bfloat16()
    dy = _q(rnd(B, D, H, W, Co, seed=71)).cuda().bfloat16()
    w = torch.zeros(Co, Ci, kd, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(x.double().permute(0, 4, 1, 2, 3), w, padding=(kd // 2, 1, 1)).backward(dy.double().permute(0, 4, 1, 2, 3))
    ws = F_.workspace(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd), x.device)
    got = {}
    try:
        for mode in (0, 2, 3):                 # 2: K quarters on the 16 x 16 x 32 kernel (3-D; round 6), 3: on the 32 x 32 x 16 kernel
            L.hupr_debug_wgrad_ci32(mode)
            dw = torch.full((Co, Ci, kd, 3, 3), float("nan"), device="cuda")
            F_.rt.check(L.hupr_conv3x3_wgrad_halo_bf16act(F_.rt.ptr(x), F_.rt.ptr(dy), F_.rt.ptr(dw), B, D, H, W, Ci, Ci, Co, Co, kd,
                                                          F_.rt.ptr(ws), ws.numel(), F_.rt.stream()))
            got[mode] = dw
            close(dw, w.grad.float(), 2e-6, "wgrad mode %d vs fp64" % mode)
    finally:
        L.hupr_debug_wgrad_ci32(1)
    close(got[2], got[0], 2e-6, "K quarters vs K halves")
    close(got[3], got[0], 2e-6, "K quarters (32 x 32 x 16 kernel) vs K halves")


def test_conv_autograd_bf16_activations(bf16_math):
    """ConvFn on bf16 tensors (fwd, dgrad, wgrad, bias grad) against fp64 on the same rounded operands."""
    from hupr_amd import functional as F_
    B, Ci, Co, D, H, W = 2, 64, 64, 4, 16, 16
    x = _q(rnd(B, Ci, D, H, W, seed=70))
    w = rnd(Co, Ci, 3, 3, 3, seed=71, scale=(Ci * 27) ** -0.5)
    b = rnd(Co, seed=72)
    gy = _q(rnd(B, Co, D, H, W, seed=73))
    wq = _bf16_round(w)
    xr, wr, br = x.double().requires_grad_(True), wq.clone().requires_grad_(True), b.double().requires_grad_(True)
    F.conv3d(xr, wr, br, 1, 1).backward(gy.double())
    xg = cl(x).cuda().bfloat16().requires_grad_(True)
    wg, bg = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = F_.conv(xg, wg, bg, None, (1, 1, 1))
    assert y.dtype == torch.bfloat16
    close(ncdhw(y.float()), F.conv3d(x.double(), wq, b.double(), 1, 1), 6e-3, "bf16act conv fwd (one bf16 store rounding)")
    y.backward(cl(gy).cuda().bfloat16())
    assert xg.grad.dtype == torch.bfloat16
    close(ncdhw(xg.grad.float()), xr.grad, 6e-3, "bf16act dgrad")
    close(wg.grad, wr.grad, 5e-5, "bf16act wgrad")
    close(bg.grad, br.grad, 1e-5, "bf16act bias grad")


@pytest.mark.parametrize("B,G,H,W,C", [(2, 8, 16, 16, 64), (3, 4, 8, 8, 128), (2, 2, 16, 16, 256)])
def test_temporal_merge_bf16_activations(B, G, H, W, C, bf16_math):
    """F_.temporal_merge on a bf16-stored map (mixed-storage GEMM kernels: bf16 source, fp32 merged map, bf16 input
    gradient from G*B batched GEMMs) against (a) fp64 on the same rounded operands and (b) the generic ConvFn on the
    fp32 copy of the same map — same products, same accumulation order, so (b) is an exact comparison."""
    from hupr_amd import functional as F_
    x = _q(rnd(B, C, G, H, W, seed=80))
    w = rnd(C, C, G, 1, 1, seed=81, scale=(C * G) ** -0.5)
    gy = rnd(B, C, 1, H, W, seed=82)
    wq = _bf16_round(w)
    xr, wr = x.double().requires_grad_(True), wq.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr)
    yr.backward(_bf16_round(gy))
    xg = cl(x).cuda().bfloat16().requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    y = F_.temporal_merge(xg, wg)
    assert y.dtype == torch.float32 and tuple(y.shape) == (B, 1, H, W, C)
    close(ncdhw(y), yr, 2e-5, "temporal merge fwd")
    y.backward(cl(gy).cuda())
    assert xg.grad.dtype == torch.bfloat16
    close(ncdhw(xg.grad.float()), xr.grad, 6e-3, "temporal merge dgrad (one bf16 store rounding)")
    close(wg.grad, wr.grad, 5e-5, "temporal merge wgrad")
    # generic path on the fp32 copy
    x32 = cl(x).cuda().requires_grad_(True)
    w32 = w.cuda().requires_grad_(True)
    y32 = F_.conv(x32, w32, None, None, (0, 0, 0))
    y32.backward(cl(gy).cuda())
    assert torch.equal(y, y32)
    close(xg.grad.float(), x32.grad, 4e-3, "dgrad vs generic")
    close(wg.grad, w32.grad, 2e-6, "wgrad vs generic")


@pytest.mark.parametrize("shape", [(32, 64, 64, 8, 64, 64, 3), (3, 64, 128, 4, 16, 24, 3), (5, 128, 256, 2, 16, 16, 3), (2, 320, 64, 1, 32, 32, 1),
                                   (7, 128, 64, 1, 16, 48, 1), (1, 64, 64, 2, 8, 8, 3), (4, 96, 72, 4, 8, 16, 3)])
def test_wgrad_halo_on_16x16x32_matches_the_32x32x16_kernel_and_fp64(shape, bf16_math):
    """The LDS-DMA weight gradient on v_mfma_f32_16x16x32_bf16 (round 5: new lane ownership, two-bit row swizzle, wave tile 64 co x 16
    ci) against the rounds-2-4 kernel on the same bf16 tensors — the same bf16 products, another fp32 summation order — and against
    fp64; level-1 shape at the bench batch, several co / ci tiles, a single tile per workgroup, ragged channel counts (96 -> 72), 2-D."""
    from hupr_amd import functional as F_
    L, rt = F_.rt.lib(), F_.rt
    B, Ci, Co, D, H, W, kd = shape
    x = rnd(B, D, H, W, Ci, seed=500).cuda().bfloat16()
    dy = rnd(B, D, H, W, Co, seed=501).cuda().bfloat16()
    ws = torch.empty(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd), dtype=torch.uint8, device="cuda")
    out = {}
    try:
        for m16 in (1, 0):
            L.hupr_debug_wgrad_m16(m16)
            dw = torch.full((Co, Ci, kd, 3, 3), float("nan"), device="cuda")
            rt.check(L.hupr_conv3x3_wgrad_halo_bf16act(rt.ptr(x), rt.ptr(dy), rt.ptr(dw), B, D, H, W, Ci, Ci, Co, Co, kd, rt.ptr(ws), ws.numel(),
                                                       rt.stream()))
            out[m16] = dw
    finally:
        L.hupr_debug_wgrad_m16(1)
    assert torch.isfinite(out[1]).all()
    close(out[1], out[0], 2e-5, "16x16x32 vs 32x32x16 weight gradient")
    if B * D * H * W <= 40000:                                   # fp64 reference on the host for the smaller cases
        xr = x.double().cpu().permute(0, 4, 1, 2, 3)
        wr = torch.zeros(Co, Ci, kd, 3, 3, dtype=torch.float64, requires_grad=True)
        yr = F.conv3d(xr, wr, None, 1, (kd // 2, 1, 1))
        yr.backward(dy.double().cpu().permute(0, 4, 1, 2, 3))
        close(out[1], wr.grad, 1e-5, "16x16x32 weight gradient vs fp64")


def test_pack_table_covers_recent_readers_and_brings_idle_models_back_in_one_launch(bf16_math):
    """The table pass after an optimiser step refreshes the weights READ during the current or previous step — not every weight the
    process holds (a second, idle model used to be repacked after every step of the first) — and a model that sat idle comes back with
    ONE table launch at its first read, not one per weight; contents always equal the one-weight pack kernels."""
    from hupr_amd import functional as F_
    L = F_.rt.lib()
    mk = lambda seed: [torch.nn.Parameter(rnd(64, 64, 3, 3, 3, seed=seed + i).cuda()) for i in range(6)]
    wa, wb = mk(700), mk(720)
    for w in wa + wb:
        F_._packed(w, 0, 1)
    for step in range(4):                                     # model A trains, model B idles
        with torch.no_grad():
            for w in wa:
                w.add_(0.01)
        F_.invalidate_packed()
        n0 = L.hupr_launch_count()
        F_.refresh_packed(wa[0].device)
        for w in wa:
            assert torch.equal(F_._packed(w, 0, 1), F_.pack_weights_bf16(w, 0))
        n = L.hupr_launch_count() - n0 - len(wa)              # (minus the reference packs)
        assert n == 1, n                                      # one table launch per step
        if step >= 2:                                         # B dropped out of the table: not repacked, its entries stale
            assert all(F_._pack_entries[(w.data_ptr(), 1)].stamp[0] != F_.PACK_EPOCH for w in wb)
    n0 = L.hupr_launch_count()
    got = [F_._packed(w, 1, 1) for w in wb]                   # B comes back
    assert L.hupr_launch_count() - n0 == 1
    for w, g in zip(wb, got):
        assert torch.equal(g, F_.pack_weights_bf16(w, 1))


@pytest.mark.parametrize("shape", [(8, 64, 64, 8, 32, 32, 3), (16, 128, 128, 4, 32, 32, 3), (4, 256, 256, 2, 16, 16, 3), (8, 320, 64, 1, 64, 64, 1),
                                   (2, 64, 128, 4, 16, 16, 3), (3, 96, 64, 2, 8, 8, 3), (8, 32, 64, 8, 64, 64, 3), (2, 32, 64, 4, 16, 16, 3)])
def test_two_weight_gradients_of_one_input_in_one_launch(shape, bf16_math):
    """hupr_conv3x3_wgrad_halo_bf16act_dual: the weight gradients of the two convolutions of a residual block (same x, two dy) as one
    launch over 2 Co output channels + one reduction — the same partial tensors and the same sums per element as two calls: identical
    bits, two launches instead of four (level-1 shape with the XCD-aware grid, levels 2 / 3, a decoder block with five ci tiles,
    small and ragged cases, the first block's 32 input channels on the K-quarter kernel)."""
    from hupr_amd import functional as F_
    L, rt = F_.rt.lib(), F_.rt
    B, Ci, Co, D, H, W, kd = shape
    assert L.hupr_conv3x3_wgrad_halo_dual_supported(B, D, H, W, Ci, Co, kd)
    x = rnd(B, D, H, W, Ci, seed=520).cuda().bfloat16()
    dya, dyb = (rnd(B, D, H, W, Co, seed=521 + i).cuda().bfloat16() for i in range(2))
    ws = torch.empty(2 * L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd), dtype=torch.uint8, device="cuda")
    single = []
    n0 = L.hupr_launch_count()
    for dy in (dya, dyb):
        dw = torch.full((Co, Ci, kd, 3, 3), float("nan"), device="cuda")
        rt.check(L.hupr_conv3x3_wgrad_halo_bf16act(rt.ptr(x), rt.ptr(dy), rt.ptr(dw), B, D, H, W, Ci, Ci, Co, Co, kd, rt.ptr(ws), ws.numel(),
                                                   rt.stream()))
        single.append(dw)
    n1 = L.hupr_launch_count()
    da, db = (torch.full((Co, Ci, kd, 3, 3), float("nan"), device="cuda") for _ in range(2))
    rt.check(L.hupr_conv3x3_wgrad_halo_bf16act_dual(rt.ptr(x), rt.ptr(dya), rt.ptr(dyb), rt.ptr(da), rt.ptr(db), B, D, H, W, Ci, Ci, Co, Co,
                                                    kd, rt.ptr(ws), ws.numel(), rt.stream()))
    n2 = L.hupr_launch_count()
    assert torch.equal(da, single[0]) and torch.equal(db, single[1])
    assert (n1 - n0, n2 - n1) == (4, 2)
    assert not L.hupr_conv3x3_wgrad_halo_dual_supported(B, D, H, W, Ci, 72, kd)          # Co % 64 != 0: two calls


@pytest.mark.parametrize("shape", [(8, 64, 64, 8, 32, 32, 3), (4, 128, 256, 2, 16, 16, 3), (8, 64, 64, 1, 32, 32, 1), (2, 96, 72, 2, 16, 16, 3)])
def test_splitk_reduction_slices_agree(shape, bf16_math):
    """The split-K reduction behind every weight gradient (S slices of the partial tensors per workgroup, eight loads of a thread in
    flight): 4 and 16 slices add the same partial tensors in another order; both against fp64, and the same bits on every run.  Round 6:
    the convolution gradients' parameter-layout write goes through LDS (hupr_k_splitk_reduce_t: one contiguous run per output channel and
    channel block instead of 4-byte stores 108 bytes apart) — the same sums bit for bit as the scattered-store kernel (+ 256), for one
    gradient and for the two of a dual launch, written into views that are only 4-byte aligned (gradient-bucket slots)."""
    from hupr_amd import functional as F_
    L, rt = F_.rt.lib(), F_.rt
    B, Ci, Co, D, H, W, kd = shape
    x = rnd(B, D, H, W, Ci, seed=510).cuda().bfloat16()
    dy = rnd(B, D, H, W, Co, seed=511).cuda().bfloat16()
    ws = torch.empty(L.hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd), dtype=torch.uint8, device="cuda")
    out = {}
    try:
        for sl in (4, 16, 0, 16):
            L.hupr_debug_splitk_slices(sl)
            dw = torch.full((Co, Ci, kd, 3, 3), float("nan"), device="cuda")
            rt.check(L.hupr_conv3x3_wgrad_halo_bf16act(rt.ptr(x), rt.ptr(dy), rt.ptr(dw), B, D, H, W, Ci, Ci, Co, Co, kd, rt.ptr(ws), ws.numel(),
                                                       rt.stream()))
            if sl in out:
                assert torch.equal(out[sl], dw)
            out[sl] = dw
    finally:
        L.hupr_debug_splitk_slices(0)
    close(out[16], out[4], 2e-6, "16 vs 4 slices")
    assert torch.equal(out[0], out[4]) or torch.equal(out[0], out[16])
    n0 = L.hupr_launch_count()
    try:
        for sl in (4, 16):
            L.hupr_debug_splitk_slices(sl + 256)              # the scattered-store kernel of rounds 1-5
            flat = torch.full((Co * Ci * kd * 9 + 3,), float("nan"), device="cuda")
            dw = flat[3:].view(Co, Ci, kd, 3, 3)              # a 4-byte aligned destination (offset 12 bytes)
            rt.check(L.hupr_conv3x3_wgrad_halo_bf16act(rt.ptr(x), rt.ptr(dy), rt.ptr(dw), B, D, H, W, Ci, Ci, Co, Co, kd, rt.ptr(ws), ws.numel(),
                                                       rt.stream()))
            assert torch.equal(dw, out[sl]), sl
            assert torch.isnan(flat[:3]).all()
            if Co % 64 == 0 and Ci % 8 == 0:                  # the two gradients of a residual block in one reduction (rows split between dw / dw2)
                ws2 = torch.empty(2 * ws.numel(), dtype=torch.uint8, device="cuda")
                dy2 = rnd(B, D, H, W, Co, seed=512).cuda().bfloat16()
                pair = {}
                for legacy in (256, 0):
                    L.hupr_debug_splitk_slices(sl + legacy)
                    da, db = (torch.full((Co, Ci, kd, 3, 3), float("nan"), device="cuda") for _ in range(2))
                    rt.check(L.hupr_conv3x3_wgrad_halo_bf16act_dual(rt.ptr(x), rt.ptr(dy), rt.ptr(dy2), rt.ptr(da), rt.ptr(db), B, D, H, W, Ci, Ci,
                                                                    Co, Co, kd, rt.ptr(ws2), ws2.numel(), rt.stream()))
                    pair[legacy] = (da, db)
                assert torch.equal(pair[0][0], pair[256][0]) and torch.equal(pair[0][1], pair[256][1]) and torch.equal(pair[0][0], out[sl])
    finally:
        L.hupr_debug_splitk_slices(0)
    assert L.hupr_launch_count() > n0
    xr = x.double().cpu().permute(0, 4, 1, 2, 3)
    wr = torch.zeros(Co, Ci, kd, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(xr, wr, None, 1, (kd // 2, 1, 1)).backward(dy.double().cpu().permute(0, 4, 1, 2, 3))
    close(out[16], wr.grad, 1e-5, "weight gradient (16 slices) vs fp64")
    close(out[4], wr.grad, 1e-5, "weight gradient (4 slices) vs fp64")


def test_pack_cache_table_refresh_matches_single_packs(bf16_math):
    """The packed-weight cache refreshes every registered weight with ONE table-driven launch (32x32xtaps LDS tiles for
    bf16 halo weights, element-wise blocks for the rest); both layouts of every entry must equal the one-weight pack
    kernels bit for bit, before and after the parameters change behind torch's version counter."""
    from hupr_amd import functional as F_
    shapes = [((64, 64, 3, 3, 3), 1), ((128, 64, 3, 3), 1), ((64, 32, 3, 3, 3), 1), ((32, 96, 3, 3), 1), ((16, 64, 3, 3), 1),
              ((64, 64, 8, 1, 1), 0), ((256, 1024, 3, 3), 1)]
    ws = [(torch.nn.Parameter(rnd(*sh, seed=130 + i).cuda()), kind) for i, (sh, kind) in enumerate(shapes)]
    single = lambda w, mode, kind: F_.pack_weights_bf16(w, mode) if kind else F_.pack_weights(w, mode)
    for rnd_ in range(2):
        for w, kind in ws:
            for mode in (0, 1):
                got = F_._packed(w, mode, kind)
                assert torch.equal(got, single(w, mode, kind)), (tuple(w.shape), mode, kind, rnd_)
        with torch.no_grad():
            for w, _ in ws:
                w.mul_(1.5).add_(0.25)
        F_.invalidate_packed()


@pytest.mark.parametrize("C,H", [(64, 16), (128, 16), (256, 16)])
def test_mscsa_level_bf16_concatenated_output(C, H, bf16_math):
    """cat_bf16 form of MSCSALevelFn: the attention kernels also write the bf16 concatenation of the four maps, and the
    backward reads a bf16 gradient that is a strided column slice of a wider tensor (as torch.cat's backward hands it
    over).  Must equal the four-output form fed with the same (bf16-representable) gradients, bit for bit."""
    from hupr_amd import functional as F_
    B, N = 2, H * H
    ra, re = rnd(B, 1, H, H, C, seed=140).cuda(), rnd(B, 1, H, H, C, seed=141).cuda()
    ws = [rnd(C, C, 1, 1, seed=142 + i, scale=C ** -0.5).cuda() for i in range(8)]
    wide = rnd(B, 1, H, H, 4 * C + 64, seed=150).cuda().bfloat16()          # gradient of a wider concatenation
    gcat = wide[..., 64:]                                                     # strided view, as CatBackward produces
    assert not gcat.is_contiguous()

    def run(cat):
        a, e = ra.clone().requires_grad_(True), re.clone().requires_grad_(True)
        w = [t.clone().requires_grad_(True) for t in ws]
        outs = F_.MSCSALevelFn.apply(a, e, cat, *w)
        if cat:
            assert len(outs) == 1 and outs[0].dtype == torch.bfloat16
            y = outs[0]
            y.backward(gcat)
        else:
            y = torch.cat([o.bfloat16() for o in outs], 4)
            torch.autograd.backward(list(outs), [gcat[..., i * C:(i + 1) * C].float().contiguous() for i in range(4)])
        return y.detach(), [a.grad, e.grad] + [t.grad for t in w]

    y1, g1 = run(True)
    y0, g0 = run(False)
    assert torch.equal(y1, y0)
    for i, (x, y) in enumerate(zip(g1, g0)):
        assert torch.equal(x, y), "gradient %d differs" % i


@pytest.mark.parametrize("C,H,B", [(128, 32, 8), (256, 16, 8), (256, 16, 3), (64, 16, 8)])
def test_mscsa_level_attentions_as_one_launch_per_kernel(C, H, B, bf16_math):
    """Levels 2 and 3 of a training batch: the four attentions of the level in ONE forward launch, one row-sum launch, one dQ launch and
    two dK / dV launches (blockIdx.z / the sample index pick the item) — the same workgroups doing the same arithmetic as twelve / sixteen
    separate launches: identical bits, fewer launches.  (C = 64: the level-1 forward and dK / dV kernels stay one launch per attention.)"""
    from hupr_amd import functional as F_, runtime as rt
    ra, re = rnd(B, 1, H, H, C, seed=340).cuda(), rnd(B, 1, H, H, C, seed=341).cuda()
    ws = [rnd(C, C, 1, 1, seed=342 + i, scale=C ** -0.5).cuda() for i in range(8)]
    gcat = rnd(B, 1, H, H, 4 * C, seed=350).cuda().bfloat16()

    def run(batch):
        old = F_.ATTN_LEVEL_BATCH
        F_.ATTN_LEVEL_BATCH = batch
        try:
            a, e = ra.clone().requires_grad_(True), re.clone().requires_grad_(True)
            w = [t.clone().requires_grad_(True) for t in ws]
            n0 = rt.lib().hupr_launch_count()
            (y,) = F_.MSCSALevelFn.apply(a, e, True, *w)
            y.backward(gcat)
            n = rt.lib().hupr_launch_count() - n0
            torch.cuda.synchronize()
            return n, [y.detach(), a.grad, e.grad] + [t.grad for t in w]
        finally:
            F_.ATTN_LEVEL_BATCH = old

    n1, g1 = run(True)
    n0, g0 = run(False)
    for i, (x, y) in enumerate(zip(g1, g0)):
        assert torch.equal(x, y), "tensor %d differs" % i
    assert n0 - n1 == (6 if C == 64 else 3 + 8), (n0, n1)          # forward 4 -> 1, backward 12 -> 4; level-1 shape: backward 12 -> 6


@pytest.mark.parametrize("Ci,Co,shape", [(64, 64, (8, 8, 32, 32)), (64, 64, (6, 8, 64, 64)), (32, 64, (4, 8, 64, 64)),
                                         # round 5: several output tiles / the 2 x 8 x 16 tile (register-resident sums).  Level 2 at the bench
                                         # batch (4 tiles per workgroup, alternating between two output tiles), with one tile per workgroup,
                                         # with a ragged tile count; level 3 (Co = 256, one tile per workgroup), its first block (128 -> 256)
                                         (128, 128, (32, 4, 32, 32)), (64, 128, (8, 4, 32, 32)), (128, 128, (12, 4, 32, 32)),
                                         (256, 256, (32, 2, 16, 16)), (128, 256, (32, 2, 16, 16))])
def test_conv_fused_batchnorm_statistics(Ci, Co, shape, bf16_math):
    """The 256-voxel convolution kernel leaves the column sums of its (bf16-rounded) output for the BatchNorm that follows
    (functional.conv(..., stats=True) -> BNActFn): mean / variance / running statistics and the normalised output must
    match the separate statistics pass over the same tensor."""
    import torch.nn as nn
    from hupr_amd import functional as F_
    B, D, H, W = shape
    x = cl(rnd(B, Ci, D, H, W, seed=160)).cuda().bfloat16()
    w = (rnd(Co, Ci, 3, 3, 3, seed=161, scale=(Ci * 27) ** -0.5)).cuda()
    if Ci % 64:                              # the 256-voxel kernel needs Ci % 64 == 0: no fused statistics, separate pass
        assert not F_.rt.lib().hupr_conv3x3_halo_stats_supported(B, D, H, W, Ci, Co, 3)
        return
    assert F_.rt.lib().hupr_conv3x3_halo_stats_supported(B, D, H, W, Ci, Co, 3)
    # at most two distinct output tiles per workgroup: 256 output channels only with one tile per workgroup
    assert not F_.rt.lib().hupr_conv3x3_halo_stats_supported(64, 4, 32, 32, Ci, 256, 3)
    assert not F_.rt.lib().hupr_conv3x3_halo_stats_supported(B, D, H, W, Ci, 192, 3)
    res = []
    saved = F_.CONV_STATS
    F_.CONV_STATS = True                     # (the library default since round 3)
    for fused in (True, False):
        bn = nn.BatchNorm3d(Co).cuda()
        with torch.no_grad():
            bn.weight.copy_(rnd(Co, seed=162).cuda() * 0.2 + 1.0)
            bn.bias.copy_(rnd(Co, seed=163).cuda() * 0.1)
        F_._conv_stats.clear()
        y = F_.conv(x, w, None, None, (1, 1, 1), stats=fused)
        assert (y.data_ptr() in F_._conv_stats) == fused
        out = F_.BNActFn.apply(y, bn.weight, bn.bias, bn, True, True)
        assert not F_._conv_stats
        res.append((y, out, bn.running_mean.clone(), bn.running_var.clone()))
    F_.CONV_STATS = saved
    (y1, o1, m1, v1), (y0, o0, m0, v0) = res
    assert torch.equal(y1, y0)
    close(m1, m0, 2e-6, "running mean")
    close(v1, v0, 2e-5, "running var")
    close(o1.float(), o0.float(), 8e-3, "normalised output (bf16 storage)")
    assert (o1 != o0).float().mean().item() < 2e-3           # only last-bit scale / shift differences


@pytest.mark.parametrize("act", ["bf16", "f32"])
def test_dual_conv_matches_two_convs(act, bf16_math):
    """DualConvFn (input gradients of the two convolutions summed in the second kernel's residual epilogue, in place)
    against two ConvFn nodes whose input gradients autograd adds."""
    from hupr_amd import functional as F_
    B, Ci, Co, D, H, W = 2, 64, 128, 4, 16, 16
    dt = torch.bfloat16 if act == "bf16" else torch.float32
    x0 = cl(rnd(B, Ci, D, H, W, seed=120)).cuda().to(dt)
    wa0, wb0 = (rnd(Co, Ci, 3, 3, 3, seed=121 + i, scale=(Ci * 27) ** -0.5).cuda() for i in range(2))
    ga, gb = (cl(rnd(B, Co, D, H, W, seed=123 + i)).cuda().to(dt) for i in range(2))

    def run(fused):
        x = x0.clone().requires_grad_(True)
        wa, wb = wa0.clone().requires_grad_(True), wb0.clone().requires_grad_(True)
        if fused:
            ya, yb = F_.DualConvFn.apply(x, wa, wb, (1, 1, 1))
        else:
            ya, yb = F_.conv(x, wa, None, None, (1, 1, 1)), F_.conv(x, wb, None, None, (1, 1, 1))
        torch.autograd.backward([ya, yb], [ga, gb])
        return ya.detach(), yb.detach(), x.grad, wa.grad, wb.grad

    f, u = run(True), run(False)
    assert torch.equal(f[0], u[0]) and torch.equal(f[1], u[1])
    close(f[2].float(), u[2].float(), 1e-2 if act == "bf16" else 1e-6, "summed input gradient")
    assert torch.equal(f[3], u[3]) and torch.equal(f[4], u[4])


def _qs_case(B, N, C, seed, kscale=None, profile=None):
    """bf16 operands of one attention with the query pre-scaled by log2(e) -> (device tensors, fp64 reference pieces).  The reference
    uses exactly what the kernels see: bf16 K, V, dO and the bf16 Q' divided by log2(e) in fp64."""
    import math
    kscale = C ** -0.25 if kscale is None else kscale
    k, q, v, g = (rnd(B, N, C, seed=seed + i, scale=(kscale if i < 2 else 1.0)) for i in range(4))
    if profile is not None:
        k = k * profile[None, :, None]                      # key norms that vary along the key axis: the running maximum keeps moving
    kb, vb, gb = k.bfloat16(), v.bfloat16(), g.bfloat16()
    qsb = (q * math.log2(math.e)).bfloat16()                # Q' as the projection GEMM's epilogue rounds it
    kr, vr = kb.double().requires_grad_(True), vb.double().requires_grad_(True)
    qr = (qsb.double() / math.log2(math.e)).requires_grad_(True)
    return (kb, qsb, vb, gb, v, g), (kr, qr, vr)


@pytest.mark.parametrize("B,N,C,residual", [(2, 4096, 64, True), (8, 512, 64, False), (2, 384, 64, True), (2, 256, 128, True),
                                            (3, 1024, 128, False), (2, 256, 256, True), (1, 384, 256, False)])
def test_flash_attention_qs_kernels_vs_fp64(B, N, C, residual, bf16_math):
    """The QS entry points (query operand = log2(e) Q as bf16; accumulator input = minus the deferred running maximum / minus the
    stored log-sum-exp) through the C ABI against fp64 on the SAME bf16 operands: forward, log-sum-exp in natural units, the bf16
    output copy, and all three gradients (dQ with respect to the unscaled Q).  (2, 4096, 64) and (8, 512, 64) take the ping-pong
    forward and the 512-thread dK / dV kernel, (2, 384, 64) the generic D = 64 kernels, the others levels 2 and 3."""
    from hupr_amd import functional as F_
    L, rt = F_.rt.lib(), F_.rt
    (kb, qsb, vb, gb, v32, g32), (kr, qr, vr) = _qs_case(B, N, C, 300)
    sref = torch.einsum("bjc,bkc->bjk", kr, qr)
    outr = torch.einsum("bjc,bjk->bkc", vr, F.softmax(sref, 1))
    vres = v32.bfloat16().float()                          # the residual term is the fp32 map; use the rounded one on both sides
    if residual:
        outr = outr + vres.double()
    outr.backward(gb.double())
    dev = lambda t: t.cuda().contiguous()
    kd, qd, vd, gd, v32d = dev(kb), dev(qsb), dev(vb), dev(gb), dev(vres)
    out, lse = torch.full((B, N, C), float("nan"), device="cuda"), torch.full((B, N), float("nan"), device="cuda")
    o16 = torch.zeros((B, N, C), dtype=torch.bfloat16, device="cuda")
    rt.check(L.hupr_attn_fwd_bf16in_ld_ws_qs(rt.ptr(kd), C, rt.ptr(qd), C, rt.ptr(vd), rt.ptr(v32d) if residual else None, rt.ptr(out),
                                             rt.ptr(lse), rt.ptr(o16), C, B, N, C, None, 0, rt.stream()))
    close(out, outr, 1e-2, "QS forward")
    close(lse, torch.logsumexp(sref, 1), 1e-4, "QS log-sum-exp")
    assert torch.equal(o16, out.bfloat16())
    dk, dq, dv = (torch.full((B, N, C), float("nan"), device="cuda") for _ in range(3))
    scr = torch.empty((B, N), device="cuda")
    # (dout32 = null: the gradient arrived bf16-stored, dO is exact)
    rt.check(L.hupr_attn_bwd_bf16in_ld_qs(rt.ptr(kd), C, rt.ptr(qd), C, rt.ptr(vd), rt.ptr(gd), C, rt.ptr(v32d), rt.ptr(out), None,
                                          rt.ptr(lse), rt.ptr(dk), C, rt.ptr(dq), C, rt.ptr(dv), rt.ptr(scr), B, N, C,
                                          1 if residual else 0, 0, rt.stream()))
    # (the reference's residual term went through ``vres``, a constant: the kernel's dV also carries dO for it)
    close(dv, vr.grad + (gb.double() if residual else 0.0), 2e-2, "QS dV")
    close(dq, qr.grad, 3e-2, "QS dQ")
    close(dk, kr.grad, 3e-2, "QS dK")
    # against the plain kernels on the same K, V, dO and the unscaled bf16 Q: two roundings of the same attention
    qpl = dev((qsb.float() / 1.4426950408889634).bfloat16())
    out0, lse0 = torch.empty_like(out), torch.empty_like(lse)
    rt.check(L.hupr_attn_fwd_bf16in_ld_ws(rt.ptr(kd), C, rt.ptr(qpl), C, rt.ptr(vd), rt.ptr(v32d) if residual else None, rt.ptr(out0),
                                          rt.ptr(lse0), None, 0, B, N, C, None, 0, rt.stream()))
    close(out, out0, 1.5e-2, "QS vs plain forward")


def test_flash_attention_qs_deferred_maximum(bf16_math):
    """The ping-pong forward keeps the running maximum it entered a key tile with unless the tile exceeds it by more than 8 binary
    orders.  Keys whose norm grows along the key axis make every regime occur — slow growth (deferred: P up to 2^8), jumps (the
    rescale branch), and a first tile far below zero — and the largest logits reach a few hundred.  Against fp64 on the same operands,
    with the backward kernels consuming the log-sum-exp it produced."""
    from hupr_amd import functional as F_
    L, rt = F_.rt.lib(), F_.rt
    B, N, C = 8, 1024, 64
    prof = torch.cat([torch.linspace(0.05, 0.3, 256), torch.linspace(0.3, 3.0, 256), torch.full((256,), 6.0), torch.linspace(6.0, 0.1, 256)])
    (kb, qsb, vb, gb, v32, g32), (kr, qr, vr) = _qs_case(B, N, C, 340, kscale=1.0, profile=prof)
    sref = torch.einsum("bjc,bkc->bjk", kr, qr)
    assert sref.abs().max().item() > 100.0
    outr = torch.einsum("bjc,bjk->bkc", vr, F.softmax(sref, 1))
    outr.backward(gb.double())
    kd, qd, vd, gd, v32d = (t.cuda().contiguous() for t in (kb, qsb, vb, gb, vb.float()))
    out, lse = torch.full((B, N, C), float("nan"), device="cuda"), torch.full((B, N), float("nan"), device="cuda")
    rt.check(L.hupr_attn_fwd_bf16in_ld_ws_qs(rt.ptr(kd), C, rt.ptr(qd), C, rt.ptr(vd), None, rt.ptr(out), rt.ptr(lse), None, 0, B, N, C,
                                             None, 0, rt.stream()))
    assert torch.isfinite(out).all() and torch.isfinite(lse).all()
    close(out, outr, 1e-2, "deferred-maximum forward")
    close(lse, torch.logsumexp(sref, 1), 1e-4, "deferred-maximum log-sum-exp")
    dk, dq, dv = (torch.empty((B, N, C), device="cuda") for _ in range(3))
    scr = torch.empty((B, N), device="cuda")
    rt.check(L.hupr_attn_bwd_bf16in_ld_qs(rt.ptr(kd), C, rt.ptr(qd), C, rt.ptr(vd), rt.ptr(gd), C, rt.ptr(v32d), rt.ptr(out), None,
                                          rt.ptr(lse), rt.ptr(dk), C, rt.ptr(dq), C, rt.ptr(dv), rt.ptr(scr), B, N, C, 0, 0, rt.stream()))
    close(dv, vr.grad, 2e-2, "dV")
    close(dq, qr.grad, 3e-2, "dQ")
    close(dk, kr.grad, 3e-2, "dK")


@pytest.mark.parametrize("B,N,C", [(1, 4096, 64), (1, 1024, 128), (1, 256, 256)])
def test_attention_qs_split_keys(B, N, C, bf16_math):
    """Single-sample inference: the key-split form of the QS forward (shares in binary orders, merged by hupr_k_attn_combine<QS>)."""
    from hupr_amd import functional as F_
    L, rt = F_.rt.lib(), F_.rt
    (kb, qsb, vb, gb, v32, g32), (kr, qr, vr) = _qs_case(B, N, C, 380)
    sref = torch.einsum("bjc,bkc->bjk", kr, qr)
    ref = torch.einsum("bjc,bjk->bkc", vr, F.softmax(sref, 1))
    nbytes = L.hupr_attn_fwd_split_ws_bytes(B, N, C)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    kd, qd, vd = (t.cuda().contiguous() for t in (kb, qsb, vb))
    out, lse = torch.full((B, N, C), float("nan"), device="cuda"), torch.full((B, N), float("nan"), device="cuda")
    rt.check(L.hupr_attn_fwd_bf16in_ld_ws_qs(rt.ptr(kd), C, rt.ptr(qd), C, rt.ptr(vd), None, rt.ptr(out), rt.ptr(lse), None, 0, B, N, C,
                                             rt.ptr(ws), nbytes, rt.stream()))
    close(out, ref, 1e-2, "QS split forward")
    close(lse, torch.logsumexp(sref, 1), 1e-4, "QS split log-sum-exp")


@pytest.mark.parametrize("C,H", [(64, 16), (128, 16), (64, 32), (256, 16), (32, 8), (96, 16)])      # the last two: no fused attention kernel
def test_mscsa_level_fused_matches_composition(C, H, bf16_math):
    """MSCSALevelFn (one GEMM per map for its four 1x1 projections with bf16 epilogue, strided attention operands,
    in-place dV accumulation, one dgrad / wgrad GEMM per map) against the same level composed from ConvFn + AttentionFn:
    the forward is bit-identical (same products, same rounding points), the gradients differ only by summation order."""
    from hupr_amd import functional as F_
    B, N = 2, H * H
    ra, re = rnd(B, 1, H, H, C, seed=90).cuda(), rnd(B, 1, H, H, C, seed=91).cuda()
    ws = [rnd(C, C, 1, 1, seed=92 + i, scale=C ** -0.5).cuda() for i in range(8)]
    gs = [rnd(B, 1, H, H, C, seed=110 + i).cuda() for i in range(4)]

    def run(fused):
        a, e = ra.clone().requires_grad_(True), re.clone().requires_grad_(True)
        w = [t.clone().requires_grad_(True) for t in ws]
        if fused:
            assert F_.mscsa_level_fused_ok(a)
            outs = F_.MSCSALevelFn.apply(a, e, False, *w)
        else:
            conv = lambda x, wt: F_.conv(x, wt, None, None, (0, 0, 0))
            att = lambda k, q, v, res: F_.AttentionFn.apply(k.reshape(B, N, C), q.reshape(B, N, C), v.reshape(B, N, C),
                                                            res).reshape(B, 1, H, H, C)
            k_c_h, q_c_h, k_h, q_h = [conv(a, w[i]) for i in range(4)]
            k_c_v, q_c_v, k_v, q_v = [conv(e, w[4 + i]) for i in range(4)]
            outs = (att(k_c_h, q_c_v, a, True), att(k_h, q_h, a, False), att(k_c_v, q_c_h, e, True), att(k_v, q_v, e, False))
        sum((o * g).sum() for o, g in zip(outs, gs)).backward()
        return [o.detach() for o in outs], [a.grad, e.grad] + [t.grad for t in w]

    o0, g0 = run(False)
    prev = F_.QS_ATTN
    try:
        F_.QS_ATTN = False                      # the rounds-1-4 kernels inside the node: the same products, the same rounding points
        o1, g1 = run(True)
        for i, (x, y) in enumerate(zip(o1, o0)):
            assert torch.equal(x, y), "attention output %d differs" % i
        for i, (x, y) in enumerate(zip(g1, g0)):
            close(x, y, 2e-5, "gradient %d (0,1: maps; 2..9: projection weights)" % i)
        F_.QS_ATTN = True                       # the default (round 5): query projections carry log2(e) before their bf16 rounding and
        o2, g2 = run(True)                      # the forward keeps a deferred maximum — another rounding of the same attention
    finally:
        F_.QS_ATTN = prev
    # Two bf16 roundings of the same level: here the query projection is rounded after the factor log2(e), there before it, and with this
    # test's unnormalised unit-variance projections (logits up to +-30) a 2^-9 change of q moves a probability by several per cent — the
    # outputs differ by 1-7 % of their scale (measured; most at C = 256), the gradients by more.  The tight checks of the QS kernels are the fp64
    # comparisons on IDENTICAL operands (test_flash_attention_qs_kernels_vs_fp64: 1e-2 / 3e-2).
    for i, (x, y) in enumerate(zip(o2, o0)):
        close(x, y, 1e-1, "QS attention output %d" % i)          # (C = 256: 7 % measured)
    for i, (x, y) in enumerate(zip(g2, g0)):
        close(x, y, 2e-1, "QS gradient %d (0,1: maps; 2..9: projection weights)" % i)


@pytest.mark.parametrize("training", [True, False])
def test_bn_block_tail_bf16_activations(training):
    """BNActFn / BNAddBNReLUFn on bf16 storage == the fp32-storage kernels on the same (bf16-representable) data,
    up to the single rounding of each stored output."""
    from hupr_amd import functional as F_
    C, vox = 64, (2, 4, 8, 8)
    x1, x2 = _q(rnd(*vox, C, seed=80) * 2 + 0.5).cuda(), _q(rnd(*vox, C, seed=81)).cuda()
    gy = _q(rnd(*vox, C, seed=82)).cuda()
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        bn1, bn2 = torch.nn.BatchNorm3d(C).cuda(), torch.nn.BatchNorm3d(C).cuda()
        with torch.no_grad():
            for i, bn in enumerate((bn1, bn2)):
                bn.weight.copy_(rnd(C, seed=83 + i).cuda() * 0.3 + 1)
                bn.bias.copy_(rnd(C, seed=85 + i).cuda() * 0.2)
                bn.running_mean.copy_(rnd(C, seed=87 + i).cuda() * 0.1)
                bn.running_var.copy_(rnd(C, seed=89 + i).cuda().abs() + 0.5)
        a, b_ = x1.detach().to(dt).requires_grad_(True), x2.detach().to(dt).requires_grad_(True)
        h = F_.BNActFn.apply(a, bn1.weight, bn1.bias, bn1, training, True)
        y = F_.BNAddBNReLUFn.apply(h, bn2.weight, bn2.bias, bn2, b_, bn1.weight, bn1.bias, bn1, training)
        y.backward(gy.to(dt))
        outs[dt] = [t.detach().float() for t in (h, y, a.grad, b_.grad, bn1.weight.grad, bn2.bias.grad, bn1.running_var)]
        assert y.dtype == dt and a.grad.dtype == dt
    names = ["bn+relu", "block tail", "dx1", "dx2", "dgamma", "dbeta", "running_var"]
    for n, r, g in zip(names, outs[torch.float32], outs[torch.bfloat16]):
        # chained bf16 roundings h -> y -> grads; a rounding can flip a ReLU mask bit at y ~ 0, so gate the L2 error
        rel = ((g - r).double().norm() / r.double().norm()).item()
        assert rel <= 1.5e-2, "bf16act BN %s: rel-L2 %.3e" % (n, rel)
    close(outs[torch.bfloat16][0], outs[torch.float32][0], 4e-3, "bf16act BN first output: one rounding")


def test_interp_mnet_cast_bf16_activations():
    from hupr_amd import functional as F_
    x = _q(rnd(2, 8, 16, 16, 64, seed=90)).cuda()
    gy = _q(rnd(2, 4, 8, 8, 64, seed=91)).cuda()
    r = {}
    for dt in (torch.float32, torch.bfloat16):
        xi = x.detach().to(dt).requires_grad_(True)
        y = F_.interp(xi, (4, 8, 8))
        y.backward(gy.to(dt))
        assert y.dtype == dt and xi.grad.dtype == dt
        r[dt] = (y.detach().float(), xi.grad.float())
    assert torch.equal(r[torch.bfloat16][0], _q(r[torch.float32][0]))
    assert torch.equal(r[torch.bfloat16][1], _q(r[torch.float32][1]))
    # casts round-trip exactly on bf16-representable data and carry the gradient back in the input dtype
    xc = x.clone().requires_grad_(True)
    yb = F_.cast(xc, torch.bfloat16)
    assert yb.dtype == torch.bfloat16 and torch.equal(yb.float(), x)
    back = F_.cast(yb, torch.float32)
    back.backward(torch.ones_like(back))
    assert xc.grad.dtype == torch.float32 and torch.equal(xc.grad, torch.ones_like(x))
    # MNet front end writing bf16 / reading a bf16 gradient
    vin = rnd(2, 8, 8, 2, 16, 16, 8, seed=92).cuda()
    w, b = (rnd(32, 2, 2, 1, 1, seed=93) * 0.5).cuda(), rnd(32, seed=94).cuda()
    gm = _q(rnd(2, 8, 16, 16, 32, seed=95)).cuda()
    m = {}
    for dt in (torch.float32, torch.bfloat16):
        wi, bi = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        o = F_.MNetFn.apply(vin, wi, bi, dt)
        o.backward(gm.to(dt))
        m[dt] = (o.detach().float(), wi.grad, bi.grad)
    assert torch.equal(m[torch.bfloat16][0], _q(m[torch.float32][0]))
    assert torch.equal(m[torch.bfloat16][1], m[torch.float32][1]) and torch.equal(m[torch.bfloat16][2], m[torch.float32][2])


@pytest.mark.parametrize("shape", [(32, 64, 8, 64, 64, True), (7, 128, 4, 32, 48, False)])
def test_conv_first_layer_shape_32_input_channels(shape, bf16_math):
    """Ci = 32 (the encoders' first convolution, `layers.py:236`): the 256-voxel kernel's 64-byte-row form (one K-step per tap, a stage = one kz
    plane of nine taps, three fragment banks, bias added in front of the parked tile's rounding; round 6) against the 128-voxel kernel on
    the same bf16 operands — same products, another fp32 summation order: equal up to one rounding of the bf16 store — and against fp64;
    the bench shape with bias, and two output tiles over an uneven tile count without; deterministic."""
    from hupr_amd import functional as F_
    L = F_.rt.lib()
    B, Co, D, H, W, with_bias = shape
    Ci = 32
    x = _q(rnd(B, D, H, W, Ci, seed=90)).cuda().bfloat16()
    w = rnd(Co, Ci, 3, 3, 3, seed=91, scale=(Ci * 27) ** -0.5).cuda()
    bias = rnd(Co, seed=92).cuda() if with_bias else None
    run = lambda: F_._conv_raw(x, w, 0, bias, None, Co, (3, 3, 3), (1, 1, 1), (D, H, W))
    try:
        y256 = run()
        y256b = run()
        L.hupr_debug_halo_tiles(15)           # bit 4 cleared: the 128-voxel kernel takes the launch
        y128 = run()
    finally:
        L.hupr_debug_halo_tiles(31)
    assert torch.equal(y256, y256b) and not torch.equal(y256, y128)
    ulp = y128.float().abs().clamp_min(2.0 ** -6) * 2.0 ** -7           # one bf16 step of the stored value
    d = (y256.float() - y128.float()).abs()
    assert bool((d <= ulp).all()), (d / ulp).max().item()
    assert (d > 0).float().mean().item() < 0.05                           # and almost all of them identical
    ref = F.conv3d(_bf16_round(ncdhw(x.float().cpu()))[:1], _bf16_round(w.cpu()), bias.cpu().double() if with_bias else None, 1, 1)
    close(ncdhw(y256.float().cpu())[:1], ref, 6e-3, "halo256m<KC = 32> vs fp64 (one bf16 store rounding)")
    close(ncdhw(y128.float().cpu())[:1], ref, 6e-3, "128-voxel kernel vs fp64 (one bf16 store rounding)")


def test_merge_down_node_matches_separate_nodes(bf16_math):
    """F_.MergeDownFn (temporal merge + next level's down-sampling of one encoder map as ONE autograd node) against the two
    separate nodes: forward bit-identical; the input gradient differs only by where the bf16 rounding of the sum happens
    (accumulated in fp32 inside the resampling backward instead of bf16 + bf16 by autograd), weight gradient identical."""
    from hupr_amd import functional as F_
    B, G, H, W, C = 2, 8, 32, 32, 64
    x0 = _q(rnd(B, G, H, W, C, seed=95)).cuda().bfloat16()
    w0 = rnd(C, C, G, 1, 1, seed=96, scale=(C * G) ** -0.5).cuda()
    gm = rnd(B, 1, H, W, C, seed=97).cuda()
    gd = _q(rnd(B, G // 2, H // 2, W // 2, C, seed=98)).cuda().bfloat16()
    outs = []
    for fused in (True, False):
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        if fused:
            m, d = F_.MergeDownFn.apply(x, w, (G // 2, H // 2, W // 2))
        else:
            m, d = F_.temporal_merge(x, w), F_.interp(x, (G // 2, H // 2, W // 2))
        torch.autograd.backward((m, d), (gm, gd))
        outs.append((m.detach(), d.detach(), x.grad.float(), w.grad))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
    # two bf16 roundings (each addend) + one of the sum on the separate path, two on the fused one: a couple of bf16 steps of
    # the larger addend, which can exceed the (possibly cancelled) sum's own step
    d = (a[2] - b[2]).abs()
    assert d.max().item() <= 2.0 ** -6 * b[2].abs().max().item() and d.mean().item() <= 2.0 ** -9 * b[2].abs().mean().item() * 4


@pytest.mark.parametrize("B,H", [(3, 64), (1, 8)])
def test_head1x1_fp32_kernels(B, H):
    """The dedicated fp32 1x1 head (32 -> 14 key-point channels padded to 16; reference models/layers.py:94) against fp64:
    forward, input gradient and weight gradient, incl. a voxel count that is not a multiple of the workgroup slices."""
    from hupr_amd import functional as F_
    x0 = rnd(B, 1, H, H, 32, seed=300).cuda()
    w0 = rnd(14, 32, 1, 1, seed=301, scale=32 ** -0.5).cuda()               # the parameter: 14 filters (reference models/layers.py:94)
    gy = rnd(B, 1, H, H, 16, seed=302).cuda()
    x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    w16 = F_._head_w16_cached(w)                                             # zero-padded copy, a pack-table entry (round 5)
    assert tuple(w16.shape) == (16, 32, 1, 1) and torch.equal(w16[:14], w0) and w16[14:].abs().max().item() == 0.0
    y = F_.Head1x1Fn.apply(x, w, w16)
    y.backward(gy)
    xd, wd = x0.double(), w0.double().reshape(14, 32)
    close(y[..., :14], xd @ wd.t(), 2e-6, "head forward")
    wd16 = torch.cat([wd, torch.zeros(2, 32, dtype=torch.float64, device=wd.device)])
    close(x.grad, gy.double() @ wd16, 2e-6, "head input gradient")
    # the weight gradient is written for the 14 real filters only (the parameter's own slot in a flat gradient bucket)
    assert tuple(w.grad.shape) == (14, 32, 1, 1)
    close(w.grad.reshape(14, 32), (gy.double().reshape(-1, 16).t() @ xd.reshape(-1, 32))[:14], 5e-6, "head weight gradient")
    assert y[..., 14:].abs().max().item() == 0.0
    # a weight update behind torch's back (what FusedAdam does) reaches the padded copy with the next table refresh
    with torch.no_grad():
        w.mul_(2.0)
    assert torch.equal(F_._head_w16_cached(w)[:14], w.detach())
    with torch.no_grad():
        w.copy_(w0)
    F_._head_w16_cached(w)
    # inside a bf16 run the model's head region goes through these kernels, the fp32 parity path through the generic convolution
    F_.set_math("bf16")
    try:
        with F_.region("head"):
            assert F_._REGION_SWITCHED and F_.MATH == "f32"
            y2 = F_.head_conv(x0, w, 14)
        assert torch.equal(y2.detach(), y.detach())
    finally:
        F_.set_math("f32")
    assert not F_._REGION_SWITCHED


@pytest.mark.parametrize("shape,Ci,Co,k", [((1, 2, 16, 16), 256, 256, 3), ((1, 4, 32, 32), 128, 128, 3), ((1, 4, 32, 32), 64, 128, 3),
                                           ((1, 1, 16, 16), 1024, 256, 1), ((1, 1, 32, 32), 640, 128, 1), ((1, 1, 64, 64), 320, 64, 1),
                                           ((2, 1, 16, 16), 256, 128, 1)])
def test_small_grid_convolution_k_slices(shape, Ci, Co, k, bf16_math):
    """Single-sample inference (config C2): the halo convolution splits its reduction over (channel chunk, kz plane) slices on
    grids that would leave the chip idle (hupr_conv3x3_halo_bf16act_ws + hupr_k_conv_partial_reduce).  Against the one-launch
    form (same products; fp32 summation order differs, so outputs agree to one bf16 rounding) and against fp64, with a residual
    and a bias riding in the reduce kernel."""
    from hupr_amd import functional as F_
    B, D, H, W = shape
    L = F_.rt.lib()
    assert L.hupr_conv3x3_halo_splitk_ws_bytes(B, D, H, W, Ci, Co, k) > 0
    x = cl(rnd(B, Ci, D, H, W, seed=400)).cuda().bfloat16()
    w = rnd(Co, Ci, k, 3, 3, seed=401, scale=(Ci * 9 * k) ** -0.5).cuda()
    bias = rnd(Co, seed=402).cuda()
    res = cl(rnd(B, Co, D, H, W, seed=403)).cuda().bfloat16()
    pad = (k // 2, 1, 1)
    with torch.no_grad():
        y1 = F_.conv(x, w, bias, res, pad)
        L.hupr_debug_halo_split_k(0)
        try:
            y0 = F_.conv(x, w, bias, res, pad)
        finally:
            L.hupr_debug_halo_split_k(1)
    ref = torch.nn.functional.conv3d(ncdhw(x.double()), w.double(), bias.double(), padding=pad) + ncdhw(res.double())
    close(ncdhw(y1.float()), ref, 1.2e-2, "sliced convolution vs fp64 (bf16 products)")
    close(y1.float(), y0.float(), 8e-3, "sliced vs one-launch")
    assert (y1 != y0).float().mean().item() < 0.2                 # mostly identical bits; differences are single bf16 roundings
    # training keeps the one-launch form (gradients enabled: no slicing, bit-identical to before)
    y2 = F_.conv(x, w.requires_grad_(True), bias, res, pad)
    assert torch.equal(y2.detach(), y0)


@pytest.mark.parametrize("B,H", [(3, 64), (1, 16), (32, 64)])
def test_streaming_temporal_merge_matches_the_generic_kernel(B, H, bf16_math):
    """hupr_tmerge_fwd_stream_bf16 (LDS-DMA ring, persistent workgroups; the level-1 merge Conv3d(64, 64, (8,1,1)), reference
    models/layers.py:208,218) against fp64 on the bf16-rounded operands and against the generic mixed-storage convolution it
    replaces (same bf16 products, fp32 accumulation in another order)."""
    from hupr_amd import functional as F_
    G, C = 8, 64
    x = rnd(B, G, H, H, C, seed=500).cuda().bfloat16()
    w = rnd(C, C, G, 1, 1, seed=501, scale=(C * G) ** -0.5).cuda().requires_grad_(True)
    assert F_.rt.lib().hupr_tmerge_stream_supported(G, H * H, C, C)
    y1 = F_.TemporalMergeFn.apply(x, w).detach()
    F_.TMERGE_STREAM = False
    try:
        y0 = F_.TemporalMergeFn.apply(x, w).detach()
    finally:
        F_.TMERGE_STREAM = True
    ref = torch.einsum("bghwc,ocg->bhwo", x.double().cpu(), w.detach().reshape(C, C, G).to(torch.bfloat16).double().cpu())
    close(y1.reshape(B, H, H, C), ref, 1e-5, "streaming merge vs fp64 (bf16-rounded operands)")
    close(y1, y0, 2e-6, "streaming vs generic")
    # backward: dx (bf16 store) and dW (fp32) of the streaming kernels against fp64 on the bf16-rounded operands, and against
    # the generic kernels (same products, other summation order)
    dy = rnd(B, 1, H, H, C, seed=502).cuda()
    xg = x.clone().requires_grad_(True)

    def grads():
        w.grad = None
        xg.grad = None
        F_.TemporalMergeFn.apply(xg, w).backward(dy)
        return xg.grad.float().cpu(), w.grad.detach().clone().cpu()
    dx1, dw1 = grads()
    F_.TMERGE_STREAM = False
    try:
        dx0, dw0 = grads()
    finally:
        F_.TMERGE_STREAM = True
    dyr = dy.to(torch.bfloat16).double().cpu().reshape(B, H, H, C)
    wr = w.detach().reshape(C, C, G).to(torch.bfloat16).double().cpu()
    dx_ref = torch.einsum("bhwo,ocg->bghwc", dyr, wr)
    dw_ref = torch.einsum("bhwo,bghwc->ocg", dyr, x.double().cpu()).reshape(C, C, G, 1, 1)
    close(dx1, dx_ref, 6e-3, "streaming dgrad vs fp64 (one bf16 rounding of the result)")
    close(dx1, dx0, 6e-3, "streaming vs generic dgrad")
    close(dw1, dw_ref, 5e-5, "streaming wgrad vs fp64 (bf16-rounded operands)")
    close(dw1, dw0, 5e-5, "streaming vs generic wgrad")
    dx2, dw2 = grads()
    assert torch.equal(dx1, dx2) and torch.equal(dw1, dw2), "the streaming backward is deterministic"


@pytest.mark.parametrize("shape", [(32, 4096, 64), (32, 1024, 128), (1, 4096, 64), (3, 1008, 128), (2, 72, 64)])
def test_mscsa_projection_stream_kernel(shape, bf16_math):
    """Round 6: the four 1 x 1 projections of an MSCSA map (reference models/layers.py:150-157) as one streaming product
    (csrc/projection.hip: weights resident in LDS, rows straight from global memory into MFMA fragments) against the implicit-GEMM engine
    it replaces — same operand roundings, another fp32 order inside a row: a few outputs one bf16 step apart — and against fp64 on the
    bf16-rounded operands; a row count that is not a multiple of the 64-row workgroup step; deterministic."""
    from hupr_amd import functional as F_
    L, rt = F_.rt.lib(), F_.rt
    B, N, C = shape
    M = B * N
    x = rnd(M, C, seed=800).cuda()
    wc = rnd(4 * C, C, seed=801, scale=C ** -0.5).cuda()
    assert L.hupr_mscsa_proj_supported(M, C)
    y = torch.full((M, 4 * C), float("nan"), dtype=torch.bfloat16, device="cuda")
    y2 = torch.full_like(y, float("nan"))
    rt.check(L.hupr_mscsa_proj_fwd_bf16(rt.ptr(x), rt.ptr(wc), rt.ptr(y), M, C, rt.stream()))
    rt.check(L.hupr_mscsa_proj_fwd_bf16(rt.ptr(x), rt.ptr(wc), rt.ptr(y2), M, C, rt.stream()))
    assert torch.equal(y, y2) and torch.isfinite(y.float()).all()
    ref = _bf16_round(x.cpu()).double() @ _bf16_round(wc.cpu()).double().t()
    close(y.float().cpu(), ref, 1e-2, "projection (bf16 store) vs fp64")
    if N % 64 == 0:
        H = int(round(N ** 0.5))
        ye = torch.empty_like(y)
        rt.check(L.hupr_conv_fwd_bf16_mixed(rt.ptr(x), 0, rt.ptr(wc), None, rt.ptr(ye), 1, B, 1, H, N // H, C, C, 1, H, N // H, 4 * C, 4 * C,
                                            1, 1, 1, 0, 0, 0, rt.stream()))
        d = (y.float() - ye.float()).abs()
        assert (d > 0).float().mean().item() < 5e-3 and (d <= torch.maximum(y.float().abs(), ye.float().abs()) * 2 ** -7 + 1e-6).all()


@pytest.mark.parametrize("shape", [(32, 4096, 64), (1, 4096, 64), (3, 1008, 64), (2, 72, 64)])
def test_mscsa_projection_input_gradient_stream_kernel(shape, bf16_math):
    """The input gradient of the fused projections, dX = dY . Wc + dV (functional.MSCSALevelFn.backward), as a streaming kernel against
    the GEMM engine call it replaces (same operand roundings, fp32 accumulate; another K order: 1e-6-relative differences) and against
    fp64 on the bf16-rounded operands; with and without the residual term; deterministic."""
    from hupr_amd import functional as F_
    L, rt = F_.rt.lib(), F_.rt
    B, N, C = shape
    M = B * N
    dy = rnd(M, 4 * C, seed=810).cuda()
    wc = rnd(4 * C, C, seed=811, scale=C ** -0.5).cuda()
    dv = rnd(M, C, seed=812).cuda()
    ref = _bf16_round(dy.cpu()).double() @ _bf16_round(wc.cpu()).double()
    for res in (dv, None):
        dx = torch.full((M, C), float("nan"), device="cuda")
        dx2 = torch.full_like(dx, float("nan"))
        rt.check(L.hupr_mscsa_proj_dgrad_f32(rt.ptr(dy), rt.ptr(wc), rt.ptr(res) if res is not None else None, rt.ptr(dx), M, C, rt.stream()))
        rt.check(L.hupr_mscsa_proj_dgrad_f32(rt.ptr(dy), rt.ptr(wc), rt.ptr(res) if res is not None else None, rt.ptr(dx2), M, C, rt.stream()))
        assert torch.equal(dx, dx2) and torch.isfinite(dx).all()
        want = ref + (dv.cpu().double() if res is not None else 0.0)
        close(dx.cpu(), want, 2e-5, "projection input gradient vs fp64")
        de = torch.empty_like(dx)
        rt.check(L.hupr_gemm_bf16(0, 0, rt.ptr(dy), rt.ptr(wc), rt.ptr(de), M, C, 4 * C, 4 * C, C, C, 1, 0, 0, 0,
                                  rt.ptr(res) if res is not None else None, C, 0, 0, rt.stream()))
        close(dx, de, 2e-5, "projection input gradient vs the GEMM engine")
