"""GPU (-m gpu): FFT-chain HIP kernels vs the oracle and the reference-derived golden vectors.
Gates: rel-L2 <= 1e-5 on Doppler bins != 8, |bin 8| <= 1e-6 * max|all|.  Bin 8 (zero Doppler; the loader's
slot f = 4) is clutter-nulled: the reference holds fp64 rounding residue there, the chain a frame-keyed dither of the same
statistics (include/hupr.h "The zero-Doppler bin"; oracle.fft_chain.zero_doppler_dither restates it) — gated against that
restatement to the same rel-L2, and through the reference loader's Normalize arithmetic (finite, unit variance)."""
import json
import os

import numpy as np
import pytest
import torch

from hupr_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _run(iq):
    from hupr_amd import preprocessing
    return preprocessing.fft_chain(torch.from_numpy(iq).cuda()).cpu().numpy()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fft_chain_vs_golden_and_oracle(seed):
    from oracle import fft_chain as offt
    iq = synth.adc_cube_int16(seed)
    got = _run(iq)[0]
    assert got.shape == (16, 64, 64, 8) and got.dtype == np.complex64
    ref = offt.generate_heatmap(synth.adc_cube_complex(iq)[0])
    keep = np.arange(16) != 8
    rel = np.linalg.norm(got[keep] - ref[keep]) / np.linalg.norm(ref[keep])
    assert rel <= 1e-5, rel
    assert np.abs(got[8]).max() <= 1e-6 * np.abs(ref).max()
    dz = offt.generate_heatmap_dithered(iq[0])[8]                       # the zero-Doppler plane: the dither's restatement
    assert np.linalg.norm(got[8] - dz) / np.linalg.norm(dz) <= 1e-5
    assert 0.1 <= np.linalg.norm(got[8]) / np.linalg.norm(ref[8]) <= 10.0      # of the reference residue's order of magnitude
    # per-element worst case on the kept bins
    assert np.abs(got[keep] - ref[keep]).max() <= 2e-5 * np.abs(ref).max()
    g = np.load(os.path.join(G, "fft_seed%d.npz" % seed))
    samp = got.reshape(-1)[::int(g["stride"])]
    idx = np.arange(got.size)[::int(g["stride"])] // (64 * 64 * 8)
    ok = idx != 8
    assert np.abs(samp[ok] - g["sample"][ok]).max() <= 2e-5 * np.abs(ref).max()
    l2 = np.sqrt((np.abs(got.astype(np.complex128)) ** 2).sum(axis=(1, 2, 3)))
    np.testing.assert_allclose(l2[keep], g["doppler_l2"][keep], rtol=1e-5)


def test_point_target_known_answer():
    g = np.load(os.path.join(G, "fft_point.npz"))
    tg = json.loads(str(g["targets"]))
    got = _run(synth.point_target_cube(tg))[0]
    mag = np.abs(got).sum(axis=3)
    assert np.unravel_index(mag.argmax(), mag.shape) == (11, 34, 21) == tuple(g["peak"])
    ref = g["sample"]
    samp = got.reshape(-1)[::int(g["stride"])]
    assert np.abs(samp - ref).max() <= 2e-5 * np.abs(ref).max()


def test_batch_is_independent_and_linear():
    """size-independent properties at a larger batch: per-frame independence, linearity."""
    a = synth.adc_cube_int16(5, nframes=8) // 2
    b = synth.adc_cube_int16(6, nframes=8) // 2
    ya, yb, yab = _run(a), _run(b), _run((a + b).astype(np.int16))
    scale = np.abs(yab).max()
    assert np.abs(yab - (ya + yb)).max() <= 2e-5 * scale
    single = _run(a[3:4])
    assert np.array_equal(single[0], ya[3])           # bit-identical regardless of batch position
    # a constant-over-chirps scene is pure clutter -> all Doppler bins ~ 0
    const = np.tile(synth.adc_cube_int16(9)[:, :, :3], (1, 1, 64, 1, 1))
    yc = _run(const)
    assert np.abs(yc).max() <= 1e-5 * scale


def test_empty_and_bad_inputs():
    from hupr_amd import preprocessing, runtime
    out = preprocessing.fft_chain(torch.zeros((0, 4, 192, 256, 2), dtype=torch.int16, device="cuda"))
    assert out.shape == (0, 16, 64, 64, 8)
    with pytest.raises(ValueError):
        preprocessing.fft_chain(torch.zeros((1, 4, 192, 256), dtype=torch.int16, device="cuda"))
    ws = torch.empty(16, dtype=torch.uint8, device="cuda")
    with pytest.raises(runtime.HuprError):
        preprocessing.fft_chain(torch.zeros((1, 4, 192, 256, 2), dtype=torch.int16, device="cuda"), ws=ws)


def test_extreme_amplitudes():
    from oracle import fft_chain as offt
    iq = np.full((1, 4, 192, 256, 2), 32767, dtype=np.int16)
    iq[:, :, ::2] = -32768                                # alternate chirps: max Doppler energy
    got = _run(iq)[0]
    ref = offt.generate_heatmap(synth.adc_cube_complex(iq)[0])
    assert np.isfinite(got.view(np.float32)).all()
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-3


def test_fused_loader_vs_oracle():
    from hupr_amd import preprocessing
    from oracle import fft_chain as offt, loader as oloader
    iq = synth.adc_cube_int16(0, nframes=2)
    dev = torch.from_numpy(iq).cuda()
    got = preprocessing.fft_chain_loader(dev).cpu().numpy()
    assert got.shape == (2, 8, 2, 64, 64, 8) and got.dtype == np.float32
    f_ok = np.arange(8) != 4                       # f=4 is the clutter-nulled bin: normalised noise
    for n in range(2):
        ref = oloader.loader_transform(offt.generate_heatmap(synth.adc_cube_complex(iq)[n]))
        assert np.abs(got[n][f_ok] - ref[f_ok]).max() <= 2e-4
        dz = oloader.loader_transform(offt.generate_heatmap_dithered(iq[n]))     # slot 4: the normalised dither plane
        assert np.abs(got[n] - dz).max() <= 5e-4
    flat = got.reshape(2, 8, 2, 4096, 8).astype(np.float64)
    assert np.isfinite(got).all()
    assert np.abs(flat.mean(axis=3)).max() < 1e-4
    assert np.abs(flat.std(axis=3, ddof=1) - 1).max() < 1e-4                     # every slot, f = 4 included
    # unfused route (complex cube -> loader glue) agrees with the fused one
    two = preprocessing.loader_normalize(preprocessing.fft_chain(dev)).cpu().numpy()
    assert np.abs(two - got).max() <= 5e-4
    g = np.load(os.path.join(G, "loader_seed0.npz"))
    samp = got[0].reshape(-1)[::int(g["stride"])]
    fi = np.arange(got[0].size)[::int(g["stride"])] // (2 * 64 * 64 * 8)
    assert np.abs(samp[fi != 4] - g["full_sample"][fi != 4]).max() <= 2e-4


def test_normalize_operator_mirror():
    from hupr_amd.datasets import Normalize
    from oracle import loader as oloader
    g = np.load(os.path.join(G, "loader_seed0.npz"))
    x = torch.from_numpy(g["slice_in"].astype(np.float32)).permute(2, 0, 1).contiguous().cuda()
    y = Normalize()(x).permute(1, 2, 0).cpu().numpy()
    assert np.abs(y - g["slice_out"]).max() <= 1e-4


def test_radar_object_dropin():
    from hupr_amd.preprocessing import RadarObject
    from oracle import fft_chain as offt
    fr = synth.adc_cube_complex(synth.adc_cube_int16(4))[0]
    out = RadarObject().generateHeatmap(fr)
    assert out.shape == (16, 64, 64, 8) and out.dtype == np.complex128
    ref = offt.generate_heatmap(fr)
    keep = np.arange(16) != 8
    assert np.linalg.norm(out[keep] - ref[keep]) / np.linalg.norm(ref[keep]) <= 1e-5


def test_dca1000_ingest_bit_exact(tmp_path):
    """raw .bin -> GPU de-interleave is pure integer shuffling: bit-exact vs the oracle; then the FFT chain runs on it."""
    from hupr_amd import preprocessing
    from oracle import dca1000
    raw = synth.randint((3 * 4 * 192 * 256 * 2,), -2048, 2048, "dca_gpu").astype(np.int16)
    dev = preprocessing.dca1000_frames(torch.from_numpy(raw).cuda())
    assert dev.shape == (3, 4, 192, 256, 2)
    assert np.array_equal(dev.cpu().numpy(), dca1000.frames_int16(raw))
    raw.tofile(tmp_path / "adc_data.bin")
    z = preprocessing.RadarObject().getadcDataFromDCA1000(str(tmp_path))
    assert z.shape == (4, 576, 256) and np.array_equal(z, dca1000.parse_dca1000(raw))
    assert preprocessing.dca1000_frames(torch.zeros(0, dtype=torch.int16, device="cuda")).shape[0] == 0
    cube = preprocessing.fft_chain(dev)
    assert cube.shape == (3, 16, 64, 64, 8) and torch.isfinite(torch.view_as_real(cube)).all()


def test_sequence_fft_cache_matches_uncached_loader():
    """Windows assembled from the per-sequence cache == the un-cached loader on the gathered ADC frames, incl. the
    sequence-edge clamping of the reference (datasets/dataset.py:120-139)."""
    from hupr_amd import preprocessing, synth
    from hupr_amd.datasets import SequenceFFTCache, window_indices
    dur, G = 12, 8
    adc_h = torch.from_numpy(synth.adc_cube_int16(40, sensor=0, nframes=dur)).cuda()
    adc_v = torch.from_numpy(synth.adc_cube_int16(40, sensor=1, nframes=dur)).cuda()
    cache = SequenceFFTCache(adc_h, adc_v, G)
    hb, vb = cache.batch([0, 5, dur - 1])
    for k, index in enumerate((0, 5, dur - 1)):
        idx = torch.tensor(window_indices(index, dur, G)).cuda()
        ref_h = preprocessing.fft_chain_loader(adc_h.index_select(0, idx))
        ref_v = preprocessing.fft_chain_loader(adc_v.index_select(0, idx))
        h, v = cache.window(index)
        assert torch.equal(h, ref_h) and torch.equal(v, ref_v)
        assert torch.equal(hb[k], ref_h) and torch.equal(vb[k], ref_v)


@pytest.mark.parametrize("window,flags", [("hann", 3), ("range", 1), ("doppler", 2)])
def test_optin_hanning_window_and_magnitude_vs_oracle(window, flags):
    """north_star's "Hanning windowing and magnitude" exist as opt-in flags (the reference has neither,
    process_iwr1843.py:130-151); default OFF = the parity path.  Gate as for the plain chain: rel-L2 <= 1e-5."""
    from hupr_amd import preprocessing
    from oracle import fft_chain as offt
    iq = synth.adc_cube_int16(1)
    dev = torch.from_numpy(iq).cuda()
    ref = offt.generate_heatmap(synth.adc_cube_complex(iq)[0], window=flags)
    got = preprocessing.fft_chain(dev, window=window)[0].cpu().numpy()
    keep = np.arange(16) != 8 if not flags & 2 else np.ones(16, bool)      # a Doppler window un-nulls bin 0 (leakage)
    rel = np.linalg.norm(got[keep] - ref[keep]) / np.linalg.norm(ref[keep])
    assert rel <= 1e-5, rel
    mag = preprocessing.fft_chain(dev, window=window, magnitude=True)[0].cpu().numpy()
    assert mag.dtype == np.float32 and mag.shape == (16, 64, 64, 8)
    rm = offt.generate_heatmap(synth.adc_cube_complex(iq)[0], window=flags, magnitude=True)
    assert np.abs(mag[keep] - rm[keep]).max() <= 2e-5 * rm.max()
    np.testing.assert_allclose(mag, np.abs(got), rtol=2e-6, atol=1e-6 * rm.max())
    # the windowed loader variant == loader glue applied to the windowed cube
    ld = preprocessing.fft_chain_loader(dev, window=window)[0].cpu().numpy()
    ld_ref = preprocessing.loader_normalize(preprocessing.fft_chain(dev, window=window))[0].cpu().numpy()
    f_ok = np.arange(8) != 4 if not flags & 2 else np.ones(8, bool)
    assert np.abs(ld[f_ok] - ld_ref[f_ok]).max() <= 2e-3


def test_window_flags_default_off_is_bit_identical_and_hann_suppresses_sidelobes():
    from hupr_amd import preprocessing, runtime as rt
    iq = synth.adc_cube_int16(2)
    dev = torch.from_numpy(iq).cuda()
    a = preprocessing.fft_chain(dev)
    b = preprocessing.fft_chain(dev, window=None, magnitude=False)
    assert torch.equal(torch.view_as_real(a), torch.view_as_real(b))
    with pytest.raises(ValueError):
        preprocessing.fft_chain(dev, window="blackman")
    # a point target between range bins: Hann trades main-lobe width for > 20 dB lower far sidelobes along range
    tg = [dict(range_bin=60.5, doppler_bin=3, az_bin=10, el_bin=2, amp=800.0)]
    cube = torch.from_numpy(synth.point_target_cube(tg)).cuda()
    plain = preprocessing.fft_chain(cube, magnitude=True)[0].sum(dim=(2, 3))[11].cpu().numpy()       # (range,) at the target's Doppler bin
    hann = preprocessing.fft_chain(cube, window="range", magnitude=True)[0].sum(dim=(2, 3))[11].cpu().numpy()
    pk = int(plain.argmax())
    far = np.r_[0:max(pk - 8, 0), min(pk + 9, 64):64]
    assert (hann[far].max() / hann.max()) < 0.1 * (plain[far].max() / plain.max())
    # flags the C ABI must refuse
    L = rt.lib()
    out = torch.empty((1, 8, 2, 64, 64, 8), device="cuda")
    ws = torch.empty(L.hupr_fft_chain_ws_bytes(1), dtype=torch.uint8, device="cuda")
    assert L.hupr_fft_chain_opts(rt.ptr(dev), 1, rt.ptr(out), 4, 1, rt.ptr(ws), ws.numel(), None) == -1      # loader + magnitude
    assert L.hupr_fft_chain_opts(rt.ptr(dev), 1, rt.ptr(out), 64, 0, rt.ptr(ws), ws.numel(), None) == -1     # unknown flag


def test_fused_elevation_mean_loader_is_bit_identical_to_loader_plus_mnet_mean():
    """hupr_fft_chain_loader_means_f32 (FFT chain + Normalize + HuPRNet's elevation mean, models/networks.py:26-27) against the
    two-step path it replaces: same planes as averaging the stored loader tensor with the MNet kernel's association, and the
    MNet front end fed by them produces the same bits (output and the means it keeps for the backward pass)."""
    from hupr_amd import functional as F_, preprocessing
    iq = np.concatenate([synth.adc_cube_int16(4, frame=f) for f in range(16)])
    dev = torch.from_numpy(iq).cuda()
    full = preprocessing.fft_chain_loader(dev)                                   # (16, 8, 2, 64, 64, 8)
    planes = preprocessing.fft_chain_loader_means(dev)                           # (16, 16, 64, 64)
    assert planes.shape == (16, 16, 64, 64) and planes.dtype == torch.float32
    x = full.reshape(16, 16, 64, 64, 8)
    want = (((x[..., 0] + x[..., 1]) + (x[..., 2] + x[..., 3])) + ((x[..., 4] + x[..., 5]) + (x[..., 6] + x[..., 7]))) * 0.125
    assert torch.equal(planes, want)
    # through the MNet front end (train mode: weights require grad -> the pixel-major means are kept)
    torch.manual_seed(0)
    w = torch.randn(32, 2, 2, 1, 1, device="cuda", requires_grad=True)
    b = torch.randn(32, device="cuda", requires_grad=True)
    for dt in (torch.float32, torch.bfloat16):
        ya = F_.MNetFn.apply(full.view(2, 8, 8, 2, 64, 64, 8), w, b, dt)
        yb = F_.MNetFn.apply(planes.view(2, 8, 16, 64, 64), w, b, dt)
        assert torch.equal(ya, yb)
        ga = torch.autograd.grad(ya.float().square().sum(), (w, b))
        gb = torch.autograd.grad(yb.float().square().sum(), (w, b))
        assert torch.equal(ga[0], gb[0]) and torch.equal(ga[1], gb[1])


def test_zero_doppler_plane_conventions():
    """The zero-Doppler bin (index 8; the loader's slot f = 4).  Static clutter removal (process_iwr1843.py:122-128) is exact on
    integer ADC samples in the Doppler-first chain, so the bin would be EXACTLY zero — 0/0 = NaN in the reference's Normalize
    (datasets/base.py:17-24), where the reference itself feeds the network its normalised fp64 rounding residue.
      default        a frame-keyed dither of the residue's statistics: finite, non-constant, unit variance after Normalize, a
                     pure function of the frame (bit-identical whatever the batch position), different from frame to frame;
      "exact"        exactly zero; the loader epilogues emit zeros, never 0/0 (round 3's behaviour, opt-in);
      "range_first"  the rounds-1/2 kernel order, the bin = that order's own fp32 rounding residue.
    All three agree on every other bin (the two Doppler-first modes bit for bit)."""
    from hupr_amd import preprocessing
    dev = torch.from_numpy(np.concatenate([synth.adc_cube_int16(9, frame=f) for f in range(4)])).cuda()
    cube = preprocessing.fft_chain(dev)
    ld = preprocessing.fft_chain_loader(dev)
    pl = preprocessing.fft_chain_loader_means(dev)
    assert torch.isfinite(torch.view_as_real(cube)).all() and torch.isfinite(ld).all() and torch.isfinite(pl).all()
    keep = [i for i in range(16) if i != 8]
    f_ok = [0, 1, 2, 3, 5, 6, 7]
    z = cube[:, 8]
    assert z.abs().max().item() > 0.0 and z.abs().max().item() <= 1e-6 * cube.abs().max().item()
    assert not torch.equal(z[0], z[1])                                              # keyed by the frame's content
    again = preprocessing.fft_chain(dev[[2, 0]])
    assert torch.equal(torch.view_as_real(again[0]), torch.view_as_real(cube[2]))   # a pure function of the frame
    s4 = ld[:, 4].reshape(4, 2, 4096, 8).double()
    assert s4.mean(2).abs().max().item() < 1e-4 and (s4.std(2, unbiased=True) - 1).abs().max().item() < 1e-4
    x = ld.reshape(4, 16, 64, 64, 8)
    want = (((x[..., 0] + x[..., 1]) + (x[..., 2] + x[..., 3])) + ((x[..., 4] + x[..., 5]) + (x[..., 6] + x[..., 7]))) * 0.125
    assert torch.equal(pl, want)                                                    # the fused elevation mean carries it too
    # opt-in: exactly zero
    cube0 = preprocessing.fft_chain(dev, zero_doppler="exact")
    ld0 = preprocessing.fft_chain_loader(dev, zero_doppler="exact")
    pl0 = preprocessing.fft_chain_loader_means(dev, zero_doppler="exact")
    assert cube0[:, 8].abs().max().item() == 0.0
    assert torch.isfinite(ld0).all() and torch.isfinite(pl0).all()
    assert ld0[:, 4].abs().max().item() == 0.0 and pl0[:, 8:10].abs().max().item() == 0.0
    assert torch.equal(torch.view_as_real(cube0[:, keep]), torch.view_as_real(cube[:, keep]))
    assert torch.equal(ld0[:, f_ok], ld[:, f_ok])
    # opt-in: the range-first order (same transform, other order of operations)
    old = preprocessing.fft_chain(dev, zero_doppler="range_first")
    old_ld = preprocessing.fft_chain_loader(dev, zero_doppler="range_first")
    rel = ((cube[:, keep] - old[:, keep]).abs().pow(2).sum().sqrt() / old[:, keep].abs().pow(2).sum().sqrt()).item()
    assert rel <= 1e-6, rel
    assert 0.0 < old[:, 8].abs().max().item() <= 1e-5 * old.abs().max().item()
    assert (ld[:, f_ok] - old_ld[:, f_ok]).abs().max().item() <= 1e-4
    o4 = old_ld[:, 4].reshape(4, 2, 4096, 8).double()
    assert torch.isfinite(old_ld).all() and (o4.std(2, unbiased=True) - 1).abs().max().item() < 1e-3
    with pytest.raises(ValueError):
        preprocessing.fft_chain(dev, zero_doppler="noise")
