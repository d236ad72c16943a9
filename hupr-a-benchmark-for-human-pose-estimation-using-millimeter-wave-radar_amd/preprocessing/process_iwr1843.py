"""Host-side mirror of the reference's preprocessing operator.

``RadarObject().generateHeatmap(frame)`` keeps the reference contract
(preprocessing/process_iwr1843.py:106,173): ``frame`` complex ndarray (4,192,256) ->
complex ndarray (16,64,64,8); the arithmetic runs in the gfx950 FFT-chain kernels
(csrc/fft_chain.hip) in complex64 and is widened to complex128 on return.

The batched device entry points (`fft_chain`, `fft_chain_loader`) are what the training
loader uses: int16 I/Q cubes resident in HBM in, complex64 cubes or normalised fp32 network
input out.
"""
import os

import numpy as np
import torch

from .. import runtime as rt

NUM_RX, NUM_CHIRP, NUM_SAMPLE = 4, 192, 256


def _workspace(n_sf, device):
    nbytes = rt.lib().hupr_fft_chain_ws_bytes(n_sf)
    return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device), nbytes


def _check_adc(adc_iq):
    if adc_iq.dtype != torch.int16 or tuple(adc_iq.shape[1:]) != (NUM_RX, NUM_CHIRP, NUM_SAMPLE, 2):
        raise ValueError("adc_iq must be int16 (n,4,192,256,2), got %s %r" % (adc_iq.dtype, tuple(adc_iq.shape)))


HANN_RANGE, HANN_DOPPLER, MAGNITUDE, ZERO_DOPPLER_EXACT, RANGE_FIRST = 1, 2, 4, 8, 16      # include/hupr.h HUPR_FFT_*

# What the zero-Doppler bin (Doppler index 8, loader slot f = 4) holds — include/hupr.h, "The zero-Doppler bin":
#   "dither"      (default) a frame-keyed stand-in for the reference's fp64 rounding residue, normalised like every other bin,
#                 so an unmodified reference Normalize / a reference-trained checkpoint sees what it was trained on;
#   "exact"       exactly zero (round 3's chain; the loader emits zeros in slot f = 4);
#   "range_first" the rounds-1/2 kernel order, whose own fp32 rounding residue fills the bin (slower).
# HUPR_FFT_ZERO_DOPPLER selects the process default; tools.Runner records the mode in `preprocess.json` next to its checkpoints and
# warns when weights trained under one convention are loaded under another.
ZERO_DOPPLER_MODES = {"dither": 0, "exact": ZERO_DOPPLER_EXACT, "range_first": RANGE_FIRST}
ZERO_DOPPLER = os.environ.get("HUPR_FFT_ZERO_DOPPLER", "dither")
if ZERO_DOPPLER not in ZERO_DOPPLER_MODES:
    raise ValueError("HUPR_FFT_ZERO_DOPPLER must be one of %s, got %r" % (sorted(ZERO_DOPPLER_MODES), ZERO_DOPPLER))


def _flags(window, magnitude, zero_doppler=None):
    """window: None / False (the reference: rectangular), "hann" / True (range + Doppler), "range", "doppler";
    zero_doppler: None (the process default ZERO_DOPPLER) or a key of ZERO_DOPPLER_MODES."""
    table = {None: 0, False: 0, "none": 0, True: 3, "hann": 3, "range": 1, "doppler": 2}
    if window not in table:
        raise ValueError("window must be one of None, 'hann', 'range', 'doppler'; got %r" % (window,))
    zd = ZERO_DOPPLER if zero_doppler is None else zero_doppler
    if zd not in ZERO_DOPPLER_MODES:
        raise ValueError("zero_doppler must be one of %s, got %r" % (sorted(ZERO_DOPPLER_MODES), zd))
    return table[window] | (MAGNITUDE if magnitude else 0) | ZERO_DOPPLER_MODES[zd]


def fft_chain(adc_iq, ws=None, window=None, magnitude=False, zero_doppler=None):
    """adc_iq: int16 GPU tensor (n,4,192,256,2) -> complex64 GPU tensor (n,16,64,64,8).
    Opt-in (defaults = the reference: no window, complex output): ``window="hann"`` applies np.hanning windows over the
    range samples and the chirp loops, ``magnitude=True`` returns |X| as fp32.  ``zero_doppler``: see ZERO_DOPPLER_MODES."""
    _check_adc(adc_iq)
    n = adc_iq.shape[0]
    flags = _flags(window, magnitude, zero_doppler)
    out = torch.empty((n, 16, 64, 64, 8), dtype=torch.float32 if magnitude else torch.complex64, device=adc_iq.device)
    if ws is None:
        ws, nbytes = _workspace(n, adc_iq.device)
    else:
        nbytes = ws.numel() * ws.element_size()
    if flags == 0:
        rt.check(rt.lib().hupr_fft_chain_c64(rt.ptr(adc_iq), n, rt.ptr(out), rt.ptr(ws), nbytes, rt.stream()))
    else:
        rt.check(rt.lib().hupr_fft_chain_opts(rt.ptr(adc_iq), n, rt.ptr(out), flags, 0, rt.ptr(ws), nbytes, rt.stream()))
    return out


def fft_chain_loader(adc_iq, ws=None, out=None, window=None, zero_doppler=None):
    """adc_iq: int16 GPU tensor (n,4,192,256,2) -> fp32 GPU tensor (n, 8, 2, 64, 64, 8):
    Doppler bins 4..11, re/im split, per-elevation Normalize (datasets glue fused in).  ``window`` / ``zero_doppler``: see
    fft_chain."""
    _check_adc(adc_iq)
    n = adc_iq.shape[0]
    if out is None:
        out = torch.empty((n, 8, 2, 64, 64, 8), dtype=torch.float32, device=adc_iq.device)
    if ws is None:
        ws, nbytes = _workspace(n, adc_iq.device)
    else:
        nbytes = ws.numel() * ws.element_size()
    flags = _flags(window, False, zero_doppler)
    if flags == 0:
        rt.check(rt.lib().hupr_fft_chain_loader_f32(rt.ptr(adc_iq), n, rt.ptr(out), rt.ptr(ws), nbytes, rt.stream()))
    else:
        rt.check(rt.lib().hupr_fft_chain_opts(rt.ptr(adc_iq), n, rt.ptr(out), flags, 1, rt.ptr(ws), nbytes, rt.stream()))
    return out


def fft_chain_loader_means(adc_iq, ws=None, zero_doppler=None):
    """adc_iq: int16 GPU tensor (n,4,192,256,2) -> fp32 GPU tensor (n, 16, 64, 64): the loader tensor of ``fft_chain_loader``
    averaged over its elevation axis, plane index 2 f + c — what ``HuPRNet.forward`` computes first (models/networks.py:26-27).
    The fused training loader hands these planes to the model instead of the 8x larger (n,8,2,64,64,8) tensor."""
    _check_adc(adc_iq)
    n = adc_iq.shape[0]
    out = torch.empty((n, 16, 64, 64), dtype=torch.float32, device=adc_iq.device)
    if ws is None:
        ws, nbytes = _workspace(n, adc_iq.device)
    else:
        nbytes = ws.numel() * ws.element_size()
    flags = _flags(None, False, zero_doppler)
    if flags == 0:
        rt.check(rt.lib().hupr_fft_chain_loader_means_f32(rt.ptr(adc_iq), n, rt.ptr(out), rt.ptr(ws), nbytes, rt.stream()))
    else:
        rt.check(rt.lib().hupr_fft_chain_opts(rt.ptr(adc_iq), n, rt.ptr(out), flags, 2, rt.ptr(ws), nbytes, rt.stream()))
    return out


def loader_normalize(cube):
    """cube: complex64 GPU tensor (n,16,64,64,8) -> fp32 (n,8,2,64,64,8) (dataset.py:144-150)."""
    if cube.dtype != torch.complex64 or tuple(cube.shape[1:]) != (16, 64, 64, 8):
        raise ValueError("cube must be complex64 (n,16,64,64,8)")
    n = cube.shape[0]
    out = torch.empty((n, 8, 2, 64, 64, 8), dtype=torch.float32, device=cube.device)
    rt.check(rt.lib().hupr_loader_normalize_c64(rt.ptr(cube), n, rt.ptr(out), rt.stream()))
    return out


def dca1000_frames(raw):
    """raw: int16 GPU tensor (whole adc_data.bin stream) -> int16 GPU tensor (n_frames,4,192,256,2), the FFT-chain
    input layout (reference getadcDataFromDCA1000 :54-83 + the per-frame slicing of :190-191)."""
    if raw.dtype != torch.int16 or raw.dim() != 1:
        raise ValueError("raw must be a flat int16 tensor")
    per_frame = NUM_RX * NUM_CHIRP * NUM_SAMPLE * 2
    n_frames = raw.numel() // per_frame
    out = torch.empty((n_frames, NUM_RX, NUM_CHIRP, NUM_SAMPLE, 2), dtype=torch.int16, device=raw.device)
    rt.check(rt.lib().hupr_dca1000_deinterleave(rt.ptr(raw), rt.ptr(out), n_frames, rt.stream()))
    return out


class RadarObject:
    """Same constants and operator surface as the reference class (process_iwr1843.py:8-34).  ``numGroup`` / ``root`` /
    ``saveRoot`` / ``rawRoot`` parametrise the directory lists the reference hard-codes in ``initialize`` (:36-46); the
    defaults reproduce its paths."""

    def __init__(self, device="cuda", numGroup=276, root="HuPR", saveRoot="HuPR", rawRoot="raw_data/iwr1843", dataRoot="../data"):
        self.root, self.saveRoot, self.sensorType = root, saveRoot, "iwr1843"
        self.radarDataFileNameGroup = [[os.path.join(rawRoot, root, "single_%d" % i, "hori"), os.path.join(rawRoot, root, "single_%d" % i, "vert")]
                                       for i in range(1, numGroup + 1)]
        self.saveDirNameGroup = [os.path.join(dataRoot, saveRoot, "single_%d" % i) for i in range(1, numGroup + 1)]
        self.numADCSamples = 256
        self.adcRatio = 4
        self.numAngleBins = self.numADCSamples // self.adcRatio
        self.numEleBins = 8
        self.numRX = 4
        self.numLanes = 2
        self.framePerSecond = 10
        self.duration = 60
        self.numFrame = self.framePerSecond * self.duration
        self.numChirp = 64 * 3
        self.idxProcChirp = 64
        self.numGroupChirp = 4
        self.numKeypoints = 14
        self.device = device

    def getadcDataFromDCA1000(self, fileName):
        """<fileName>/adc_data.bin -> complex128 ndarray (4, n_chirps, 256), like the reference; the de-interleave
        runs on the GPU.  Use ``dca1000_frames`` directly to keep the cube on the device for ``fft_chain``."""
        raw = np.fromfile(os.path.join(fileName, "adc_data.bin"), dtype=np.int16)
        fr = dca1000_frames(torch.from_numpy(raw).to(self.device)).cpu().numpy()      # (F,4,192,256,2)
        z = fr[..., 0].astype(np.float64) + 1j * fr[..., 1].astype(np.float64)
        return np.ascontiguousarray(z.transpose(1, 0, 2, 3).reshape(self.numRX, -1, self.numADCSamples))

    def saveRadarData(self, matrix, dirName, idxFrame):
        """``<dirName>/<idxFrame:09d>.npy`` (reference :180-182)."""
        np.save(os.path.join(dirName, "%09d.npy" % idxFrame), matrix)

    def processRadarDataHoriVert(self, frames_per_call=64):
        """The reference's offline driver (:184-196): every sequence's two captures -> one complex128 ``(16,64,64,8)`` cube per
        frame and sensor under ``<saveDir>/{hori,vert}/%09d.npy``.  Here a capture goes to the GPU once, is de-interleaved there
        and transformed ``frames_per_call`` frames at a time (the reference parses 115 200 chirps and runs ~200 000 tiny FFT calls
        per frame in Python).  Training does not need these files: ``datasets.HuPRRawADC`` feeds the network straight from
        the captures."""
        for names, save_dir in zip(self.radarDataFileNameGroup, self.saveDirNameGroup):
            for sensor, name in zip(("hori", "vert"), names):
                out_dir = os.path.join(save_dir, sensor)
                os.makedirs(out_dir, exist_ok=True)
                raw = np.fromfile(os.path.join(name, "adc_data.bin"), dtype=np.int16)
                frames = dca1000_frames(torch.from_numpy(raw).to(self.device))
                n = min(frames.shape[0], self.numFrame)
                for f0 in range(0, n, frames_per_call):
                    cubes = fft_chain(frames[f0:min(n, f0 + frames_per_call)]).cpu().numpy().astype(np.complex128)
                    for k, cube in enumerate(cubes):
                        self.saveRadarData(cube, out_dir, f0 + k)
                print("%s, finished %d frames" % (name, n))

    def generateHeatmap(self, frame, window=None, magnitude=False):
        """frame: complex ndarray (4,192,256) -> complex128 ndarray (16,64,64,8) (float64 with ``magnitude``).
        ``window`` / ``magnitude`` are opt-in extras (see ``fft_chain``); the defaults are the reference's arithmetic."""
        frame = np.asarray(frame)
        if frame.shape != (self.numRX, self.numChirp, self.numADCSamples):
            raise ValueError("frame must be (4,192,256), got %r" % (frame.shape,))
        iq = np.stack([frame.real, frame.imag], axis=-1)
        if not np.all(np.abs(iq) <= 32767) or not np.array_equal(iq, np.rint(iq)):
            raise ValueError("generateHeatmap expects integer-valued 16-bit ADC samples")
        dev = torch.from_numpy(iq.astype(np.int16)[None]).to(self.device)
        out = fft_chain(dev, window=window, magnitude=magnitude)
        return out[0].cpu().numpy().astype(np.float64 if magnitude else np.complex128)
