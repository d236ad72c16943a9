"""ORACLE (test infrastructure, not product): the loader glue between the FFT chain and
the network — Doppler-bin selection, re/im split and per-elevation ``Normalize``.

Follows reference ``datasets/dataset.py:144-150`` (keep Doppler indices
numChirps//2 - numFrames//2 .. +numFrames//2 = 4..11, real and imaginary parts as separate
slots) and ``datasets/base.py:13-24`` (Normalize: per channel  x-min, /max, then
(x-mean)/std with the *unbiased* std of torch.std_mean; channel = elevation bin because
ToTensor turns the (64,64,8) HWC slice into CHW).

Parity pinned by tests/golden/loader_*.npz generated from the imported reference.
"""
import numpy as np

NUM_CHIRPS = 16
NUM_FRAMES = 8


def normalize_hwc(x):
    """x: float64 (R, A, E) -> float64 (R, A, E), statistics per E channel over the (R,A) plane."""
    x = np.asarray(x, dtype=np.float64)
    c = x.shape[2]
    flat = x.reshape(-1, c)
    z = flat - flat.min(axis=0, keepdims=True)
    z = z / z.max(axis=0, keepdims=True)
    mean = z.mean(axis=0, keepdims=True)
    std = z.std(axis=0, ddof=1, keepdims=True)
    return ((z - mean) / std).reshape(x.shape)


def loader_transform(cube):
    """cube: complex (16, R, A, E) (one sensor-frame from the FFT chain)
    -> float32 (F=8, 2, R, A, E): slot [f,0] = normalised real part of Doppler bin 4+f,
    slot [f,1] = normalised imaginary part (stored into a float32 tensor, dataset.py:129-130)."""
    lo = NUM_CHIRPS // 2 - NUM_FRAMES // 2
    out = np.empty((NUM_FRAMES, 2) + cube.shape[1:], dtype=np.float32)
    for f in range(NUM_FRAMES):
        out[f, 0] = normalize_hwc(cube[lo + f].real)
        out[f, 1] = normalize_hwc(cube[lo + f].imag)
    return out
