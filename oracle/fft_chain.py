"""ORACLE (test infrastructure, not product): CPU restatement of the IWR1843
range -> Doppler -> elevation -> azimuth FFT chain in NumPy complex128.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product path never does.

Follows reference ``preprocessing/process_iwr1843.py``:
  * TDM-MIMO demux                 :113-120
  * static clutter removal         :122-128 (-> clutterRemoval :85-104)
  * range-Doppler fft2             :131-134
  * zero padding / stacking        :137-143
  * elevation then azimuth FFT     :144-151
  * range crop 94..31 (descending) :154,163
  * Doppler fftshift + keep 24..39 :164,167-171
  * az/el fftshift + flips         :48-52 (postProcessFFT3D)

Parity is pinned by golden vectors produced by running the *imported* reference
in the build container (tests/golden/make_golden.py -> tests/golden/fft_*.npz);
the reference itself ships no tests for this path.
"""
import numpy as np

NUM_RX = 4
NUM_CHIRP = 192          # 64 chirp loops x 3 TX
NUM_LOOPS = 64
NUM_SAMPLE = 256
NUM_AZ = 64
NUM_EL = 8
RANGE_HI, RANGE_LO = 94, 30   # kept range bins 94,93,...,31
NUM_DOPPLER_KEEP = 16


def demux(frame):
    """(4,192,256) complex -> azimuth array (8,64,256), elevation array (4,64,256).

    chirp % 3 == 0 -> TX1 -> azimuth rows 0..3; == 2 -> TX3 -> azimuth rows 4..7;
    == 1 -> TX2 -> the elevated row (reference :113-120).
    """
    frame = np.asarray(frame, dtype=np.complex128)
    az = np.concatenate([frame[:, 0::3, :], frame[:, 2::3, :]], axis=0)
    el = frame[:, 1::3, :].copy()
    return az, el


def range_doppler(az, el, window=0):
    """Clutter removal (mean over the 64 chirp loops) + unnormalised fft2 over (chirp, sample).

    window (opt-in, 0 = the reference, which applies none — :130-134 are bare fft2 calls): bit 0 multiplies the 256 range
    samples by np.hanning(256), bit 1 multiplies the mean-free chirp loops by np.hanning(64) — the build's definition of
    north_star's "Hanning windowing", pinned only against this restatement.
    """
    az = az - az.mean(axis=1, keepdims=True)
    el = el - el.mean(axis=1, keepdims=True)
    if window & 1:
        az, el = az * np.hanning(NUM_SAMPLE)[None, None, :], el * np.hanning(NUM_SAMPLE)[None, None, :]
    if window & 2:
        az, el = az * np.hanning(NUM_LOOPS)[None, :, None], el * np.hanning(NUM_LOOPS)[None, :, None]
    az = np.fft.fft2(az, axes=(1, 2))
    el = np.fft.fft2(el, axes=(1, 2))
    return az, el


def zero_doppler_dither(iq):
    """THE BUILD'S definition (not the reference's — pinned only against this restatement, like the opt-in windows): what the
    HIP chain puts where clutter removal cancels the signal.  The reference's zero-Doppler bin is the fp64 rounding residue of
    its fft2 (:122-134; ~2e-16 of the other bins, white over range and antenna, a pure function of the frame), which its
    Normalize (datasets/base.py:17-24) inflates to a unit-variance channel; pocketfft's residue cannot be reproduced by another
    implementation, so the chain substitutes a frame-keyed dither of the same statistics (csrc/fft_chain.hip,
    ``zero_doppler_dither``): per (virtual antenna v, ADC sample s) two 16-bit integers hashed from the exact chirp sums
    T[v, s] = sum over the 64 chirp loops of the int16 samples, scaled by 2^-53.

    iq: int16 (4, 192, 256, 2)  ->  complex128 (12, 256): rows 0..7 azimuth antennas (TX1 rx0..3, TX3 rx0..3), 8..11 elevated.
    """
    iq = np.asarray(iq)
    assert iq.dtype == np.int16 and iq.shape == (NUM_RX, NUM_CHIRP, NUM_SAMPLE, 2)
    t = np.concatenate([iq[:, 0::3], iq[:, 2::3], iq[:, 1::3]], axis=0).astype(np.int64).sum(axis=1)      # (12, 256, 2)
    m = np.uint64(0xFFFFFFFF)
    pos = (np.arange(12, dtype=np.uint64)[:, None] * np.uint64(256) + np.arange(256, dtype=np.uint64)[None, :])
    u = lambda x: x.astype(np.int64).astype(np.uint64) & m           # two's-complement uint32 image of an int
    h = ((u(t[..., 0]) * np.uint64(0x9E3779B1)) & m) ^ ((u(t[..., 1]) * np.uint64(0x85EBCA77)) & m) ^ ((pos * np.uint64(0xC2B2AE3D)) & m)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & m
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & m
    h ^= h >> np.uint64(16)
    re = (h & np.uint64(0xFFFF)).astype(np.uint16).view(np.int16).astype(np.float64)
    im = (h >> np.uint64(16)).astype(np.uint16).view(np.int16).astype(np.float64)
    return (re + 1j * im) * 2.0 ** -53


def generate_heatmap_dithered(iq):
    """``generate_heatmap`` of the int16 cube ``iq`` (4,192,256,2) with the zero-Doppler row of the range-Doppler map replaced by
    the range FFT of ``zero_doppler_dither`` — what the HIP chain computes by default.  Every other bin is the reference's."""
    frame = np.asarray(iq)[..., 0].astype(np.float64) + 1j * np.asarray(iq)[..., 1].astype(np.float64)
    az, el = demux(frame)
    az, el = range_doppler(az, el)
    d = np.fft.fft(zero_doppler_dither(iq), axis=1)
    az[:, 0, :], el[:, 0, :] = d[:8], d[8:]
    return _index_map(angle_cube(az, el))


def angle_cube(az, el):
    """Zero-padded cube M[e, a, d, s] after elevation (rows a=2..5 only) and azimuth FFTs."""
    M = np.zeros((NUM_EL, NUM_AZ, NUM_LOOPS, NUM_SAMPLE), dtype=np.complex128)
    M[0, 0:8] = az
    M[1, 2:6] = el
    M[:, 2:6] = np.fft.fft(M[:, 2:6], axis=0)      # elevation first (order matters)
    M = np.fft.fft(M, axis=1)                        # then azimuth
    return M


def generate_heatmap(frame, window=0, magnitude=False):
    """Closed form of ``RadarObject.generateHeatmap`` (reference :106-173).

    frame: complex (4, 192, 256)  ->  complex128 (16 doppler, 64 range, 64 az, 8 el).
    window / magnitude: opt-in extras (see range_doppler; magnitude = np.abs of the result); defaults = the reference.
    """
    az, el = demux(frame)
    az, el = range_doppler(az, el, window)
    out = _index_map(angle_cube(az, el))
    return np.abs(out) if magnitude else out


def _index_map(M5):
    """out[i, r, a, e] = M5[(3-e)%8, (31-a)%64, (56+i)%64, 94-r]   (reference :154-171, :48-52)"""
    e_idx = (3 - np.arange(NUM_EL)) % NUM_EL
    a_idx = (31 - np.arange(NUM_AZ)) % NUM_AZ
    d_idx = (56 + np.arange(NUM_DOPPLER_KEEP)) % NUM_LOOPS
    r_idx = RANGE_HI - np.arange(64)
    out = M5[e_idx[None, None, None, :], a_idx[None, None, :, None],
             d_idx[:, None, None, None], r_idx[None, :, None, None]]
    return np.ascontiguousarray(out)


def generate_heatmap_percell(frame):
    """Loop-shaped variant: one tiny FFT call per (chirp, sample) cell like the reference's
    hot loop A (:144-151) and per-cell fftshift gather like hot loop B (:160-164).

    Same numbers as :func:`generate_heatmap`; exists only so ``bench.py`` can time what the
    reference *literally* executes next to the vectorised form.
    """
    az, el = demux(frame)
    az, el = range_doppler(az, el)
    M = np.zeros((NUM_EL, NUM_AZ, NUM_LOOPS, NUM_SAMPLE), dtype=np.complex128)
    M[0, 0:8] = az
    M[1, 2:6] = el
    for d in range(NUM_LOOPS):
        for s in range(NUM_SAMPLE):
            for a in (2, 3, 4, 5):
                M[:, a, d, s] = np.fft.fft(M[:, a, d, s])
            for e in range(NUM_EL):
                M[e, :, d, s] = np.fft.fft(M[e, :, d, s])
    tmp = np.zeros((NUM_LOOPS, 64, NUM_AZ, NUM_EL), dtype=np.complex128)
    for e in range(NUM_EL):
        for a in range(NUM_AZ):
            for r in range(64):
                tmp[:, r, a, e] = np.fft.fftshift(M[e, a, :, RANGE_HI - r])
    out = np.zeros((NUM_DOPPLER_KEEP, 64, NUM_AZ, NUM_EL), dtype=np.complex128)
    for i, d in enumerate(range(NUM_LOOPS // 2 - 8, NUM_LOOPS // 2 + 8)):
        cell = np.fft.fftshift(tmp[d], axes=(1, 2))     # (r, a, e): shift az & el
        out[i] = cell[:, ::-1, ::-1]
    return out
