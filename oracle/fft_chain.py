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
    M5 = angle_cube(az, el)
    # out[i, r, a, e] = M5[(3-e)%8, (31-a)%64, (56+i)%64, 94-r]   
    e_idx = (3 - np.arange(NUM_EL)) % NUM_EL
    a_idx = (31 - np.arange(NUM_AZ)) % NUM_AZ
    d_idx = (56 + np.arange(NUM_DOPPLER_KEEP)) % NUM_LOOPS
    r_idx = RANGE_HI - np.arange(64)
    out = M5[e_idx[None, None, None, :], a_idx[None, None, :, None],
             d_idx[:, None, None, None], r_idx[None, :, None, None]]
    out = np.ascontiguousarray(out)
    return np.abs(out) if magnitude else out


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
