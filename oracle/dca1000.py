"""ORACLE (test infrastructure, not product): DCA1000 raw capture parsing, vectorised NumPy restatement of
``RadarObject.getadcDataFromDCA1000`` (reference preprocessing/process_iwr1843.py:54-83).

File = int16 stream in groups of four [I(2k), I(2k+1), Q(2k), Q(2k+1)] (2 LVDS lanes, complex).  Complex sample n of
the stream sits at I = raw[4*(n//2) + n%2], Q = raw[4*(n//2) + 2 + n%2].  Per chirp the stream holds
[rx0 x 256][rx1 x 256][rx2 x 256][rx3 x 256]; result shape (4 rx, total chirps, 256 samples) complex128.
Pinned against the imported reference in tests/test_oracle_golden.py (skipped when /root/reference is absent)
and by tests/golden/dca1000_small.npz.
"""
import numpy as np

NUM_RX, NUM_SAMPLES = 4, 256


def parse_dca1000(raw):
    """raw: int16 ndarray (whole adc_data.bin) -> complex128 (4, n_chirps, 256)."""
    raw = np.asarray(raw, dtype=np.int16)
    assert raw.size % (4 * NUM_RX * NUM_SAMPLES // 2 * 2) == 0 or raw.size % 4 == 0
    g = raw.reshape(-1, 4).astype(np.float64)
    re = g[:, 0:2].reshape(-1)          # I(2k), I(2k+1), ...
    im = g[:, 2:4].reshape(-1)
    z = re + 1j * im                    # complex stream
    n_chirps = z.size // (NUM_RX * NUM_SAMPLES)
    z = z[:n_chirps * NUM_RX * NUM_SAMPLES].reshape(n_chirps, NUM_RX, NUM_SAMPLES)
    return np.ascontiguousarray(z.transpose(1, 0, 2))


def frames_int16(raw, chirps_per_frame=192):
    """raw int16 stream -> (n_frames, 4, 192, 256, 2) int16 I/Q: the device layout of the FFT chain."""
    z = parse_dca1000(raw)
    n_frames = z.shape[1] // chirps_per_frame
    z = z[:, :n_frames * chirps_per_frame].reshape(NUM_RX, n_frames, chirps_per_frame, NUM_SAMPLES)
    z = z.transpose(1, 0, 2, 3)
    return np.stack([z.real, z.imag], axis=-1).astype(np.int16)
