"""Integer-exact, counter-based synthetic Neuropixels-like int16 generator.

Every sample x(t, c) is a pure function of (t, c, n_channels, seed): a 64-tap exponential FIR over
Irwin-Hall(4) noise drawn from a splitmix64 finaliser.  numpy here and the HIP kernel
``mts_synth`` (csrc/transform.hip) produce identical bytes, so any shard can make its own
samples on device.  Known answers (checked in tests/test_synth.py): nc=64, seed=0, t in [0, 60000):
raw sha1 d84c76a9ea6d9614b91cd7d5597dcb58d0760aa5, x[0:2, 0:4] = [[10, 9, -11, -8], [15, -1, -14, -3]].
"""
import numpy as np

TAPS = np.array([256, 230, 207, 187, 168, 151, 136, 122, 110, 99, 89, 80, 72, 65, 59, 53, 47, 43, 38,
                 35, 31, 28, 25, 23, 20, 18, 17, 15, 13, 12, 11, 10, 9, 8, 7, 6, 6, 5, 5, 4, 4, 3, 3, 3,
                 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0], dtype=np.int64)
N_TAPS = 64
SEED_MUL = 0xD1B54A32D192ED03
_M64 = (1 << 64) - 1


def _noise(t0, t1, nc, seed):
    """s(t, c) for t in [t0, t1): Irwin-Hall(4) over the four 16-bit lanes of splitmix64(idx)."""
    t = np.arange(t0, t1, dtype=np.int64)[:, None]
    c = np.arange(nc, dtype=np.int64)[None, :]
    with np.errstate(over='ignore'):
        idx = (t * np.int64(nc) + c).astype(np.uint64)
        idx ^= np.uint64((seed * SEED_MUL) & _M64)
        z = idx + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    m = np.uint64(0xFFFF)
    s = (z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + (z >> np.uint64(48))
    return s.astype(np.int64) - 131070


def synth_int16(t0, t1, n_channels, seed=0, block=32768):
    """Samples t in [t0, t1) as an (t1-t0, n_channels) int16 C-contiguous array."""
    out = np.empty((t1 - t0, n_channels), dtype=np.int16)
    for a in range(t0, t1, block):
        b = min(t1, a + block)
        s = _noise(a - (N_TAPS - 1), b, n_channels, seed)       # rows a-63 .. b-1
        y = np.zeros((b - a, n_channels), dtype=np.int64)
        for k in range(N_TAPS):
            if TAPS[k]:
                y += TAPS[k] * s[N_TAPS - 1 - k: N_TAPS - 1 - k + (b - a)]
        out[a - t0:b - t0] = ((4 * y) >> 23).astype(np.int16)
    return out
