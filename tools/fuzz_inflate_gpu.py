"""Randomised inflate parity on a GPU box: streams from stdlib zlib at random levels / strategies / flush patterns over
random structured data, decoded through mts_debug_inflate; then the same streams with a random bit flipped: the verdict
(ok / corrupt / wrong size) and, when ok, the bytes must be zlib's.

    python tools/fuzz_inflate_gpu.py [seed] [seconds]
"""
import os
import sys
import time
import zlib
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mtscomp_amd import hip  # noqa: E402


def data_of(r):
    n = int(r.choice([0, 1, 5, 300, 5000, 70000, 300000, 1500000]))
    n = int(n * r.uniform(0.5, 1.5))
    k = r.randint(0, 7)
    if k == 0:
        return bytes(n)
    if k == 1:
        return r.randint(0, 256, size=n).astype(np.uint8).tobytes()
    if k == 2:
        return r.randint(0, 4, size=n).astype(np.uint8).tobytes()
    if k == 3:
        per = int(r.randint(1, 300))
        return (r.randint(0, 256, size=per).astype(np.uint8).tobytes() * (n // per + 1))[:n]
    if k == 4:
        words = [bytes(r.randint(97, 123, size=r.randint(2, 9)).astype(np.uint8)) for _ in range(100)]
        out = bytearray()
        while len(out) < n:
            out += words[r.randint(0, 100)] + b' '
        return bytes(out[:n])
    if k == 5:
        return np.cumsum(r.randint(-2, 3, size=n // 2 + 1)).astype(np.int16).tobytes()[:n]
    out = bytearray()
    while len(out) < n:
        m = int(r.randint(1, 5000))
        out += bytes([r.randint(0, 256)]) * m if r.randint(0, 2) else r.randint(0, 256, size=m).astype(np.uint8).tobytes()
    return bytes(out[:n])


def encode(r, data):
    level = int(r.randint(0, 10))
    strategy = int(r.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    co = zlib.compressobj(level, zlib.DEFLATED, 15, int(r.randint(1, 10)), strategy)
    if r.randint(0, 3) == 0 and len(data) > 10:
        step = int(r.randint(1, len(data)))
        return b''.join(co.compress(data[i:i + step]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(data), step)) + co.flush()
    return co.compress(data) + co.flush()


def verdict(z, n):
    try:
        out = zlib.decompress(z)
    except zlib.error:
        return -1, None
    return (0, out) if len(out) == n else (-2, None)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.
    r = np.random.RandomState(seed)
    t0 = time.time()
    n_ok = n_flip = bad = 0
    while time.time() - t0 < budget:
        data = data_of(r)
        z = encode(r, data)
        if os.environ.get('FUZZ_TRACE'):
            print('stream of %d bytes for %d' % (len(z), len(data)), flush=True)
        st, out = hip.debug_inflate(z, len(data))
        n_ok += 1
        if st != 0 or out != data:
            bad += 1
            print('MISMATCH clean stream: n %d z %d status %d head %s' % (len(data), len(z), st, z[:8].hex()))
        for _ in range(3):
            b = bytearray(z)
            i = int(r.randint(0, len(b)))
            b[i] ^= 1 << int(r.randint(0, 8))
            want, wout = verdict(bytes(b), len(data))
            t1 = time.time()
            st, out = hip.debug_inflate(bytes(b), len(data))
            # (a stream the fast path declines goes block after block through k_inf_wave at ~2000 blocks/s: with memLevel 1 a
            #  block is 127 tokens, a MB of compressed literals 16 000 blocks -- seed 31 has such a stream, 8.9 s for 1.09 MB)
            if time.time() - t1 > 2.0 + 10.0 * len(z) / 1e6:
                bad += 1
                print('SLOW damaged stream (%.1f s): byte %d of %d, zlib says %d' % (time.time() - t1, i, len(z), want))
            n_flip += 1
            same = st == want                     # (the class too: zlib's data check comes before the size, mtscomp.py:618-628)
            if not same or (want == 0 and out != wout):
                bad += 1
                print('MISMATCH flipped bit %d of byte %d: n %d z %d zlib says %d, device %d' % (0, i, len(data), len(z), want, st))
    print('inflate fuzz seed %d: %d streams, %d damaged, %d mismatches' % (seed, n_ok, n_flip, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
