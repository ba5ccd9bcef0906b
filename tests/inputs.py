"""Seeded byte-stream cases shared by the oracle and the GPU parity tests."""
import numpy as np

from mtscomp_amd.synth import synth_int16


def _rng(seed):
    return np.random.RandomState(seed)


def ar1_stream(nt, nc, seed=0):
    """The transformed (delta + channel-major) stream of the synthetic recording."""
    x = synth_int16(0, nt, nc, seed)
    d = np.diff(x, axis=0)
    d = np.concatenate((x[:1], d), axis=0)
    return d.tobytes(order='F')


def textlike(n, seed):
    r = _rng(seed)
    words = [bytes(r.randint(97, 123, size=r.randint(2, 9)).astype(np.uint8)) for _ in range(200)]
    out = bytearray()
    while len(out) < n:
        out += words[r.randint(0, len(words))] + b' '
    return bytes(out[:n])


def repeats(n, seed):
    """Long matches, some at distances around MAX_DIST, runs (dist 1) and lazy-match bait."""
    r = _rng(seed)
    base = r.randint(0, 256, size=40000).astype(np.uint8).tobytes()
    out = bytearray(base[:5000])
    while len(out) < n:
        k = r.randint(0, 6)
        if k == 0:
            out += bytes([r.randint(0, 256)]) * r.randint(1, 700)
        elif k == 1:
            d = r.randint(32400, 32600)
            if d < len(out):
                L = r.randint(3, 400)
                s = len(out) - d
                out += out[s:s + L]
        elif k == 2:
            d = r.randint(1, min(len(out), 5000))
            L = r.randint(3, 300)
            for _ in range(L):
                out.append(out[-d])
        elif k == 3:
            out += r.randint(0, 4, size=r.randint(1, 300)).astype(np.uint8).tobytes()
        else:
            s = r.randint(0, len(base) - 600)
            out += base[s:s + r.randint(1, 600)]
    return bytes(out[:n])


def farcopies(n, seed):
    """Nothing but short copies from 16-32 KiB back: every token costs ~20 bits (13 of them distance extra
    bits), more than the 16 bits per token the block-pack kernel's LDS image is sized for."""
    r = _rng(seed)
    out = bytearray(r.randint(0, 256, size=33000).astype(np.uint8).tobytes())
    while len(out) < n:
        d = r.randint(16385, 32000)
        s = len(out) - d
        out += out[s:s + r.randint(4, 40)]
    return bytes(out[:n])


def skewlen(n, seed, ratio=1.65, nsym=22, nlen=8):
    """Tokens drawn i.i.d. from a geometric distribution (ratio just above the golden one): the rare ones are 14 literal
    values, the frequent ones copies of length 4..11 from far back in random data.  The literal/length Huffman tree of
    most blocks comes out 16-17 high, so zlib's gen_bitlen overflow repair (lengths clamped to 15) runs."""
    r = _rng(seed)
    p = np.array([ratio ** i for i in range(nsym)])
    p /= p.sum()
    out = bytearray(r.randint(0, 256, size=40000).astype(np.uint8).tobytes())
    while len(out) < n:
        for u in r.choice(nsym, size=4096, p=p):
            if u < nsym - nlen:
                out.append(int(u))
            else:
                s = len(out) - r.randint(5000, 30000)
                out += out[s:s + 4 + (u - (nsym - nlen))]
    return bytes(out[:n])


def cases_small():
    r = _rng(1)
    c = {
        'empty': b'',
        'one': b'a',
        'two': b'ab',
        'three': b'abc',
        'aaa': b'a' * 10,
        'zeros_1k': bytes(1000),
        'zeros_70k': bytes(70000),
        'zeros_64k': bytes(65536),             # one block whose last pass-B round ends with the block (stale LDS staging once)
        'zeros_999468': bytes(999468),
        'rand_300': r.randint(0, 256, size=300).astype(np.uint8).tobytes(),
        'rand_70k': r.randint(0, 256, size=70000).astype(np.uint8).tobytes(),
        'rand4_50k': r.randint(0, 4, size=50000).astype(np.uint8).tobytes(),
        'text_100k': textlike(100000, 2),
        'repeats_200k': repeats(200000, 3),
        'farcopies_900k': farcopies(900000, 5),
        'skewlen_450k': skewlen(450000, 6),
        'first50': (lambda b: b + b)(r.randint(0, 256, size=50).astype(np.uint8).tobytes()),
        'ar1_8ch': ar1_stream(3000, 8),
        'ar1_64ch_4k': ar1_stream(4000, 64),
        'ramp': (np.arange(40000) % 251).astype(np.uint8).tobytes(),
        'wrap16': np.tile(np.array([32767, -32767], dtype=np.int16), 20000).tobytes(),
    }
    return c
