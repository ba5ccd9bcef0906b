"""Randomised end-to-end parity on a GPU box: random dtypes / shapes / flags / contents through the C ABI's host entry
points, every chunk compared byte for byte with the reference's statement sequence on numpy + stdlib zlib (oracle.ref_*:
compressed bytes, status, decoded bytes).  Found the pass-B staging bug on run-length streams.

    python tools/fuzz_gpu.py [seed] [seconds]          (FUZZ_LEVELS=1: zlib levels 4..9 instead of 6 only; FUZZ_LEVELS=2: levels 1..9,
                                                        i.e. deflate_fast too; MTS_FAST_LIST_BYTES=1 makes its candidate lists 256 positions at a time)
"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mtscomp_amd import hip  # noqa: E402
from oracle import oracle as O  # noqa: E402

DTYPES = ['uint8', 'int8', 'int16', 'uint16', 'int32', 'int64', 'float32', 'float64']


def content(r, kind, nt, nc):
    if kind == 0:
        return np.cumsum(r.randint(-3, 4, size=(nt, nc)), axis=0)                    # random walk
    if kind == 1:
        return np.zeros((nt, nc))
    if kind == 2:
        return r.randint(-100000, 100000, size=(nt, nc))                              # incompressible
    if kind == 3:
        return np.tile(r.randint(-50, 50, size=(max(1, nt // 7 + 1), nc)), (8, 1))[:nt]   # far repeats
    if kind == 4:
        return np.sin(np.arange(nt)[:, None] / 9.) * 1000 + r.randn(nt, nc) * r.choice([0.01, 1, 100])
    if kind == 5:                                                                     # pieces: zeros / constant / noise / ramp
        x = np.zeros((nt, nc))
        t = 0
        while t < nt:
            n = int(r.randint(1, max(2, nt // 3)))
            k = r.randint(0, 4)
            if k == 1:
                x[t:t + n] = r.randint(-9, 9)
            elif k == 2:
                x[t:t + n] = r.randint(-2000, 2000, size=x[t:t + n].shape)
            elif k == 3:
                x[t:t + n] = np.arange(t, t + n)[:x[t:t + n].shape[0], None] * r.randint(1, 5)
            t += n
        return x
    if kind == 6:                                                                     # short periods along time
        per = int(r.randint(1, 40))
        return np.tile(r.randint(-5, 5, size=(per, nc)), (nt // per + 1, 1))[:nt].cumsum(axis=0) % 7
    if kind == 7:
        return r.randint(0, 4, size=(nt, nc))                                         # four symbols
    if kind == 8:                                                                     # repeats around the 32 KiB window edge
        per = max(1, int((32768 + r.randint(-400, 400)) // r.choice([1, 2, 4, 8])))
        x = np.tile(r.randint(-30000, 30000, size=(per, nc)), (nt // per + 1, 1))[:nt]
        idx = r.randint(0, nt, size=max(1, nt // 500))
        x[idx] += r.randint(-2, 3, size=(len(idx), nc))
        return x
    x = np.zeros((nt, nc))                                                            # sparse spikes
    idx = r.randint(0, nt, size=max(1, nt // 50))
    x[idx, r.randint(0, nc, size=len(idx))] = r.randint(-3000, 3000, size=len(idx))
    return x


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.
    r = np.random.RandomState(seed)
    t0 = time.time()
    n = bad = 0
    while time.time() - t0 < budget:
        dt = np.dtype(r.choice(DTYPES))
        nc = int(r.choice([1, 2, 3, 7, 16, 64, 100, 385, 400, 1024]))
        nchunks = int(r.randint(1, 6))
        rows = [int(r.choice([1, 2, 5, 63, 64, 65, 300, 1000, 2999, 7500])) for _ in range(nchunks)]
        if r.randint(0, 12) == 0:
            rows = [int(r.randint(20000, 120000))]
            nc = int(r.choice([1, 4, 16]))
        nt = sum(rows)
        kind = int(r.randint(0, 10))
        x = content(r, kind, nt, nc).astype(dt)
        fl = int(r.randint(0, 8))
        td, sd, of = bool(fl & 1), bool(fl & 2), 'F' if fl & 4 else 'C'
        b = np.concatenate(([0], np.cumsum(rows)))
        if os.environ.get('FUZZ_TRACE'):
            print('case dtype %s nc %d rows %s kind %d flags %d' % (dt, nc, rows, kind, fl), flush=True)
        lv = os.environ.get('FUZZ_LEVELS')
        level = int(r.choice([1, 2, 3, 1, 2, 3, 6, 9])) if lv == '2' else int(r.choice([6, 6, 6, 4, 5, 7, 8, 9])) if lv else 6
        try:
            z = hip.compress_chunks(x, b, fl, level)
        except Exception:
            print('FAILED case dtype %s nc %d rows %s kind %d flags %d level %d (chunk count so far %d)' % (dt, nc, rows, kind, fl, level, n), flush=True)
            np.save('gpurun_out/fuzz_fail_x.npy', x)
            raise
        st, arrs = hip.decompress_chunks(z, rows, nc, dt, fl)
        for i in range(len(rows)):
            c = x[b[i]:b[i + 1]]
            with np.errstate(all='ignore'):
                want = O.ref_compress_chunk(c, td, sd, of, level)
                ok = z[i] == want and st[i] == 0 and \
                    arrs[i].tobytes() == O.ref_decompress_chunk(want, rows[i], nc, dt, td, sd, of).tobytes()
            if not ok:
                bad += 1
                print('MISMATCH dtype %s nc %d rows %s kind %d flags %d level %d chunk %d: %d vs %d bytes, status %d'
                      % (dt, nc, rows, kind, fl, level, i, len(z[i]), len(want), st[i]))
        n += len(rows)
    print('fuzz seed %d: %d chunks, %d mismatches' % (seed, n, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
