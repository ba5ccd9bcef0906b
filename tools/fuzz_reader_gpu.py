"""Randomised file-level parity on a GPU box: random recordings (dtype, shape, chunk duration, flags) compressed with
mtscomp_amd.compress, read back through Reader[...] with random slices / steps / column picks, with the decoded-chunk cache
on, tiny, or off and random batch sizes; every result compared with numpy indexing of the raw array (bit for bit for
integers, against the oracle's decode for floats).

    python tools/fuzz_reader_gpu.py [seed] [seconds]
"""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import mtscomp_amd  # noqa: E402
from mtscomp_amd import api  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.
    r = np.random.RandomState(seed)
    tmp = Path(tempfile.mkdtemp(prefix='mtsfuzz_'))
    api.CONFIG_PATH = tmp / '.mtscomp'
    t0 = time.time()
    files = reads = bad = 0
    while time.time() - t0 < budget:
        dt = np.dtype(r.choice(['uint8', 'int16', 'int16', 'uint16', 'int32', 'int64']))
        nc = int(r.choice([1, 3, 16, 100, 385]))
        nt = int(r.choice([1, 7, 500, 4000, 30000]) * r.uniform(0.5, 1.5)) + 1
        rate = float(r.choice([100., 1000., 2500.]))
        arr = np.cumsum(r.randint(-3, 4, size=(nt, nc)), axis=0).astype(dt)
        raw, out, meta = tmp / 'd.bin', tmp / 'd.cbin', tmp / 'd.ch'
        arr.tofile(raw)
        os.environ['MTSCOMP_DEVICE_CACHE_GB'] = str(r.choice(['8', '0', '0.00005', '0.001']))
        kw = dict(chunk_duration=float(r.choice([0.05, 0.5, 1., 3.])), do_spatial_diff=bool(r.randint(0, 2)),
                  do_time_diff=bool(r.randint(0, 4) > 0), chunk_order=str(r.choice(['F', 'C'])))
        mtscomp_amd.compress(raw, out, meta, sample_rate=rate, n_channels=nc, dtype=dt, check_after_compress=bool(r.randint(0, 2)), **kw)
        rd = mtscomp_amd.decompress(out, meta)
        rd.batch_size = int(r.choice([1, 2, 5, 64]))
        rd.set_cache_size(int(r.choice([1, 2, 10])))
        files += 1
        for _ in range(25):
            a, b = sorted(int(v) for v in r.randint(-nt - 3, nt + 3, size=2))
            step = [None, 1, 2, 3, 17][r.randint(0, 5)]
            k = r.randint(0, 5)
            if k == 0:
                item = slice(a, b, step)
            elif k == 1:
                item = (slice(a, b, step), slice(int(r.randint(0, nc)), None, int(r.randint(1, 3))))
            elif k == 2:
                item = int(r.randint(-nt, nt))
            elif k == 3:
                item = (slice(a, b), int(r.randint(0, nc)))
            else:
                item = slice(None, None, step)
            got, want = rd[item], arr[item]
            reads += 1
            if got.shape != want.shape or got.dtype != want.dtype or not np.array_equal(got, want):
                bad += 1
                print('MISMATCH', dt, (nt, nc), kw, os.environ['MTSCOMP_DEVICE_CACHE_GB'], rd.batch_size, item)
        rd.close()
        for f in (raw, out, meta):
            f.unlink()
    print('reader fuzz seed %d: %d files, %d reads, %d mismatches' % (seed, files, reads, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
