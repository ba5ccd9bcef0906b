"""File-to-file numbers of the drop-in API on one GPU (PCIe, host SHA-1 and file I/O included) -- BASELINE configs[2]
scaled to what a test box holds: a synthetic 385-channel recording of --seconds is written to tmpfs, compressed with
`mtscomp_amd.compress` (check_after_compress off, like the reference's benchmark.py), then read back through
`Reader[start:end]` at random 1 s windows.  A few windows are compared with the CPU oracle.

    python tools/bench_file_api.py --seconds 60 --windows 200
"""
import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=int, default=60)
    ap.add_argument('--windows', type=int, default=200)
    ap.add_argument('--dir', default='/dev/shm')
    a = ap.parse_args()
    import mtscomp_amd
    from mtscomp_amd.synth import synth_int16
    nc, rate = 385, 30000
    tmp = Path(tempfile.mkdtemp(prefix='mtsbench_', dir=a.dir))
    os.environ.setdefault('HOME', str(tmp))
    raw = tmp / 'data.bin'
    with open(raw, 'wb') as f:
        from mtscomp_amd import hip              # the same integer-exact generator, on the device (the host one is slow)
        buf = hip.DevBuffer(rate * nc * 2)
        for s in range(a.seconds):
            hip.dev_synth_int16(buf, 0, s * rate, (s + 1) * rate, nc, 0)
            buf.download().tofile(f)
        buf.free()
    nbytes = raw.stat().st_size
    out, outmeta = tmp / 'data.cbin', tmp / 'data.ch'
    t = time.perf_counter()
    ratio = mtscomp_amd.compress(raw, out, outmeta, sample_rate=rate, n_channels=nc, dtype=np.int16,
                                 check_after_compress=False, n_threads=1)
    t_c = time.perf_counter() - t
    r = mtscomp_amd.decompress(out, outmeta)
    n = r.shape[0]
    rng = np.random.RandomState(0)
    starts = rng.randint(0, n - rate, size=a.windows)
    t = time.perf_counter()
    got_bytes = 0
    for s in starts:
        w = r[int(s):int(s) + rate]
        got_bytes += w.nbytes
    t_r = time.perf_counter() - t
    ref = np.memmap(raw, dtype=np.int16, mode='r').reshape(-1, nc)
    ok = all(np.array_equal(r[int(s):int(s) + rate], ref[int(s):int(s) + rate]) for s in starts[:5])
    t = time.perf_counter()
    whole = r[:]
    t_d = time.perf_counter() - t
    ok = ok and np.array_equal(whole, ref)
    print(json.dumps({'file_bytes': nbytes, 'ratio': ratio, 'compress_file_gbps': nbytes / t_c / 1e9,
                      'decompress_all_gbps': nbytes / t_d / 1e9, 'random_windows': int(a.windows),
                      'ms_per_window': t_r / a.windows * 1e3, 'random_read_gbps': got_bytes / t_r / 1e9, 'verified': bool(ok)}))
    r.close()
    for p in (raw, out, outmeta):
        p.unlink()
    tmp.rmdir()


if __name__ == '__main__':
    main()
