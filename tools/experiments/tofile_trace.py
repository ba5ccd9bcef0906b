import os, sys, time, tempfile, shutil, threading
from pathlib import Path
import numpy as np
sys.path.insert(0, '.')
import bench, mtscomp_amd
from mtscomp_amd import hip, api
tmp = Path(tempfile.mkdtemp(dir='/dev/shm')); os.environ['HOME'] = str(tmp)
n_s, cb = bench.build_synth_file(hip, 0, 60, tmp, 385)
nbytes = n_s * 385 * 2
back = tmp / 'back.bin'
os.environ['MTSCOMP_TOFILE_WRITERS'] = sys.argv[1] if len(sys.argv) > 1 else '2'
ev = []
T0 = [0]
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); ev.append((label, t0 - T0[0], time.perf_counter() - T0[0])); return r
    setattr(obj, name, g)
wrap(os, 'pwrite', 'pwrite'); wrap(os, 'preadv', 'preadv'); wrap(hip, 'decompress_chunks', 'decode')
for rep in range(3):
    ev.clear(); T0[0] = time.perf_counter()
    r = mtscomp_amd.decompress(tmp / 'data.cbin', tmp / 'data.ch', back, overwrite=True, check_after_decompress=False)
    t_done = time.perf_counter() - T0[0]
    r.close()
    t_all = time.perf_counter() - T0[0]
    print('rep %d: tofile returns at %.1f ms, close at %.1f ms  (%.2f GB/s)' % (rep, t_done * 1e3, t_all * 1e3, nbytes / t_all / 1e9))
    for lab in ('preadv', 'decode', 'pwrite'):
        xs = [(a, b) for l, a, b in ev if l == lab]
        print('  %-7s n=%3d first start %.1f ms, last end %.1f ms, mean %.1f ms, sum %.1f ms' % (lab, len(xs), min(a for a, b in xs) * 1e3, max(b for a, b in xs) * 1e3,
              np.mean([b - a for a, b in xs]) * 1e3, sum(b - a for a, b in xs) * 1e3))
    print('  decode calls:', ' '.join('%.0f-%.0f' % (a * 1e3, b * 1e3) for l, a, b in ev if l == 'decode'))
    print('  ', hip.last_stage_times())
shutil.rmtree(tmp)
