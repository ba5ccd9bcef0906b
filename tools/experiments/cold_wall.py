"""Where the wall time of ONE cold 1 s window goes (read-ahead off): pread into page-locked memory, the codec call (copy in,
decode, rows out), the rest of Reader.__getitem__.   MTSCOMP_READ_AHEAD=0 python tools/experiments/cold_wall.py"""
import os, sys, time, tempfile, shutil
from pathlib import Path
import numpy as np
sys.path.insert(0, '.')
os.environ.setdefault('MTSCOMP_READ_AHEAD', '0')
import bench, mtscomp_amd
from mtscomp_amd import hip, api
tmp = Path(tempfile.mkdtemp(dir='/dev/shm')); os.environ['HOME'] = str(tmp)
n_s, _ = bench.build_synth_file(hip, 0, 200, tmp, 385)
r = mtscomp_amd.decompress(tmp / 'data.cbin', tmp / 'data.ch')
acc = {'pread': [], 'codec': [], 'total': []}
orig_pread, orig_rows = r._pread_pinned, r.codec.cache_read_rows
def pread(nbytes, base):
    t0 = time.perf_counter(); v = orig_pread(nbytes, base); acc['pread'].append(time.perf_counter() - t0); return v
def rows(*a, **k):
    t0 = time.perf_counter(); v = orig_rows(*a, **k); acc['codec'].append(time.perf_counter() - t0); return v
r._pread_pinned = pread
r.codec.cache_read_rows = rows
for k in range(2, 130, 3):
    s = k * 30000 + 15000
    t0 = time.perf_counter(); x = r[s:s + 30000]; acc['total'].append(time.perf_counter() - t0)
    del x
print('cold window (2 chunks, read-ahead %s): ' % os.environ['MTSCOMP_READ_AHEAD'] +
      ', '.join('%s %.3f ms (median %.3f)' % (k, np.mean(v) * 1e3, np.median(v) * 1e3) for k, v in acc.items()))
print('device stages: ' + ', '.join('%s %.3f' % kv for kv in hip.last_stage_times(0)))
r.close(); shutil.rmtree(tmp)
