import os, sys, time, tempfile, shutil
from pathlib import Path
import numpy as np
sys.path.insert(0, '.')
import bench, mtscomp_amd
from mtscomp_amd import hip
tmp = Path(tempfile.mkdtemp(dir='/dev/shm')); os.environ['HOME'] = str(tmp)
n_s, _ = bench.build_synth_file(hip, 0, 300, tmp, 385)
r = mtscomp_amd.decompress(tmp / 'data.cbin', tmp / 'data.ch')
tot = []; stages = {}
for k in range(2, 200, 3):
    s = k * 30000 + 15000
    t0 = time.perf_counter(); x = r[s:s + 30000]; tot.append(time.perf_counter() - t0)
    for name, ms in hip.last_stage_times(0):
        stages.setdefault(name, []).append(ms)
print('cold window of two chunks: %.2f ms (median %.2f)' % (np.mean(tot) * 1e3, np.median(tot) * 1e3))
print('device stages of the two-chunk batch (ms): ' + ', '.join('%s %.3f' % (k, np.mean(v)) for k, v in stages.items()) + '; sum %.3f' % sum(np.mean(v) for v in stages.values()))
r.close()
shutil.rmtree(tmp)
