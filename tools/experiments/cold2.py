import os, sys, time, tempfile, shutil
from pathlib import Path
import numpy as np
sys.path.insert(0, '.')
import bench, mtscomp_amd
from mtscomp_amd import hip, api
tmp = Path(tempfile.mkdtemp(dir='/dev/shm')); os.environ['HOME'] = str(tmp)
n_s, _ = bench.build_synth_file(hip, 0, 120, tmp, 385)
r = mtscomp_amd.decompress(tmp / 'data.cbin', tmp / 'data.ch')
for k in range(2, 40, 3):
    s = k * 30000 + 15000
    t0 = time.perf_counter(); x = r[s:s + 30000]; dt = time.perf_counter() - t0
    sys.stderr.write('window total %.3f ms\n' % (dt * 1e3))
r.close(); shutil.rmtree(tmp)
