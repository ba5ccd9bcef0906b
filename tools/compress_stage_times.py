"""Two compress passes of the headline workload (60 x 23.1 MB chunks, 385 ch, level 6): the default target of the PMC scripts."""
import sys

import numpy as np

sys.path.insert(0, ".")
from mtscomp_amd import hip  # noqa: E402

nc = 385; rate = 30000; n = 60; cb = rate * nc * 2
raw = hip.DevBuffer(n * cb)
hip.dev_synth_int16(raw, 0, 0, n * rate, nc, 0)
bound = (hip.compress_bound(cb) + 255) // 256 * 256
cbuf = hip.DevBuffer(n * bound)
b = np.arange(n + 1, dtype=np.int64) * rate; sl = np.arange(n, dtype=np.int64) * bound; sz = np.zeros(n, dtype=np.int64)
for i in range(2):
    hip.dev_compress_chunks(raw, nc, 2, b, 5, 6, cbuf, sl, sz)
print(hip.last_stage_times())
