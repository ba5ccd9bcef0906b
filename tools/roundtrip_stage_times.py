"""One compress + two decompress passes of the headline workload (60 x 23.1 MB chunks) -- the target of the PMC scripts:
    bash tools/pmc_sq.sh tools/roundtrip_stage_times.py [compress passes, default 1]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from mtscomp_amd import hip  # noqa: E402

nc = 385; rate = 30000; n = 60; cb = rate * nc * 2
raw = hip.DevBuffer(n * cb)
hip.dev_synth_int16(raw, 0, 0, n * rate, nc, 0)
bound = (hip.compress_bound(cb) + 255) // 256 * 256
cbuf, back = hip.DevBuffer(n * bound), hip.DevBuffer(n * cb)
b = np.arange(n + 1, dtype=np.int64) * rate; sl = np.arange(n, dtype=np.int64) * bound; sz = np.zeros(n, dtype=np.int64)
rows = np.full(n, rate, dtype=np.int64); oo = np.arange(n, dtype=np.int64) * cb; st = np.zeros(n, dtype=np.int32)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    hip.dev_compress_chunks(raw, nc, 2, b, 5, 6, cbuf, sl, sz)
print(hip.last_stage_times())
for i in range(2):
    hip.dev_decompress_chunks(cbuf, sl, sz, rows, nc, 2, 5, back, oo, st)
    assert not st.any()
print(back.diff(raw), hip.last_stage_times())
