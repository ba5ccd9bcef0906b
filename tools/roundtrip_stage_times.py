"""One compress + two decompress passes of the headline workload (60 x 23.1 MB chunks) -- the target of the PMC scripts when the
inflate kernels are of interest:  bash tools/pmc_sq.sh tools/roundtrip_stage_times.py"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from mtscomp_amd import hip  # noqa: E402

L = hip.lib(); nc = 385; rate = 30000; n = 60
raw = torch.empty((n * rate, nc), dtype=torch.int16, device="cuda")
for k in range(n):
    L.mts_dev_synth_int16(0, None, C.c_void_p(raw[k * rate:].data_ptr()), k * rate, (k + 1) * rate, nc, 0)
bound = (hip.compress_bound(rate * nc * 2) + 255) // 256 * 256
cbuf = torch.empty(n * bound, dtype=torch.uint8, device="cuda")
back = torch.empty_like(raw)
b = np.arange(n + 1, dtype=np.int64) * rate; sl = np.arange(n, dtype=np.int64) * bound; sz = np.zeros(n, dtype=np.int64)
rows = np.full(n, rate, dtype=np.int64); oo = np.arange(n, dtype=np.int64) * rate * nc * 2; st = np.zeros(n, dtype=np.int32)
lp = lambda a: a.ctypes.data_as(C.POINTER(C.c_long))  # noqa: E731
rc = L.mts_dev_compress_chunks(0, None, C.c_void_p(raw.data_ptr()), nc, 2, lp(b), n, 5, 6, C.c_void_p(cbuf.data_ptr()), lp(sl), lp(sz))
assert rc == 0
for i in range(2):
    rc = L.mts_dev_decompress_chunks(0, None, C.c_void_p(cbuf.data_ptr()), lp(sl), lp(sz), lp(rows), n, nc, 2, 5, C.c_void_p(back.data_ptr()), lp(oo),
                                     st.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0 and not st.any()
print(torch.equal(back, raw), hip.last_stage_times())
