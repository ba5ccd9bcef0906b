"""Rounds and scorings of k_match5's candidate loops on the headline workload, from a build with -DMTS_M5_STATS=1
(tools/build_variant.sh stats -DMTS_M5_STATS=1; MTSCOMP_HIP_LIB=gpurun_scratch/lib_stats.so python tools/m5_stats.py)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from mtscomp_amd import hip  # noqa: E402

nc = 385; rate = 30000; n = 8; cb = rate * nc * 2
raw = hip.DevBuffer(n * cb)
hip.dev_synth_int16(raw, 0, 0, n * rate, nc, 0)
bound = (hip.compress_bound(cb) + 255) // 256 * 256
cbuf = hip.DevBuffer(n * bound)
b = np.arange(n + 1, dtype=np.int64) * rate; sl = np.arange(n, dtype=np.int64) * bound; sz = np.zeros(n, dtype=np.int64)
out = (C.c_ulonglong * 16)()
L = hip.lib()
L.mts_debug_m5_stats(out)
hip.dev_compress_chunks(raw, nc, 2, b, 5, 6, cbuf, sl, sz)
assert L.mts_debug_m5_stats(out) == 0
g, r1, r2, s1, s2 = [int(v) for v in out[:5]]
print("groups walked %d (%.2f per 64 owned positions)" % (g, g / (n * cb / 64)))
print("rounds per group: newest word %.2f, other 96 %.2f; lane use %.1f%% / %.1f%%; scorings per owned position %.3f + %.3f"
      % (r1 / g, r2 / g, 100 * s1 / (64 * r1), 100 * s2 / (64 * max(r2, 1)), s1 / (n * cb), s2 / (n * cb)))
t = [int(v) for v in out[5:11]]
if sum(t):                                                                       # -DMTS_M5_STATS=2: wave clocks per phase
    names = ["commit (slot entry, keys, tables)", "budget + row masks", "newest word (head + rounds)", "other 96 (rounds)", "store", "groups without an owned slot"]
    tot = sum(t)
    for nm, v in zip(names, t):
        print("  %-36s %7.0f clocks per group walked (%4.1f %%)" % (nm, v / g, 100.0 * v / tot))
    print("  all: %.0f clocks of a wave per group walked" % (tot / g))
