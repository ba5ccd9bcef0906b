"""Where the segment resolver's workers spend their clocks (k_inf_lz_seg) on the headline workload, from a build with
-DMTS_LZ2_STATS=1 (tools/build_variant.sh lz2s -DMTS_LZ2_STATS=1; MTSCOMP_HIP_LIB=gpurun_scratch/lib_lz2s.so python tools/lz2_stats.py).
The instrumented kernel is ~40 % slower than the plain one; the proportions are what it is for."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
from mtscomp_amd import hip
nc = 385; rate = 30000; n = 60; cb = rate * nc * 2
raw = hip.DevBuffer(n * cb)
hip.dev_synth_int16(raw, 0, 0, n * rate, nc, 0)
bound = (hip.compress_bound(cb) + 255) // 256 * 256
cbuf = hip.DevBuffer(n * bound)
b = np.arange(n + 1, dtype=np.int64) * rate; sl = np.arange(n, dtype=np.int64) * bound; sz = np.zeros(n, dtype=np.int64)
hip.dev_compress_chunks(raw, nc, 2, b, 5, 6, cbuf, sl, sz)
out = hip.DevBuffer(n * cb)
rows = np.full(n, rate, dtype=np.int64); oo = np.arange(n, dtype=np.int64) * cb; st = np.zeros(n, dtype=np.int32)
L = hip.lib()
o = (C.c_ulonglong * 16)()
for rep in range(2):
    L.mts_debug_lz2_stats(o)
    hip.dev_decompress_chunks(cbuf, sl, sz, rows, nc, 2, 5, out, oo, st)
    assert not st.any()
    L.mts_debug_lz2_stats(o)
    v = [int(x) for x in o[:12]]; c_ld = int(o[12])
    g, it, nm, tot, cw, cp, cl, pend, slow, first, ncp, waves = v
    print("waves %d groups %d  iters/group %.2f  no-move iters/group %.3f  copies/group %.1f slow/group %.2f" % (waves, g, it / g, nm / g, ncp / g, slow / g))
    print("  pending lanes per iter %.1f; moved in first iter %.1f per group" % (pend / it, first / g))
    print("  cycles per group per worker: all %.0f  wait(flush ring) %.0f  pre %.0f  loop %.0f  (loop per iter %.0f)" % (tot / g, cw / g, cp / g, cl / g, cl / it))
    print("  per worker: total cycles %.0f; the window load and its wait: %.0f cycles per round (lane 0's clock)" % (tot / waves, c_ld / it))
    print(" ".join("%s %.2f" % (k[:12], x) for k, x in hip.last_stage_times()))
