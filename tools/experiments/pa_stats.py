import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
from mtscomp_amd import hip
nc = 385; rate = 30000; n = 8; cb = rate * nc * 2
raw = hip.DevBuffer(n * cb); hip.dev_synth_int16(raw, 0, 0, n * rate, nc, 0)
bound = (hip.compress_bound(cb) + 255) // 256 * 256
cbuf, back = hip.DevBuffer(n * bound), hip.DevBuffer(n * cb)
b = np.arange(n + 1, dtype=np.int64) * rate; sl = np.arange(n, dtype=np.int64) * bound; sz = np.zeros(n, dtype=np.int64)
rows = np.full(n, rate, dtype=np.int64); oo = np.arange(n, dtype=np.int64) * cb; st = np.zeros(n, dtype=np.int32)
hip.dev_compress_chunks(raw, nc, 2, b, 5, 6, cbuf, sl, sz)
out = (C.c_ulonglong * 8)()
hip.lib().mts_debug_pa_stats(out)
hip.dev_decompress_chunks(cbuf, sl, sz, rows, nc, 2, 5, back, oo, st)
hip.lib().mts_debug_pa_stats(out)
j, nj, m, subs, jc, mf, nocp, nt = [int(x) for x in out]
print('sub-sequences %d (pieces per sub-sequence %.1f); second decodes: joined %d, not joined %d (of them without a checkpoint %d)' % (subs, nt / max(subs, 1), j, nj, nocp))
print('pieces before the join %.2f on average; checkpoint piece index %.1f on average; pieces wasted before giving up %.1f' % (m / max(j, 1), jc / max(j, 1), mf / max(nj, 1)))
