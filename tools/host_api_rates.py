"""extras.host_api of the bench line alone (mts_compress_chunks / mts_decompress_chunks on the 60-chunk recording, pageable and
page-locked host memory), e.g. under MTS_HOST_THREADS / MTS_PIPE_BYTES settings:  MTS_HOST_THREADS=16 python tools/host_api_rates.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, '.')
import bench
from mtscomp_amd import hip
nc, n = 385, 60
raw = hip.DevBuffer(n * bench.RATE * nc * 2)
hip.dev_synth_int16(raw, 0, 0, n * bench.RATE, nc, 0)
x = raw.download(dtype=np.int16).reshape(n * bench.RATE, nc)
raw.free()
r = bench.extra_host_api(hip, 0, x, nc)
print(json.dumps({'MTS_HOST_THREADS': os.environ.get('MTS_HOST_THREADS'), 'MTS_PIPE_BYTES': os.environ.get('MTS_PIPE_BYTES'),
                  **{k: {a: round(b, 2) for a, b in v.items() if a in ('compress_gbps', 'decompress_gbps')} for k, v in r.items() if isinstance(v, dict)}}))
