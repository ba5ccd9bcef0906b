import os, sys, json
sys.path.insert(0, '.')
import numpy as np
import bench
from mtscomp_amd import hip
from mtscomp_amd.synth import synth_int16
x = synth_int16(0, 60 * 30000, 385, 0)
for w in ('2', '4', '2', '4'):
    os.environ['MTSCOMP_TOFILE_WRITERS'] = w
    r = bench.extra_file_to_file(x, 385, False)
    print('writers', w, 'compress %.2f' % r['compress_gbps'], 'decompress %.2f' % r['decompress_gbps'], json.dumps(r['decompress_gbps_by_call']), flush=True)
