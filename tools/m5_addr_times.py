"""Does k_match5's time follow where its buffers lie?  N fresh processes of the headline compress (60 x 23.1 MB, level 6), each
printing the device addresses of the stream, the sorted keys and the match table (MTS_DEBUG_ADDR=1) next to the match stage's
time -- VERDICT r5 item 8: "takes one of three times from process to process".

    python tools/m5_addr_times.py [N=10] [lib.so]
"""
import os, re, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
from mtscomp_amd import hip
nc = 385; rate = 30000; n = 60; cb = rate * nc * 2
raw = hip.DevBuffer(n * cb)
hip.dev_synth_int16(raw, 0, 0, n * rate, nc, 0)
bound = (hip.compress_bound(cb) + 255) // 256 * 256
cbuf = hip.DevBuffer(n * bound)
b = np.arange(n + 1, dtype=np.int64) * rate; sl = np.arange(n, dtype=np.int64) * bound; sz = np.zeros(n, dtype=np.int64)
ms = []
for i in range(5):
    hip.dev_compress_chunks(raw, nc, 2, b, 5, 6, cbuf, sl, sz)
    ms.append(dict(hip.last_stage_times())['match'])
print('match_ms ' + ' '.join('%%.3f' %% v for v in ms))
''' % str(ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
env = dict(os.environ, MTS_DEBUG_ADDR='1', PYTHONWARNINGS='ignore')
if len(sys.argv) > 2:
    env['MTSCOMP_HIP_LIB'] = str(Path(sys.argv[2]).resolve())
for k in range(n):
    r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=600)
    addr = [ln for ln in r.stderr.splitlines() if ln.startswith('[addr]')]
    ms = [ln for ln in r.stdout.splitlines() if ln.startswith('match_ms')]
    a = addr[-1] if addr else '[addr] ?'
    vals = dict(re.findall(r'(\w+) (0x[0-9a-f]+)', a))
    mods = ' '.join('%s %s (mod 2M %6x, mod 64K %5x, mod 4K %4x)' % (k2, v, int(v, 16) & 0x1fffff, int(v, 16) & 0xffff, int(v, 16) & 0xfff) for k2, v in vals.items())
    print('run %2d  %s  |  %s' % (k, ms[-1] if ms else r.stderr[-300:], mods), flush=True)

# Second question: does the level change INSIDE a process when the workspaces are freed and allocated again (mts_release)?
CHILD2 = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from mtscomp_amd import hip
nc = 385; rate = 30000; n = 60; cb = rate * nc * 2
raw = hip.DevBuffer(n * cb)
hip.dev_synth_int16(raw, 0, 0, n * rate, nc, 0)
bound = (hip.compress_bound(cb) + 255) // 256 * 256
cbuf = hip.DevBuffer(n * bound)
b = np.arange(n + 1, dtype=np.int64) * rate; sl = np.arange(n, dtype=np.int64) * bound; sz = np.zeros(n, dtype=np.int64)
for rep in range(8):
    hip.release()
    ms = []
    for i in range(3):
        hip.dev_compress_chunks(raw, nc, 2, b, 5, 6, cbuf, sl, sz)
        ms.append(dict(hip.last_stage_times())['match'])
    print('after release %%d: match_ms %%s' %% (rep, ' '.join('%%.3f' %% v for v in ms)), flush=True)
""" % str(ROOT)
if os.environ.get('M5_RELEASE_TEST', '1') != '0':
    r = subprocess.run([sys.executable, '-c', CHILD2], env=env, capture_output=True, text=True, timeout=900)
    print('one process, workspaces freed and allocated again between the lines:')
    print(r.stdout.strip() or r.stderr[-500:])
    for ln in r.stderr.splitlines():
        if ln.startswith('[addr]'):
            print('   ', ln)
