"""A/B of library builds on one box: per-stage device times of the headline workload (60 x 23.1 MB chunks, level 6) for each
library given, alternating, every run in a fresh process; prints the sha1 of the compressed bytes so that a variant that is
faster because it is wrong shows.

    python tools/ab_stage_times.py [--level 6] [--chunks 60] [--reps 2] libA.so libB.so ...
"""
import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

CHILD = r'''
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %r)
from mtscomp_amd import hip
level, n = int(sys.argv[1]), int(sys.argv[2])
nc = 385; rate = 30000; cb = rate * nc * 2
raw = hip.DevBuffer(n * cb)
hip.dev_synth_int16(raw, 0, 0, n * rate, nc, 0)
bound = (hip.compress_bound(cb) + 255) // 256 * 256
cbuf, back = hip.DevBuffer(n * bound), hip.DevBuffer(n * cb)
b = np.arange(n + 1, dtype=np.int64) * rate; sl = np.arange(n, dtype=np.int64) * bound; sz = np.zeros(n, dtype=np.int64)
rows = np.full(n, rate, dtype=np.int64); oo = np.arange(n, dtype=np.int64) * cb; st = np.zeros(n, dtype=np.int32)
acc = {}
for i in range(4):
    hip.dev_compress_chunks(raw, nc, 2, b, 5, level, cbuf, sl, sz)
    if i:
        for k, v in hip.last_stage_times(): acc.setdefault(k, []).append(v)
    hip.dev_decompress_chunks(cbuf, sl, sz, rows, nc, 2, 5, back, oo, st)
    assert not st.any()
    if i:
        for k, v in hip.last_stage_times(): acc.setdefault(k, []).append(v)
host = cbuf.download()
h = hashlib.sha1()
for k in range(n): h.update(host[int(sl[k]):int(sl[k]) + int(sz[k])].tobytes())
print(json.dumps({"sha1": h.hexdigest(), "ok": back.diff(raw)[0] == 0, "ms": {k: round(min(v), 3) for k, v in acc.items()}}))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--level', type=int, default=6)
    ap.add_argument('--chunks', type=int, default=60)
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--stages', default='delta_transpose,hash_sort,match,parse_fixpoint,parse_emit,block_trees,block_pack,inflate_scan,inflate_passA,inflate_passB,inflate_lz,cumsum_transpose')
    ap.add_argument('libs', nargs='+')
    a = ap.parse_args()
    stages = a.stages.split(',')
    for rep in range(a.reps):
        for spec in a.libs:                                  # path[@ENV=VALUE[@ENV=VALUE...]]: a library build, plus environment switches for that run
            lib, *sets = spec.split('@')
            env = dict(os.environ, MTSCOMP_HIP_LIB=str(Path(lib).resolve()), PYTHONWARNINGS='ignore')
            env.update(dict(kv.split('=', 1) for kv in sets))
            r = subprocess.run([sys.executable, '-c', CHILD % str(ROOT), str(a.level), str(a.chunks)], env=env, capture_output=True, text=True, timeout=900)
            line = r.stdout.strip().split('\n')[-1] if r.stdout.strip() else ''
            try:
                d = json.loads(line)
                ms = d['ms']
                comp = sum(v for k, v in ms.items() if not k.startswith('inflate') and k not in ('adler32', 'cumsum_transpose'))
                print('%-28s %s ok=%s  compress %.2f  ' % (Path(lib).name + ''.join('@' + x for x in sets), d['sha1'][:10], d['ok'], comp) + ' '.join('%s %.2f' % (k.replace('inflate_', 'i_')[:10], ms[k]) for k in stages if k in ms), flush=True)
            except Exception:  # noqa: BLE001
                print('%-28s FAILED rc=%d %s' % (spec, r.returncode, (r.stderr or r.stdout)[-400:]), flush=True)


if __name__ == '__main__':
    main()
