"""Generates tests/golden/* by IMPORTING THE REFERENCE (/root/reference/mtscomp.py).

Runs only in the build container (the reference does not exist on the GPU box).  The fixtures are
data: input arrays (or the seed of the integer generator that makes them), the reference's `.cbin`
bytes (or sha1 + length for the large case), the full `.ch` JSON text, per-chunk sha1/adler32 of the
transformed byte stream, and a few `Reader[...]` results.  Nothing of the reference's source is kept.

    python oracle/gen_golden.py        # rewrites tests/golden/
"""
import hashlib
import json
import sys
import tempfile
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, '/root/reference')

import mtscomp as ref  # noqa: E402  (the reference, imported from /root/reference)

from mtscomp_amd.synth import synth_int16  # noqa: E402

GOLD = ROOT / 'tests' / 'golden'


def sha1(b):
    return hashlib.sha1(bytes(b)).hexdigest()


def make_input(spec):
    kind = spec['kind']
    if kind == 'synth':
        return synth_int16(spec['t0'], spec['t1'], spec['nc'], spec['seed']).astype(spec.get('dtype', 'int16'))
    if kind == 'zeros':
        return np.zeros(spec['shape'], dtype=spec['dtype'])
    if kind == 'ramp':
        n = spec['shape'][0] * spec['shape'][1]
        return (np.arange(n, dtype=np.int64) * spec['step']).astype(spec['dtype']).reshape(spec['shape'])
    if kind == 'wrap':
        a = np.empty(spec['shape'], dtype=np.int16)
        a[0::2] = 32767
        a[1::2] = -32767
        a[:, 1::2] *= -1
        return a
    if kind == 'randstate':
        r = np.random.RandomState(spec['seed'])
        info = np.iinfo(spec['dtype'])
        lo, hi = spec.get('lo', info.min), spec.get('hi', int(info.max) + 1)
        return r.randint(lo, hi, size=spec['shape'], dtype=np.int64).astype(spec['dtype'])
    if kind == 'randn':           # like the reference's own float fixtures (tests.py:70-92): noise around a slow sine
        r = np.random.RandomState(spec['seed'])
        t = np.arange(spec['shape'][0])[:, np.newaxis] / 250.
        return (np.sin(10 * t) + r.normal(0, spec['scale'], size=spec['shape'])).astype(spec['dtype'])
    raise ValueError(kind)


CASES = [
    dict(name='ar1_8ch_3chunks', input=dict(kind='synth', t0=0, t1=3000, nc=8, seed=0), sample_rate=1000.),
    dict(name='ar1_8ch_short_last', input=dict(kind='synth', t0=0, t1=2750, nc=8, seed=2), sample_rate=1000.),
    dict(name='np385_1sample', input=dict(kind='synth', t0=0, t1=1, nc=385, seed=0), sample_rate=30000.),
    dict(name='np385_8samples', input=dict(kind='synth', t0=0, t1=8, nc=385, seed=1), sample_rate=30000.),
    dict(name='zeros', input=dict(kind='zeros', shape=[5000, 7], dtype='int16'), sample_rate=2500.),
    dict(name='ramp', input=dict(kind='ramp', shape=[4000, 5], dtype='int16', step=3), sample_rate=2000.),
    dict(name='wrap_fullrange', input=dict(kind='wrap', shape=[3000, 6]), sample_rate=1500.),
    dict(name='uniform_random_stored', input=dict(kind='randstate', seed=11, shape=[6000, 4], dtype='int16'),
         sample_rate=3000.),
    dict(name='multiblock_24ch', input=dict(kind='synth', t0=100, t1=4100, nc=24, seed=5), sample_rate=4000.),
    dict(name='spatial_diff', input=dict(kind='synth', t0=0, t1=2000, nc=12, seed=6), sample_rate=1000.,
         kwargs=dict(do_spatial_diff=True)),
    dict(name='no_time_diff', input=dict(kind='synth', t0=0, t1=2000, nc=12, seed=7), sample_rate=1000.,
         kwargs=dict(do_time_diff=False)),
    dict(name='both_diffs_order_c', input=dict(kind='synth', t0=0, t1=2000, nc=12, seed=8), sample_rate=1000.,
         kwargs=dict(do_spatial_diff=True, chunk_order='C')),
    dict(name='order_c', input=dict(kind='synth', t0=0, t1=2000, nc=12, seed=9), sample_rate=1000.,
         kwargs=dict(chunk_order='C')),
    dict(name='uint16', input=dict(kind='randstate', seed=12, shape=[3000, 9], dtype='uint16', lo=30000, hi=30040),
         sample_rate=1000.),
    dict(name='uint8', input=dict(kind='randstate', seed=13, shape=[5000, 10], dtype='uint8', lo=0, hi=255),
         sample_rate=2500.),
    dict(name='int32', input=dict(kind='randstate', seed=14, shape=[2000, 6], dtype='int32', lo=-70000, hi=70000),
         sample_rate=1000.),
    dict(name='tiny_chunks', input=dict(kind='synth', t0=0, t1=700, nc=19, seed=10), sample_rate=1234.,
         kwargs=dict(chunk_duration=.01)),
    # float dtypes go through the reference as they are (tests.py:212-237): diff / cumsum in the item type
    # (float32 noise does not pass the reference's own post-compression check, atol 1e-16: it is switched off, as a user must)
    dict(name='float32', input=dict(kind='randn', seed=21, shape=[2500, 6], dtype='float32', scale=.1), sample_rate=1000.,
         kwargs=dict(check_after_compress=False)),
    dict(name='float64_spatial', input=dict(kind='randn', seed=22, shape=[1200, 5], dtype='float64', scale=.3), sample_rate=500.,
         kwargs=dict(do_spatial_diff=True)),
    dict(name='comp_level_ignored', input=dict(kind='synth', t0=0, t1=1500, nc=8, seed=0), sample_rate=1000.,
         kwargs=dict(comp_level=1)),
    # BASELINE config-1 shape, first two chunks: too big to commit -> sha1 + length only
    dict(name='ar1_64ch_2s_30k', input=dict(kind='synth', t0=0, t1=60000, nc=64, seed=0), sample_rate=30000.,
         big=True),
]

SLICES = ['0:10', '990:1010', '-5:', '::97', '1234:1234', '5:2000:3']


def parse_slice(s):
    parts = [int(p) if p else None for p in s.split(':')]
    return slice(*parts)


def run_case(case, tmp):
    arr = make_input(case['input'])
    kwargs = dict(case.get('kwargs', {}))
    raw = tmp / (case['name'] + '.bin')
    arr.tofile(raw)
    out, outmeta = tmp / (case['name'] + '.cbin'), tmp / (case['name'] + '.ch')
    ratio = ref.compress(raw, out, outmeta, sample_rate=case['sample_rate'], n_channels=arr.shape[1],
                         dtype=arr.dtype, n_threads=1, quiet=True, **kwargs)
    cbin = out.read_bytes()
    ch_text = outmeta.read_text()
    meta = json.loads(ch_text)
    entry = dict(name=case['name'], input=case['input'], sample_rate=case['sample_rate'], kwargs=kwargs,
                 dtype=str(arr.dtype), shape=list(arr.shape), ratio=ratio, raw_sha1=sha1(arr.tobytes()),
                 cbin_sha1=sha1(cbin), cbin_len=len(cbin), ch_text=ch_text)
    # per-chunk transformed stream digests (what zlib.compress was fed, mtscomp.py:394)
    streams = []
    cb, co = meta['chunk_bounds'], meta['chunk_offsets']
    for i in range(len(cb) - 1):
        st = zlib.decompress(cbin[co[i]:co[i + 1]])
        streams.append(dict(sha1=sha1(st), adler32=zlib.adler32(st), n=len(st)))
    entry['streams'] = streams
    # reader results
    r = ref.decompress(out, outmeta, quiet=True)
    full = r[:]
    if arr.dtype.kind == 'f':       # diff followed by cumsum is not exact in floating point (mtscomp.py:884-885)
        assert np.allclose(full, arr, atol=1e-4)
    else:
        assert np.array_equal(full, arr)
    entry['decoded_sha1'] = sha1(np.ascontiguousarray(full).tobytes())
    entry['slices'] = []
    for s in SLICES:
        v = r[parse_slice(s)]
        entry['slices'].append(dict(s=s, shape=list(v.shape), sha1=sha1(np.ascontiguousarray(v).tobytes())))
    entry['row_17_cols'] = r[17 % arr.shape[0], 1:4].tolist() if arr.shape[1] >= 4 else None
    tbl = []
    n = arr.shape[0]
    for i0, i1 in [(0, 0), (0, 1), (n // 2, n // 2 + 1), (0, n), (n - 1, 10 * n), (cb[1] - 1, cb[1]), (cb[1], cb[1])]:
        tbl.append([int(i0), int(i1)] + [int(v) for v in r._chunks_for_interval(i0, i1)])
    entry['chunks_for_interval'] = tbl
    r.close()
    if not case.get('big'):
        (GOLD / (case['name'] + '.cbin')).write_bytes(cbin)
        if case['input']['kind'] in ('randstate', 'randn'):
            np.save(GOLD / (case['name'] + '.input.npy'), arr)
    # chopped file (mtscomp.py:750-796)
    if case['name'] == 'ar1_8ch_3chunks':
        r = ref.Reader()
        r.open(out, outmeta)
        cho = tmp / 'chopped.cbin'
        r.chop(2, cho)
        r.close()
        entry['chop2'] = dict(cbin_sha1=sha1(cho.read_bytes()), ch_text=cho.with_suffix('.ch').read_text())
    return entry


def main():
    GOLD.mkdir(parents=True, exist_ok=True)
    for f in GOLD.glob('*'):
        f.unlink()
    entries = []
    with tempfile.TemporaryDirectory() as td:
        ref.CONFIG_PATH = Path(td) / '.mtscomp'          # keep the user's config out of it
        for case in CASES:
            entries.append(run_case(case, Path(td)))
            print(case['name'], entries[-1]['cbin_len'], entries[-1]['ratio'])
    manifest = dict(reference_version=ref.__version__, zlib=zlib.ZLIB_RUNTIME_VERSION,
                    numpy=np.__version__, cases=entries)
    (GOLD / 'manifest.json').write_text(json.dumps(manifest, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
