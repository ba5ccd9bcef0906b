"""Golden fixtures produced by importing the reference (oracle/gen_golden.py) pin
  (a) the oracle's chunk restatement (C and numpy+zlib) byte for byte, and
  (b) the host layer (.cbin/.ch writer, Reader slicing) of mtscomp_amd, driven here through the
      test-only OracleCodec (the GPU run of the same checks is tests/test_gpu_api.py)."""
import hashlib
import json
import zlib
from pathlib import Path

import numpy as np
import pytest

import mtscomp_amd
from mtscomp_amd import api
from mtscomp_amd.synth import synth_int16
from oracle import oracle as O
from tests.codec_oracle import OracleCodec

GOLD = Path(__file__).resolve().parent / 'golden'
MANIFEST = json.loads((GOLD / 'manifest.json').read_text())
CASES = {c['name']: c for c in MANIFEST['cases']}


def sha1(b):
    return hashlib.sha1(bytes(b)).hexdigest()


def make_input(case):
    spec = case['input']
    kind = spec['kind']
    if kind == 'synth':
        return synth_int16(spec['t0'], spec['t1'], spec['nc'], spec['seed'])
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
    if kind in ('randstate', 'randn'):
        return np.load(GOLD / (case['name'] + '.input.npy'))
    raise ValueError(kind)


def same_as_reference_decode(got, case, arr):
    """The reader's result against what the reference's reader returned: integers come back exactly; for floats diff
    followed by cumsum is not the identity (mtscomp.py:884-885) and the reference's own result is what is pinned."""
    got = np.ascontiguousarray(got)
    if arr.dtype.kind == 'f':
        return got.dtype == arr.dtype and sha1(got.tobytes()) == case['decoded_sha1'] and np.allclose(got, arr, atol=1e-4)
    return np.array_equal(got, arr)


def golden_cbin(case):
    p = GOLD / (case['name'] + '.cbin')
    return p.read_bytes() if p.exists() else None


def parse_slice(s):
    return slice(*[int(p) if p else None for p in s.split(':')])


@pytest.fixture
def tmp_cfg(tmp_path, monkeypatch):
    monkeypatch.setattr(api, 'CONFIG_PATH', tmp_path / '.mtscomp')
    return tmp_path


def test_manifest_environment():
    assert MANIFEST['zlib'].startswith('1.2.11')


@pytest.mark.parametrize('name', sorted(CASES))
def test_inputs_reproduce(name):
    case = CASES[name]
    arr = make_input(case)
    assert list(arr.shape) == case['shape'] and str(arr.dtype) == case['dtype']
    assert sha1(arr.tobytes()) == case['raw_sha1']


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_chunks_equal_reference(name):
    """Every chunk of every fixture: C oracle == numpy+zlib restatement == the reference's bytes."""
    case = CASES[name]
    arr = make_input(case)
    meta = json.loads(case['ch_text'])
    flags = O.make_flags(meta['do_time_diff'], meta['do_spatial_diff'], meta['chunk_order'])
    cb, co = meta['chunk_bounds'], meta['chunk_offsets']
    cbin = golden_cbin(case)
    h = hashlib.sha1()
    for i in range(len(cb) - 1):
        chunk = arr[cb[i]:cb[i + 1]]
        stream = O.delta_transpose(chunk, flags).tobytes()
        assert sha1(stream) == case['streams'][i]['sha1']
        assert O.adler32(stream) == case['streams'][i]['adler32'] == zlib.adler32(stream)
        z = O.compress_chunk(chunk, flags, 6)
        assert len(z) == co[i + 1] - co[i]
        assert z == O.ref_compress_chunk(chunk, meta['do_time_diff'], meta['do_spatial_diff'], meta['chunk_order'])
        if cbin is not None:
            assert z == cbin[co[i]:co[i + 1]]
            rc, back = O.decompress_chunk(cbin[co[i]:co[i + 1]], cb[i + 1] - cb[i], arr.shape[1], arr.dtype, flags)
            want = O.ref_decompress_chunk(cbin[co[i]:co[i + 1]], cb[i + 1] - cb[i], arr.shape[1], arr.dtype, meta['do_time_diff'],
                                          meta['do_spatial_diff'], meta['chunk_order'])
            assert rc == 0 and back.tobytes() == want.tobytes() and (arr.dtype.kind == 'f' or np.array_equal(back, chunk))
        h.update(z)
    assert h.hexdigest() == case['cbin_sha1'] == meta['sha1_compressed']


@pytest.mark.parametrize('name', sorted(n for n in CASES if not CASES[n]['input'].get('nc', 0) == 64))
def test_host_layer_writes_reference_files(name, tmp_cfg):
    """compress() through the host layer: .cbin and .ch byte-identical to the reference's."""
    case = CASES[name]
    arr = make_input(case)
    raw = tmp_cfg / 'data.bin'
    arr.tofile(raw)
    out, outmeta = tmp_cfg / 'data.cbin', tmp_cfg / 'data.ch'
    codec = OracleCodec()
    ratio = mtscomp_amd.compress(raw, out, outmeta, sample_rate=case['sample_rate'], n_channels=arr.shape[1],
                                 dtype=arr.dtype, codec=codec, **case['kwargs'])
    assert ratio == pytest.approx(case['ratio'], rel=0, abs=0)
    assert sha1(out.read_bytes()) == case['cbin_sha1']
    assert outmeta.read_text() == case['ch_text']
    r = mtscomp_amd.decompress(out, outmeta, codec=codec)
    assert same_as_reference_decode(r[:], case, arr)
    for s in case['slices']:
        v = r[parse_slice(s['s'])]
        assert list(v.shape) == s['shape'] and sha1(np.ascontiguousarray(v).tobytes()) == s['sha1'], s['s']
    if case['row_17_cols'] is not None:
        assert r[17 % arr.shape[0], 1:4].tolist() == case['row_17_cols']
    for i0, i1, c0, c1 in case['chunks_for_interval']:
        assert r._chunks_for_interval(i0, i1) == (c0, c1)
    r.close()


def test_reader_opens_reference_file_and_chop(tmp_cfg):
    case = CASES['ar1_8ch_3chunks']
    out = tmp_cfg / 'ref.cbin'
    out.write_bytes(golden_cbin(case))
    out.with_suffix('.ch').write_text(case['ch_text'])
    codec = OracleCodec()
    r = mtscomp_amd.Reader(codec=codec)
    r.open(out)                       # .ch found next to the .cbin
    assert r.shape == (3000, 8) and r.n_chunks == 3
    cho = tmp_cfg / 'chopped.cbin'
    r.chop(2, cho)
    r.close()
    assert sha1(cho.read_bytes()) == case['chop2']['cbin_sha1']
    assert cho.with_suffix('.ch').read_text() == case['chop2']['ch_text']
    r2 = mtscomp_amd.decompress(cho, codec=codec)
    assert r2.n_chunks == 2 and np.array_equal(r2[:], make_input(case)[:2000])
    r2.close()
