"""GPU parity at the API level: the host layer + HipCodec against the golden fixtures (files the reference
wrote), the reference's own property tests re-expressed, and size-independent properties at full size."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

import mtscomp_amd
from mtscomp_amd import api, hip
from mtscomp_amd.synth import synth_int16
from oracle import oracle as O
from tests.test_golden import CASES, golden_cbin, make_input, parse_slice, sha1

pytestmark = pytest.mark.gpu


@pytest.fixture
def tmp_cfg(tmp_path, monkeypatch):
    monkeypatch.setattr(api, 'CONFIG_PATH', tmp_path / '.mtscomp')
    api.set_codec(None)
    return tmp_path


@pytest.mark.parametrize('name', sorted(CASES))
def test_golden_files_byte_identical(name, tmp_cfg):
    case = CASES[name]
    arr = make_input(case)
    raw = tmp_cfg / 'data.bin'
    arr.tofile(raw)
    out, outmeta = tmp_cfg / 'data.cbin', tmp_cfg / 'data.ch'
    ratio = mtscomp_amd.compress(raw, out, outmeta, sample_rate=case['sample_rate'], n_channels=arr.shape[1],
                                 dtype=arr.dtype, **case['kwargs'])
    assert ratio == case['ratio']
    assert sha1(out.read_bytes()) == case['cbin_sha1']
    assert outmeta.read_text() == case['ch_text']
    r = mtscomp_amd.decompress(out, outmeta)
    assert np.array_equal(r[:], arr)
    for s in case['slices']:
        v = r[parse_slice(s['s'])]
        assert list(v.shape) == s['shape'] and sha1(np.ascontiguousarray(v).tobytes()) == s['sha1'], s['s']
    r.close()


def test_reads_reference_written_files(tmp_cfg):
    for name in ('ar1_8ch_3chunks', 'uniform_random_stored', 'tiny_chunks', 'both_diffs_order_c', 'int32'):
        case = CASES[name]
        out = tmp_cfg / (name + '.cbin')
        out.write_bytes(golden_cbin(case))
        out.with_suffix('.ch').write_text(case['ch_text'])
        r = mtscomp_amd.decompress(out)
        assert np.array_equal(r[:], make_input(case)), name
        r.close()


def test_corrupt_chunk_is_ioerror_others_readable(tmp_cfg):
    case = CASES['ar1_8ch_3chunks']
    meta = json.loads(case['ch_text'])
    b = bytearray(golden_cbin(case))
    b[meta['chunk_offsets'][1] + 40] ^= 0xff
    out = tmp_cfg / 'bad.cbin'
    out.write_bytes(bytes(b))
    out.with_suffix('.ch').write_text(case['ch_text'])
    r = mtscomp_amd.decompress(out)
    arr = make_input(case)
    assert np.array_equal(r[0:900], arr[0:900])
    assert np.array_equal(r[2000:3000], arr[2000:3000])
    with pytest.raises(IOError):
        r[1000:1500]
    r.close()


@pytest.mark.parametrize('dtype', ['uint8', 'uint16', 'int8', 'int16', 'int32'])
def test_dtypes_roundtrip(dtype, tmp_cfg):
    # tests.py:240-243: (100 x 1000) from a transposed non-contiguous array
    arr = np.array(np.random.RandomState(1).randint(low=0, high=255, size=(1000, 100)), dtype=dtype).T
    raw = tmp_cfg / 'd.bin'
    arr.tofile(raw)
    mtscomp_amd.compress(raw, tmp_cfg / 'd.cbin', tmp_cfg / 'd.ch', sample_rate=1234., n_channels=arr.shape[1], dtype=arr.dtype)
    r = mtscomp_amd.decompress(tmp_cfg / 'd.cbin', tmp_cfg / 'd.ch')
    assert np.array_equal(r[:], arr)
    r.close()


def test_comp_decomp_files_and_sha1(tmp_cfg):
    # tests.py:381-410
    arr = np.array(np.random.RandomState(2).randint(low=0, high=255, size=(1000, 1000)), dtype=np.int16).T
    raw = tmp_cfg / 'd.bin'
    arr.tofile(raw)
    out, outmeta = tmp_cfg / 'd.cbin', tmp_cfg / 'd.ch'
    mtscomp_amd.compress(raw, out, outmeta, sample_rate=1234., n_channels=1000, dtype=arr.dtype)
    dec = tmp_cfg / 'd.decomp.bin'
    mtscomp_amd.decompress(out, outmeta, out=dec).close()
    assert raw.read_bytes() == dec.read_bytes()
    meta = json.loads(outmeta.read_text())
    assert meta['sha1_compressed'] == sha1(out.read_bytes())
    assert meta['sha1_uncompressed'] == sha1(raw.read_bytes())


def test_diff_cumsum_roundtrip():
    # tests.py:190-205, integer arrays
    x = np.random.RandomState(3).randint(-30000, 30000, size=(6997, 19)).astype(np.int16)
    for ax1 in (None, 0, 1):
        for ax2 in (None, 0, 1):
            if ax1 == ax2 and ax1 is not None:
                continue
            d = mtscomp_amd.diff_along_axis(mtscomp_amd.diff_along_axis(x, ax1), ax2)
            want = O.ref_diff_along_axis(O.ref_diff_along_axis(x, ax1), ax2)
            assert np.array_equal(d, want)
            back = mtscomp_amd.cumsum_along_axis(mtscomp_amd.cumsum_along_axis(d, ax2), ax1)
            assert back.dtype == x.dtype and np.array_equal(back, x)


def test_float_dtype_fails_loudly(tmp_cfg):
    arr = np.zeros((100, 4), dtype=np.float32)
    raw = tmp_cfg / 'f.bin'
    arr.tofile(raw)
    with pytest.raises(NotImplementedError):
        mtscomp_amd.compress(raw, tmp_cfg / 'f.cbin', tmp_cfg / 'f.ch', sample_rate=100., n_channels=4, dtype=arr.dtype)


def test_full_size_chunk_properties():
    """BASELINE configs[1] chunk shape (385 ch x 30000): byte identity vs zlib for one chunk, and for a
    batch: round trip, chunk independence (batch == one by one), adler trailer == adler of the stream."""
    import zlib
    x = synth_int16(0, 3 * 30000, 385, 0)
    bounds = [0, 30000, 60000, 90000]
    flags = hip.make_flags(True, False, 'F')
    got = hip.compress_chunks(x, bounds, flags, 6)
    want0 = O.ref_compress_chunk(x[:30000])
    assert got[0] == want0
    single = hip.compress_chunks(x[30000:60000], [0, 30000], flags, 6)
    assert single[0] == got[1]
    for i in range(3):
        stream = zlib.decompress(got[i])
        assert stream == O.delta_transpose(x[bounds[i]:bounds[i + 1]], flags).tobytes()
    st, arrs = hip.decompress_chunks(got, [30000] * 3, 385, 'int16', flags)
    assert st == [0, 0, 0] and all(np.array_equal(arrs[i], x[bounds[i]:bounds[i + 1]]) for i in range(3))
    ratio = sum(map(len, got)) / x.nbytes
    assert 0.35 < ratio < 0.37
