"""GPU parity at the API level: the host layer + HipCodec against the golden fixtures (files the reference
wrote), the reference's own property tests re-expressed, and size-independent properties at full size."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest
from zlib import compress as zlib_compress

import mtscomp_amd
from mtscomp_amd import api, hip
from mtscomp_amd.synth import synth_int16
from oracle import oracle as O
from tests.test_golden import CASES, golden_cbin, make_input, parse_slice, same_as_reference_decode, sha1

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
    assert same_as_reference_decode(r[:], case, arr)
    for s in case['slices']:
        v = r[parse_slice(s['s'])]
        assert list(v.shape) == s['shape'] and sha1(np.ascontiguousarray(v).tobytes()) == s['sha1'], s['s']
    r.close()


def test_reads_reference_written_files(tmp_cfg):
    for name in ('ar1_8ch_3chunks', 'uniform_random_stored', 'tiny_chunks', 'both_diffs_order_c', 'int32', 'float32', 'float64_spatial'):
        case = CASES[name]
        out = tmp_cfg / (name + '.cbin')
        out.write_bytes(golden_cbin(case))
        out.with_suffix('.ch').write_text(case['ch_text'])
        r = mtscomp_amd.decompress(out)
        assert same_as_reference_decode(r[:], case, make_input(case)), name
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


def test_wrong_size_chunk_gets_the_references_verdict(tmp_cfg, monkeypatch):
    """mtscomp.py:618-628: zlib.decompress (and its data check) comes first, the size assert after it.  A chunk that inflates to
    another size than the header says is an AssertionError when its stream is whole and an IOError when it is damaged."""
    import zlib
    case = CASES['ar1_8ch_3chunks']
    meta = json.loads(case['ch_text'])
    cb = golden_cbin(case)
    offs = meta['chunk_offsets']
    arr = make_input(case)
    other = zlib.compress(zlib.decompress(cb[offs[1]:offs[2]])[:-16])            # a row short
    damaged = other[:-2] + bytes([other[-2] ^ 0x40]) + other[-1:]                 # ... and its check value broken
    for k, (chunk1, exc) in enumerate(((other, AssertionError), (damaged, IOError))):
        m = dict(meta)
        m['chunk_offsets'] = [offs[0], offs[1], offs[1] + len(chunk1), offs[1] + len(chunk1) + offs[3] - offs[2]]
        out = tmp_cfg / ('odd%d.cbin' % k)
        out.write_bytes(cb[:offs[1]] + chunk1 + cb[offs[2]:])
        out.with_suffix('.ch').write_text(json.dumps(m))
        for cache_gb in ('0', '1'):                                               # host path and device cache
            monkeypatch.setenv('MTSCOMP_DEVICE_CACHE_GB', cache_gb)
            r = mtscomp_amd.decompress(out)
            assert np.array_equal(r[2000:3000], arr[2000:3000])
            with pytest.raises(exc):
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


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_float_transforms_bit_identical_to_numpy(dtype):
    """np.diff / np.cumsum on float items (tests.py:190-205, :212-237 run the reference on float arrays): every flag
    combination, with -0.0, inf, denormals and a wide range of magnitudes, compared bit for bit with numpy."""
    import zlib
    r = np.random.RandomState(4)
    for fl in range(8):
        x = (r.randn(3001, 37) * 10 ** r.uniform(-3, 6, size=(3001, 37))).astype(dtype)
        x[0, 0] = -0.0; x[5, 3] = np.inf; x[7, 7] = 1e-42 if dtype == 'float32' else 1e-310; x[100:110, 5] = 0
        td, sd, order = bool(fl & 1), bool(fl & 2), 'F' if fl & 4 else 'C'
        want = O.ref_diff_along_axis(O.ref_diff_along_axis(x, 0 if td else None), 1 if sd else None).tobytes(order=order)
        assert hip.delta_transpose(x, fl).tobytes() == want, (dtype, fl)
        with np.errstate(all='ignore'):
            back = O.ref_decompress_chunk(zlib.compress(want, 1), 3001, 37, dtype, td, sd, order)
        assert hip.cumsum_transpose(want, 3001, 37, dtype, fl).tobytes() == back.tobytes(), (dtype, fl)
        z = hip.compress_chunks(x, [0, 1000, 3001], fl, 6)
        assert z[1] == O.ref_compress_chunk(x[1000:], td, sd, order, 6)
        st, arrs = hip.decompress_chunks(z, [1000, 2001], 37, dtype, fl)
        with np.errstate(all='ignore'):
            assert st == [0, 0] and arrs[1].tobytes() == O.ref_decompress_chunk(z[1], 2001, 37, dtype, td, sd, order).tobytes()


def test_float_files_like_the_reference_tests(tmp_cfg):
    # tests.py:212-237 (test_low / test_high on float arrays): np.allclose after the round trip
    t = np.linspace(0., 10., 20000)[:, np.newaxis]
    for arr in (np.zeros((20000, 19), dtype=np.float32), np.sin(10 * t) + np.random.RandomState(0).normal(0, .02, size=(20000, 19))):
        raw = tmp_cfg / 'f.bin'
        arr.tofile(raw)
        mtscomp_amd.compress(raw, tmp_cfg / 'f.cbin', tmp_cfg / 'f.ch', sample_rate=20000., n_channels=19, dtype=arr.dtype)
        r = mtscomp_amd.decompress(tmp_cfg / 'f.cbin', tmp_cfg / 'f.ch')
        assert r[:].dtype == arr.dtype and np.allclose(arr, r[:])
        assert np.allclose(arr[5000:5100], r[5000:5100])
        r.close()
        for f in ('f.cbin', 'f.ch'):
            (tmp_cfg / f).unlink()


def test_float_dtype_fails_loudly(tmp_cfg):
    arr = np.zeros((100, 4), dtype=np.float16)
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


def test_one_chunk_with_more_blocks_than_the_chain_walk_holds():
    """A single chunk of 4 s (92 MB, ~1280 deflate blocks): more candidates than the inflate chain walk keeps in LDS (1024), so
    the walk takes its older path, 64 candidates at a time; byte identity with zlib and the round trip, next to a chunk of the
    usual size in the same batch (which takes the LDS path)."""
    import zlib
    x = synth_int16(0, 5 * 30000, 385, 3)
    bounds = [0, 4 * 30000, 5 * 30000]
    flags = hip.make_flags(True, False, 'F')
    got = hip.compress_chunks(x, bounds, flags, 6)
    assert got[0] == zlib.compress(O.delta_transpose(x[:120000], flags).tobytes(), 6)
    assert got[1] == O.ref_compress_chunk(x[120000:])
    st, arrs = hip.decompress_chunks(got, [120000, 30000], 385, 'int16', flags)
    assert st == [0, 0] and np.array_equal(arrs[0], x[:120000]) and np.array_equal(arrs[1], x[120000:])
    bad = bytearray(got[0])
    bad[len(bad) // 2] ^= 0x10
    st, _ = hip.decompress_chunks([bytes(bad), got[1]], [120000, 30000], 385, 'int16', flags)
    assert st[0] != 0 and st[1] == 0


def test_stress_shape_1024_channels_levels():
    """BASELINE configs[4] chunk shape (1024 ch x 7500 rows, chunk = 0.25 s): byte identity vs zlib at levels 1, 2, 3
    (deflate_fast), 6 and 9 through the C ABI's level parameter (the Python API, like the reference, always uses 6), round trip."""
    import zlib
    x = synth_int16(0, 2 * 7500, 1024, 5)
    bounds = [0, 7500, 15000]
    flags = hip.make_flags(True, False, 'F')
    stream0 = O.delta_transpose(x[:7500], flags).tobytes()
    stream1 = O.delta_transpose(x[7500:], flags).tobytes()
    for level in (1, 2, 3, 6, 9):
        got = hip.compress_chunks(x, bounds, flags, level)
        assert got[0] == zlib.compress(stream0, level), level
        assert got[1] == zlib.compress(stream1, level), level
        st, arrs = hip.decompress_chunks(got, [7500, 7500], 1024, 'int16', flags)
        assert st == [0, 0] and np.array_equal(arrs[0], x[:7500]) and np.array_equal(arrs[1], x[7500:])
    with pytest.raises(hip.HipError):
        hip.compress_chunks(x, bounds, flags, 0)
    with pytest.raises(hip.HipError):
        hip.compress_chunks(x, bounds, flags, 10)


def test_wide_batch_with_tiny_chunks():
    """1024 int16 channels (row tiles of 16) and chunks of 1, 2 and 63 rows next to one of 7500: the inverse transform's
    per-tile sums once overran their scratch buffer here (found by tools/fuzz_gpu.py)."""
    rows = [2, 1, 63, 1, 7500]
    b = np.concatenate(([0], np.cumsum(rows)))
    x = synth_int16(0, int(b[-1]), 1024, 9)
    flags = hip.make_flags(True, False, 'F')
    z = hip.compress_chunks(x, b, flags, 6)
    assert all(z[i] == O.ref_compress_chunk(x[b[i]:b[i + 1]]) for i in range(4))
    st, arrs = hip.decompress_chunks(z, rows, 1024, 'int16', flags)
    assert st == [0] * 5 and all(np.array_equal(arrs[i], x[b[i]:b[i + 1]]) for i in range(5))


def test_host_copies_whose_thread_shares_are_whole_pages():
    """Copies of 8 MB and more cross the bus through pinned pieces filled (and emptied) by several host threads.  With a size
    whose eighth is a whole number of 4 KiB pages plus a remainder, the shares once ended `size % 8` bytes short of the size: the
    last bytes of the raw data, of the compressed range and of the decoded rows were never copied (found by tools/fuzz_gpu.py as a
    last chunk that would not decode, once in ~4000 such calls)."""
    n = 8 * 4096 * 300 + 6                                        # 9 830 406 bytes: floor(n / 8) is a multiple of 4096, n % 8 = 6
    r = np.random.RandomState(5)
    x = np.tile(r.randint(0, 256, size=4099).astype(np.uint8), n // 4099 + 1)[:n].reshape(-1, 1).copy()
    x[-8:, 0] = np.arange(1, 9)                                   # a tail that says when it is missing
    flags = hip.make_flags(False, False, 'C')
    z = hip.compress_chunks(x, np.array([0, n]), flags, 6)        # raw bytes host -> device
    assert z[0] == O.ref_compress_chunk(x, False, False, 'C', 6)
    st, arrs = hip.decompress_chunks(z, [n], 1, 'uint8', flags)   # decoded bytes device -> host
    assert st == [0] and np.array_equal(arrs[0], x)
    # a compressed range of such a size host -> device: stored chunks (random bytes), the last one small
    rnd = r.randint(0, 256, size=n + 70000).astype(np.uint8)
    sizes = []
    m = n
    for k in range(40):                                           # the stored stream of k bytes is k + 5 per 65535 + 6: find the size that gives n
        zs = [zlib_compress(rnd[:m].tobytes()), zlib_compress(rnd[m:m + 3000].tobytes())]
        tot = len(zs[0]) + len(zs[1])
        if tot == n:
            break
        m -= tot - n
    assert tot == n, tot
    st, arrs = hip.decompress_chunks(zs, [m, 3000], 1, 'uint8', flags)
    assert st == [0, 0] and arrs[0].tobytes() == rnd[:m].tobytes() and arrs[1].tobytes() == rnd[m:m + 3000].tobytes()


def test_more_chunks_than_a_grid_dimension():
    """70 000 chunks of one row in one call: several kernels take the chunk from blockIdx.y (at most 65 535), so the entry
    points split such a call; levels 6 and 1, and back."""
    import zlib
    n = 70000
    x = synth_int16(0, n, 3, 4)
    b = np.arange(n + 1)
    flags = hip.make_flags(True, False, 'F')
    for level in (6, 1):
        z = hip.compress_chunks(x, b, flags, level)
        assert len(z) == n
        for i in (0, 1, 32767, 32768, 65535, 65536, n - 1):
            assert z[i] == zlib.compress(x[i:i + 1].tobytes(), level), (level, i)
        st, arrs = hip.decompress_chunks(z, [1] * n, 3, 'int16', flags)
        assert not any(st)
        assert np.array_equal(np.concatenate([a for a in arrs]), x)


def test_dependent_copy_chains_do_not_starve_the_resolver():
    """int64 items of small magnitude, no differencing: the stream is literal + 7-byte match at distance 8, each match
    reading what the one before it wrote.  Fourteen resolver waves polling the LDS for their sources used to leave the
    one wave that could progress so little of it that the bounded waits expired (a false 'corrupt', after seconds)."""
    import time
    r = np.random.RandomState(5)
    x = (np.sin(np.arange(7500)[:, None] / 9.) * 1000 + r.randn(7500, 1024)).astype(np.int64)
    z = hip.compress_chunks(x, [0, 7500], 0, 6)
    assert z[0] == O.ref_compress_chunk(x, False, False, 'C')
    t = time.perf_counter()
    st, arrs = hip.decompress_chunks(z, [7500], 1024, 'int64', 0)
    assert st == [0] and np.array_equal(arrs[0], x)
    assert time.perf_counter() - t < 2.0


# ---- decoded-chunk cache in HBM (Reader slices) ----------------------------------------------------
def _write_recording(tmp, nt=30000, nc=16, chunk=0.1, rate=10000., seed=7):
    arr = synth_int16(0, nt, nc, seed)
    raw = tmp / 'rec.bin'
    arr.tofile(raw)
    out, outmeta = tmp / 'rec.cbin', tmp / 'rec.ch'
    mtscomp_amd.compress(raw, out, outmeta, sample_rate=rate, n_channels=nc, dtype=arr.dtype, chunk_duration=chunk)
    return arr, out, outmeta


@pytest.mark.parametrize('cache_gb', ['8', '0.0001', '0'])           # roomy, 105 KB (three decoded chunks: evicts all the time), off
def test_slices_through_device_cache(cache_gb, tmp_cfg, monkeypatch):
    monkeypatch.setenv('MTSCOMP_DEVICE_CACHE_GB', cache_gb)
    arr, out, outmeta = _write_recording(tmp_cfg)                  # 30 chunks of 1000 rows (32 KB decoded each)
    r = mtscomp_amd.decompress(out, outmeta)
    rng = np.random.RandomState(0)
    for _ in range(60):
        a = int(rng.randint(0, arr.shape[0] - 1))
        b = int(min(arr.shape[0], a + rng.randint(1, 3500)))
        step = [None, 1, 2, 7][rng.randint(0, 4)]
        got = r[a:b:step]
        assert got.dtype == arr.dtype and np.array_equal(got, arr[a:b:step]), (a, b, step)
    assert np.array_equal(r[-5:], arr[-5:]) and np.array_equal(r[12345], arr[12345])
    assert np.array_equal(r[999:1001, 3:9], arr[999:1001, 3:9])
    assert np.array_equal(r[:], arr)                               # long slices take the streaming path
    if float(cache_gb) > 0:
        assert r._dev_cache is not None
        if cache_gb == '8':
            assert (hip.cache_query(r._dev_cache, list(range(30))) > 0).sum() >= 25     # what was touched stays resident
        else:
            assert 1 <= (hip.cache_query(r._dev_cache, list(range(30))) > 0).sum() <= 3
    else:
        assert r._dev_cache is None
    r.close()
    assert r._dev_cache is None


def test_device_cache_corrupt_chunk_and_short_last(tmp_cfg):
    arr, out, outmeta = _write_recording(tmp_cfg, nt=10450)        # 10 chunks of 1000 rows + one of 450
    meta = json.loads(outmeta.read_text())
    b = bytearray(out.read_bytes())
    b[meta['chunk_offsets'][4] + 30] ^= 0x55
    out.write_bytes(bytes(b))
    r = mtscomp_amd.decompress(out, outmeta)
    assert np.array_equal(r[9500:10450], arr[9500:10450])
    assert np.array_equal(r[3000:4000], arr[3000:4000])
    with pytest.raises(IOError, match='#4'):
        r[3990:4010]
    assert np.array_equal(r[3990:4000], arr[3990:4000])            # its neighbour stayed resident and readable
    assert np.array_equal(r[5000:5010], arr[5000:5010])
    r.close()


def test_cache_c_abi_miss_and_eviction():
    x = synth_int16(0, 4000, 8, 3)
    bounds = np.arange(5) * 1000
    z = hip.compress_chunks(x, bounds, hip.make_flags(), 6)
    flags = hip.make_flags()
    cid = hip.cache_create(3 * 16384)                              # room for three decoded chunks (16 000 B each, 4 KiB granules)
    try:
        with pytest.raises(hip.HipError) as e:                     # not resident, no bytes
            hip.cache_read_rows(cid, [0], b'', [0], [0], [1000], 8, np.int16, flags, 0, 10)
        assert e.value.code == hip.E_MISS
        buf = b''.join(z)
        offs = np.concatenate(([0], np.cumsum([len(c) for c in z])))[:-1]
        lens = [len(c) for c in z]
        st, rows = hip.cache_read_rows(cid, [0, 1, 2], buf, offs[:3], lens[:3], [1000] * 3, 8, np.int16, flags, 500, 2500)
        assert st == [0, 0, 0] and np.array_equal(rows, x[500:2500])
        assert list(hip.cache_query(cid, [0, 1, 2, 3]) > 0) == [True, True, True, False]
        st, rows = hip.cache_read_rows(cid, [1, 2], b'', [0, 0], [0, 0], [1000] * 2, 8, np.int16, flags, 0, 2000)   # from HBM only
        assert np.array_equal(rows, x[1000:3000])
        st, rows = hip.cache_read_rows(cid, [3], buf, [offs[3]], [lens[3]], [1000], 8, np.int16, flags, 0, 1000)
        assert np.array_equal(rows, x[3000:4000])
        assert list(hip.cache_query(cid, [0, 1, 2, 3]) > 0) == [False, True, True, True]                                # LRU: chunk 0 went
        st, rows = hip.cache_read_rows(cid, [0], buf, [offs[0]], [lens[0]], [999], 8, np.int16, flags, 0, 999)     # wrong row count
        assert st == [hip.CHUNK_BADSIZE]
    finally:
        hip.cache_destroy(cid)
    with pytest.raises(hip.HipError):
        hip.cache_query(cid, [0])


def test_device_cache_concurrent_readers(tmp_cfg):
    # the reference's Reader is used from ThreadPool workers (mtscomp.py:648): slices from several threads at once
    from multiprocessing.pool import ThreadPool
    arr, out, outmeta = _write_recording(tmp_cfg, nt=20000)
    r = mtscomp_amd.decompress(out, outmeta)
    rng = np.random.RandomState(5)
    jobs = [(int(a), int(a + n)) for a, n in zip(rng.randint(0, 19000, size=64), rng.randint(1, 1000, size=64))]
    with ThreadPool(8) as pool:
        got = pool.map(lambda ab: r[ab[0]:ab[1]], jobs)
    for (a, b), g in zip(jobs, got):
        assert np.array_equal(g, arr[a:b]), (a, b)
    r.close()


def test_long_slices_decode_into_the_result(tmp_cfg):
    arr, out, outmeta = _write_recording(tmp_cfg, nt=30450)          # 30 chunks of 1000 rows + one of 450
    r = mtscomp_amd.decompress(out, outmeta)
    r.batch_size = 4                                                 # slices over more than 4 chunks: batch by batch, in place
    assert np.array_equal(r[:], arr)
    for s in (slice(5, -7, 3), slice(999, 30001), slice(12345, 30450), slice(0, 30450, 1000)):
        assert np.array_equal(r[s], arr[s]), s
    assert len(r._cache) == 0                                        # nothing of it went through the chunk cache
    meta = json.loads(outmeta.read_text())
    b = bytearray(out.read_bytes())
    b[meta['chunk_offsets'][17] + 25] ^= 0x10
    out.write_bytes(bytes(b))
    r2 = mtscomp_amd.decompress(out, outmeta)
    r2.batch_size = 4
    with pytest.raises(IOError, match='#17'):
        r2[:]
    assert np.array_equal(r2[0:16000], arr[0:16000])
    r.close(); r2.close()


def test_two_shards_on_one_device_match_the_goldens(tmp_cfg):
    # HipCodec drives its devices from one host thread each (api.py: run_lanes); with devices=[0, 0] the two shard
    # threads meet on one engine: round-robin sharding, the engine lock and the gather of the streams in chunk order
    api.set_codec(api.HipCodec(devices=[0, 0]))
    try:
        for name in ('ar1_8ch_3chunks', 'tiny_chunks', 'ar1_8ch_short_last'):
            case = CASES[name]
            arr = make_input(case)
            raw = tmp_cfg / (name + '.bin')
            arr.tofile(raw)
            out = tmp_cfg / (name + '.cbin')
            mtscomp_amd.compress(raw, out, out.with_suffix('.ch'), sample_rate=case['sample_rate'], n_channels=arr.shape[1],
                                 dtype=arr.dtype, **case['kwargs'])
            assert sha1(out.read_bytes()) == case['cbin_sha1'], name
            assert out.with_suffix('.ch').read_text() == case['ch_text'], name
            r = mtscomp_amd.decompress(out)
            assert same_as_reference_decode(r[:], case, arr), name
            r.close()
    finally:
        api.set_codec(None)


def test_reader_and_tofile_over_two_lanes_of_one_device(tmp_cfg, monkeypatch):
    # the in-process multi-device paths on the one device a test box has: HipCodec(devices=[0, 0]) has two lanes -- two decoded-chunk
    # caches (chunk k in cache k mod 2), two tofile pipelines with their own page-locked buffers, long slices batch by batch on
    # alternating lanes -- which meet on one engine.  Against the goldens' decode, numpy indexing and the file the Writer read.
    codec = api.HipCodec(devices=[0, 0])
    api.set_codec(codec)
    made = []
    real_create = hip.cache_create
    monkeypatch.setattr(hip, 'cache_create', lambda cap, device=0: made.append(real_create(cap, device=device)) or made[-1])
    try:
        arr = synth_int16(0, 30450, 96, 21)                            # 21 chunks of 1450 rows
        raw, out, outmeta, back = tmp_cfg / 'x.bin', tmp_cfg / 'x.cbin', tmp_cfg / 'x.ch', tmp_cfg / 'back.bin'
        arr.tofile(raw)
        mtscomp_amd.compress(raw, out, outmeta, sample_rate=1450., n_channels=96, dtype=arr.dtype, check_after_compress=False)
        want = [O.ref_compress_chunk(arr[k * 1450:(k + 1) * 1450]) for k in range(21)]
        assert out.read_bytes() == b''.join(want)
        r = mtscomp_amd.decompress(out, outmeta, check_after_decompress=False)
        for s_ in (slice(100, 1300), slice(1400, 4400, 3), slice(30440, None), slice(2900, 4350), slice(0, 11600, 7)):
            assert np.array_equal(r[s_], arr[s_]), s_
        assert len(made) == 2                                        # one cache per lane ...
        held = [set(np.nonzero(hip.cache_query(c, list(range(21))))[0].tolist()) for c in made]
        assert held[0] and held[1] and all(k % 2 == 0 for k in held[0]) and all(k % 2 == 1 for k in held[1])      # ... chunk k in cache k mod 2
        items = [(slice(10, 50), slice(1, 5)), (slice(1200, 9000, 7), slice(None, None, 2)), slice(30400, None), (slice(3000, 3010), 4),
                 (slice(40, 40), slice(0, 3)), (slice(1449, 1452), slice(0, 96)), (slice(0, 30450, 1450), 0)]
        got = r.read_slices(items)
        assert all(np.array_equal(g, arr[it]) and g.shape == arr[it].shape for g, it in zip(got, items))
        monkeypatch.setattr(api, 'TOFILE_PIECE_CHUNKS', 3)
        r.tofile(back, overwrite=True)                               # 7 pieces, lanes in turn
        assert back.read_bytes() == raw.read_bytes()
        r.tofile(back, overwrite=True)                               # once more over the existing file (the buffers come from the pool)
        assert back.read_bytes() == raw.read_bytes()
        r.batch_chunks = r.batch_size = 4
        assert np.array_equal(r[700:29000], arr[700:29000])          # long slice: batches of 4 chunks on alternating lanes
        # decompress() of a list / a range with two lanes: shards of one call
        st, arrs = codec.decompress(want, [1450] * 21, 96, np.int16, hip.make_flags(True, False, 'F'))
        assert st == [0] * 21 and all(np.array_equal(a, arr[k * 1450:(k + 1) * 1450]) for k, a in enumerate(arrs))
        b = bytearray(out.read_bytes())
        b[r.chunk_offsets[13] + 40] ^= 0x20
        out.write_bytes(bytes(b))
        r2 = mtscomp_amd.decompress(out, outmeta, check_after_decompress=False)
        with pytest.raises(IOError, match='#13'):
            r2[17400:20300]
        with pytest.raises(IOError, match='#13'):
            r2.tofile(back, overwrite=True)
        assert np.array_equal(r2[0:5000], arr[0:5000])
        r.close(); r2.close()
    finally:
        api.set_codec(None)
        codec.close()


def test_concurrent_calls_into_one_engine():
    # four host threads inside mts_compress_chunks / mts_decompress_chunks of the same device at once
    from multiprocessing.dummy import Pool
    flags = hip.make_flags(True, False, 'F')
    xs = [synth_int16(1000 * k, 1000 * k + 2400, 48, k) for k in range(4)]
    bounds = [0, 1000, 2000, 2400]
    want = [[O.ref_compress_chunk(x[bounds[i]:bounds[i + 1]]) for i in range(3)] for x in xs]

    def work(k):
        for _ in range(3):
            got = hip.compress_chunks(xs[k], bounds, flags, 6, device=0)
            assert got == want[k], k
            st, arrs = hip.decompress_chunks(got, [1000, 1000, 400], 48, 'int16', flags, device=0)
            assert st == [0, 0, 0] and all(np.array_equal(arrs[i], xs[k][bounds[i]:bounds[i + 1]]) for i in range(3))
        return True
    with Pool(4) as pool:
        assert all(pool.map(work, range(4)))


def test_sort_order_guard_repeats_the_stage(monkeypatch):
    # the hash sort's fast ranking relies on a hardware property; the match stage checks what it relies on (positions
    # ascending inside a hash run) and the stage is repeated with the ballot ranking when that fails.  The hook damages
    # the first sort of every call: the streams must still be zlib's, for both match kernels (levels 6 and 9)
    x = synth_int16(0, 3000, 16, 3)
    flags = hip.make_flags(True, False, 'F')
    want6, want9 = O.compress_chunk(x, flags, 6), O.compress_chunk(x, flags, 9)
    monkeypatch.setenv('MTS_SORT_INJECT_DISORDER', '1')
    assert hip.compress_chunks(x, [0, 3000], flags, 6, device=0)[0] == want6
    assert hip.compress_chunks(x, [0, 3000], flags, 9, device=0)[0] == want9
    assert 'hash_sort_retry' in [n for n, _ in hip.last_stage_times(0)]
    monkeypatch.delenv('MTS_SORT_INJECT_DISORDER')
    assert hip.compress_chunks(x, [0, 3000], flags, 6, device=0)[0] == want6
    assert 'hash_sort_retry' not in [n for n, _ in hip.last_stage_times(0)]


def test_read_slices_gathers_on_the_device(tmp_cfg):
    """Reader[rows, columns] and Reader.read_slices on a 385-channel file with 23.1 MB chunks (BASELINE configs[2] shape): the
    rows and columns are picked on the device (mts_cache_read_slices) and must be numpy's."""
    arr = synth_int16(0, 4 * 30000, 385, 2)
    raw = tmp_cfg / 'np.bin'
    arr.tofile(raw)
    mtscomp_amd.compress(raw, tmp_cfg / 'np.cbin', tmp_cfg / 'np.ch', sample_rate=30000., n_channels=385, dtype=np.int16,
                         check_after_compress=False)
    r = mtscomp_amd.decompress(tmp_cfg / 'np.cbin', tmp_cfg / 'np.ch')
    items = [(slice(100, 30100), slice(0, 385)), (slice(29990, 60010, 3), slice(10, 300, 7)), (slice(119000, None), 384),
             (slice(45000, 45001), slice(None)), slice(60000, 60010), (slice(5, 5), slice(2, 9))]
    got = r.read_slices(items)
    for g, it in zip(got, items):
        want = arr[it]
        assert g.shape == want.shape and g.dtype == want.dtype and np.array_equal(g, want), it
    assert np.array_equal(r[31000:32000, 5:50:5], arr[31000:32000, 5:50:5])
    assert np.array_equal(r[31000:32000, -1], arr[31000:32000, -1])
    assert np.array_equal(r[1000:90000:1000, :], arr[1000:90000:1000, :])
    # other dtypes / itemsizes through the same gather
    for dtype in ('uint8', 'int32', 'int64'):
        a2 = np.random.RandomState(3).randint(0, 200, size=(5000, 7)).astype(dtype)
        p = tmp_cfg / ('g_%s.bin' % dtype)
        a2.tofile(p)
        mtscomp_amd.compress(p, p.with_suffix('.cbin'), p.with_suffix('.ch'), sample_rate=1000., n_channels=7, dtype=a2.dtype,
                             check_after_compress=False)
        r2 = mtscomp_amd.decompress(p.with_suffix('.cbin'), p.with_suffix('.ch'))
        assert np.array_equal(r2[10:4000:3, 1:6:2], a2[10:4000:3, 1:6:2]) and np.array_equal(r2[:, 6], a2[:, 6])
        r2.close()
    r.close()


def test_release_then_same_shape_again():
    """mts_release() frees the workspaces; the next batch of the SAME shape must upload its tile / segment / block
    descriptors again (they are cached per allocation, not per address: a freed buffer usually comes back where it was)."""
    x = synth_int16(0, 6000, 24, 11)
    bounds = [0, 1500, 3000, 4500, 6000]
    flags = hip.make_flags()
    first = hip.compress_chunks(x, bounds, flags, 6)
    again = hip.compress_chunks(x, bounds, flags, 6)                 # (descriptors reused)
    hip.lib().mts_release()
    after = hip.compress_chunks(x, bounds, flags, 6)
    want = [O.ref_compress_chunk(x[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]
    assert first == want and again == want and after == want
    hip.lib().mts_release()
    y = synth_int16(0, 6000, 24, 12)                                 # same shape, other bytes, fresh buffers
    assert hip.compress_chunks(y, bounds, flags, 6) == [O.ref_compress_chunk(y[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]
    st, arrs = hip.decompress_chunks(after, [1500] * 4, 24, 'int16', flags)
    assert st == [0] * 4 and np.array_equal(np.concatenate(arrs), x)


def test_read_slices_argument_checks_and_many_requests():
    """mts_cache_read_slices: a key listed twice and an output offset that is not a multiple of the item size are refused;
    more requests than one grid dimension holds (65535) are served in slabs."""
    import ctypes as C
    x = synth_int16(0, 3000, 8, 5)
    z = hip.compress_chunks(x, [0, 1000, 2000, 3000], hip.make_flags(), 6)
    flags = hip.make_flags()
    buf = b''.join(z)
    lens = [len(c) for c in z]
    offs = list(np.concatenate(([0], np.cumsum(lens)))[:-1])
    cid = hip.cache_create(1 << 20)
    try:
        with pytest.raises(hip.HipError) as e:
            hip.cache_read_slices(cid, [0, 1, 0], buf, [offs[0], offs[1], offs[0]], [lens[0], lens[1], lens[0]], [1000] * 3, 8, np.int16, flags,
                                  [(0, 10, 1, 0, 8, 1)])
        assert e.value.code == -1
        n_req = 70000
        rng = np.random.RandomState(1)
        r0 = rng.randint(0, 2990, size=n_req)
        c0 = rng.randint(0, 7, size=n_req)
        reqs = [(int(a), int(a) + 3, 1, int(c), int(c) + 2, 1) for a, c in zip(r0, c0)]
        st, got = hip.cache_read_slices(cid, [0, 1, 2], buf, offs, lens, [1000] * 3, 8, np.int16, flags, reqs)
        assert st == [0, 0, 0] and len(got) == n_req
        for k in (0, 1, 65534, 65535, 65536, n_req - 1):
            assert np.array_equal(got[k], x[r0[k]:r0[k] + 3, c0[k]:c0[k] + 2]), k
        # a misaligned output offset straight through the C ABI
        L = hip.lib()
        keys = np.array([0], dtype=np.int64)
        zero = np.zeros(1, dtype=np.int64)
        rows = np.array([1000], dtype=np.int64)
        req = np.array([0, 4, 1, 0, 8, 1], dtype=np.int64)
        out = np.zeros(256, dtype=np.uint8)
        status = np.zeros(1, dtype=np.int32)
        lp = lambda a: a.ctypes.data_as(C.POINTER(C.c_long))  # noqa: E731
        for off, want_rc in ((1, -1), (2, 0)):
            ooff = np.array([off], dtype=np.int64)
            rc = L.mts_cache_read_slices(cid, 1, lp(keys), out.ctypes.data_as(C.c_void_p), lp(zero), lp(zero), lp(rows), 8, 2, flags, 1, lp(req),
                                         out.ctypes.data_as(C.c_void_p), lp(ooff), 200, status.ctypes.data_as(C.POINTER(C.c_int)))
            assert rc == want_rc, (off, rc)
        assert np.array_equal(out[2:2 + 64].view(np.int16).reshape(4, 8), x[0:4])
    finally:
        hip.cache_destroy(cid)


def test_leading_channels_are_decoded_from_a_prefix(tmp_cfg):
    """Reader[rows, :k] of a cold file: the chunks are inflated only as far as the first k channels reach (channel-major streams:
    a prefix), from a prefix of their compressed bytes; more channels or whole rows later decode the chunk again."""
    arr = synth_int16(0, 3 * 30000, 385, 4)
    raw = tmp_cfg / 'lead.bin'
    arr.tofile(raw)
    mtscomp_amd.compress(raw, tmp_cfg / 'lead.cbin', tmp_cfg / 'lead.ch', sample_rate=30000., n_channels=385, dtype=np.int16,
                         check_after_compress=False)
    r = mtscomp_amd.decompress(tmp_cfg / 'lead.cbin', tmp_cfg / 'lead.ch', partial_decode=True)
    reads = []
    orig = r._pread
    r._pread = lambda length, start: (reads.append(length), orig(length, start))[1]
    assert np.array_equal(r[100:45000, 0:32], arr[100:45000, 0:32])            # chunks 0 and 1, cold: prefixes only
    full = [r.chunk_offsets[i + 1] - r.chunk_offsets[i] for i in range(3)]
    assert len(reads) == 2 and all(n < f // 4 for n, f in zip(reads, full))
    del reads[:]
    assert np.array_equal(r[200:300, 5:30:5], arr[200:300, 5:30:5]) and reads == []      # resident (leading 32 channels)
    assert np.array_equal(r[200:40000, 0:64], arr[200:40000, 0:64])            # more channels than the entries hold: decoded again
    assert len(reads) >= 1                                                     # (one read of both chunks, whole)
    assert np.array_equal(r[29990:30010], arr[29990:30010])                    # whole rows: whole chunks
    assert np.array_equal(r[60000:60010, 380:385], arr[60000:60010, 380:385])  # trailing channels: whole chunk
    got = r.read_slices([(slice(0, 90000, 1000), slice(0, 8)), (slice(45000, 45010), slice(2, 3))])
    assert np.array_equal(got[0], arr[0:90000:1000, 0:8]) and np.array_equal(got[1], arr[45000:45010, 2:3])
    got = r.read_slices([(slice(0, 100), slice(0, 4)), (slice(5, 5), slice(None))])      # an empty request with all columns
    assert np.array_equal(got[0], arr[0:100, 0:4]) and got[1].shape == (0, 385)
    r.close()
    # C ABI: a prefix that is too short is a miss, not an error and not a wrong answer
    z = hip.compress_chunks(arr[:30000], [0, 30000], hip.make_flags(), 6)[0]
    cid = hip.cache_create(1 << 30)
    try:
        with pytest.raises(hip.HipError) as e:
            hip.cache_read_slices(cid, [0], z[:200000], [0], [200000], [30000], 385, np.int16, hip.make_flags(), [(0, 10, 1, 0, 32, 1)], n_leading=32)
        assert e.value.code == hip.E_MISS
        st, got = hip.cache_read_slices(cid, [0], z[:len(z) // 8], [0], [len(z) // 8], [30000], 385, np.int16, hip.make_flags(),
                                        [(10, 20, 1, 0, 32, 1)], n_leading=32)
        assert st == [0] and np.array_equal(got[0], arr[10:20, 0:32])
        with pytest.raises(hip.HipError):                                      # a request beyond the leading channels
            hip.cache_read_slices(cid, [0], b'', [0], [0], [30000], 385, np.int16, hip.make_flags(), [(10, 20, 1, 0, 33, 1)], n_leading=32)
        # damage in the part of the stream the prefix does not reach goes unnoticed by the partial read and is caught by a whole one
        bad = bytearray(z)
        bad[len(z) - 1000] ^= 0x40
        st, got = hip.cache_read_slices(cid, [7], bytes(bad), [0], [len(bad)], [30000], 385, np.int16, hip.make_flags(), [(10, 20, 1, 0, 32, 1)], n_leading=32)
        assert st == [0] and np.array_equal(got[0], arr[10:20, 0:32])
        st, _ = hip.cache_read_slices(cid, [8], bytes(bad), [0], [len(bad)], [30000], 385, np.int16, hip.make_flags(), [(10, 20, 1, 0, 32, 1)])
        assert st == [hip.CHUNK_CORRUPT]
    finally:
        hip.cache_destroy(cid)


def test_partial_reads_check_the_whole_chunk_by_default(tmp_cfg):
    """mtscomp.py:618-621: every read inflates -- and adler32-checks -- the whole chunk.  Default settings: damage BEHIND the prefix
    the first 32 channels need makes r[a:b, :32] raise; with partial_decode=True (the caller's explicit choice) damage INSIDE the prefix
    is refused too (the whole-chunk retry gives the reference's verdict), and a stream the prefix decoder cannot follow (Z_FIXED:
    fixed Huffman blocks only) is read correctly through that retry."""
    import zlib
    arr = synth_int16(0, 2 * 30000, 385, 9)
    raw = tmp_cfg / 'pc.bin'
    arr.tofile(raw)
    mtscomp_amd.compress(raw, tmp_cfg / 'pc.cbin', tmp_cfg / 'pc.ch', sample_rate=30000., n_channels=385, dtype=np.int16,
                         check_after_compress=False)
    hdr = json.loads((tmp_cfg / 'pc.ch').read_text())
    offs = hdr['chunk_offsets']
    data = bytearray((tmp_cfg / 'pc.cbin').read_bytes())
    behind = bytearray(data)
    behind[offs[1] - 1000] ^= 0x40                                  # the tail of chunk 0: far behind the first 32 channels
    (tmp_cfg / 'behind.cbin').write_bytes(behind)
    r = mtscomp_amd.decompress(tmp_cfg / 'behind.cbin', tmp_cfg / 'pc.ch')
    with pytest.raises(IOError):
        r[100:200, :32]
    assert np.array_equal(r[30100:30200, :32], arr[30100:30200, :32])      # the other chunk stays readable
    r.close()
    inside = bytearray(data)
    for k in range(2000, 2064):
        inside[k] ^= 0xff                                           # inside the first channels' part of chunk 0
    (tmp_cfg / 'inside.cbin').write_bytes(inside)
    r = mtscomp_amd.decompress(tmp_cfg / 'inside.cbin', tmp_cfg / 'pc.ch', partial_decode=True)
    with pytest.raises(IOError):
        r[100:200, :32]
    r.close()
    # a chunk of fixed-Huffman blocks only (another encoder's output): the prefix decoder declines, the whole-chunk retry reads it
    from oracle import oracle as O
    flags = hip.make_flags()
    zs = []
    for i in range(2):
        stream = O.delta_transpose(arr[i * 30000:(i + 1) * 30000], flags)
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        zs.append(c.compress(bytes(stream)) + c.flush())
    (tmp_cfg / 'fixed.cbin').write_bytes(b''.join(zs))
    hdr['chunk_offsets'] = [0, len(zs[0]), len(zs[0]) + len(zs[1])]
    (tmp_cfg / 'fixed.ch').write_text(json.dumps(hdr))
    r = mtscomp_amd.decompress(tmp_cfg / 'fixed.cbin', tmp_cfg / 'fixed.ch', partial_decode=True)
    assert np.array_equal(r[100:200, :32], arr[100:200, :32])
    r.close()
