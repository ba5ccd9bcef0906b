"""Host-side logic without a GPU: C-ABI surface, error behaviour, scheduler/sharding, config, Reader
semantics (driven through the test-only OracleCodec)."""
import os
import ctypes
import json
import re
from pathlib import Path

import numpy as np
import pytest

import mtscomp_amd
from mtscomp_amd import api, hip
from tests.codec_oracle import OracleCodec

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture
def tmp_cfg(tmp_path, monkeypatch):
    monkeypatch.setattr(api, 'CONFIG_PATH', tmp_path / '.mtscomp')
    return tmp_path


def test_cabi_library_exports_every_declared_symbol():
    header = (ROOT / 'include' / 'mtscomp_hip.h').read_text()
    declared = set(re.findall(r'\b(mts_[a-z0-9_]+)\s*\(', header))
    assert declared == set(hip.EXPORTS)
    L = ctypes.CDLL(str(hip.LIB_PATH))
    for sym in declared:
        assert hasattr(L, sym), sym
    assert hip.lib().mts_version() >= 100
    assert hip.compress_bound(23100000) == 23100000 + (23100000 >> 12) + (23100000 >> 14) + (23100000 >> 25) + 13
    assert hip.lib().mts_strerror(-2) == b'no usable gfx950 device'


@pytest.mark.skipif(Path('/dev/kfd').exists(), reason='needs a box without GPU')
def test_no_cpu_fallback_without_device(tmp_cfg):
    assert hip.device_count() == 0
    with pytest.raises(hip.HipError):
        hip.delta_transpose(np.zeros((4, 4), dtype=np.int16), 5)
    with pytest.raises(hip.HipError) as e:                       # the decoded-chunk cache lives in HBM or nowhere
        hip.cache_create(1 << 20)
    assert e.value.code == hip.E_NODEV
    api.set_codec(None)
    arr = np.zeros((100, 4), dtype=np.int16)
    arr.tofile(tmp_cfg / 'd.bin')
    with pytest.raises(hip.HipError):
        mtscomp_amd.compress(tmp_cfg / 'd.bin', sample_rate=100., n_channels=4, dtype='int16')
    # the product package must not reach for the oracle
    src = ''.join(p.read_text() for p in (ROOT / 'mtscomp_amd').glob('*.py'))
    assert 'oracle' not in src.replace('test-only', '').replace('CPU oracle here', '')


def test_config_defaults_and_merge(tmp_cfg):
    c = mtscomp_amd.read_config()
    assert c.check_after_compress and c.check_after_decompress and c.do_time_diff and not c.do_spatial_diff
    assert c.algorithm == 'zlib' and c.chunk_order == 'F' and c.comp_level == -1 and c.cache_size == 10
    mtscomp_amd.write_config(sample_rate=1234, chunk_duration=None, foo='bar')
    c = mtscomp_amd.read_config(n_channels=7)
    assert c.sample_rate == 1234 and c.chunk_duration == 1. and c.foo == 'bar' and c.n_channels == 7


def test_load_raw_data(tmp_cfg):
    p = tmp_cfg / 'x.bin'
    for arr in [np.zeros((0, 1)), np.zeros((10, 10)), (np.random.randn(100, 10) * 100).astype(np.int16)]:
        arr.tofile(p)
        for mmap in (True, False):
            got = mtscomp_amd.load_raw_data(p, n_channels=arr.shape[1], dtype=arr.dtype, mmap=mmap)
            assert np.array_equal(got, arr)
            del got
    with pytest.raises(ValueError):
        mtscomp_amd.load_raw_data(p, n_channels=7, dtype=np.int16)


def _write(tmp, arr, **kw):
    raw = tmp / 'data.bin'
    arr.tofile(raw)
    codec = kw.pop('codec', None) or OracleCodec()
    mtscomp_amd.compress(raw, tmp / 'data.cbin', tmp / 'data.ch', sample_rate=1234., n_channels=arr.shape[1],
                         dtype=arr.dtype, codec=codec, **kw)
    return mtscomp_amd.decompress(tmp / 'data.cbin', tmp / 'data.ch', codec=codec), codec


def test_errors_like_the_reference(tmp_cfg):
    arr = np.zeros((100, 4), dtype=np.int16)
    raw = tmp_cfg / 'd.bin'
    arr.tofile(raw)
    codec = OracleCodec()
    with pytest.raises(ValueError):
        mtscomp_amd.compress(raw, n_channels=4, dtype='int16', codec=codec)                 # no sample rate
    with pytest.raises(ValueError):
        mtscomp_amd.compress(raw, sample_rate=100., dtype='int16', codec=codec)             # no n_channels
    with pytest.raises(ValueError):
        mtscomp_amd.compress(raw, sample_rate=100., n_channels=4, codec=codec)              # no dtype
    with pytest.raises(ValueError):
        mtscomp_amd.compress(raw, sample_rate=100., n_channels=3, dtype='int16', codec=codec)   # bad size
    (tmp_cfg / 'e.bin').write_bytes(b'')
    with pytest.raises(Exception):
        mtscomp_amd.compress(tmp_cfg / 'e.bin', sample_rate=100., n_channels=4, dtype='int16', codec=codec)


def test_check_fail_raises_runtime_error(tmp_cfg):
    # tests.py:345-378: flip bytes of the raw file between write and check
    arr = (np.random.RandomState(0).randn(5000, 8) * 1000).astype(np.int16)
    raw = tmp_cfg / 'd.bin'
    arr.tofile(raw)

    def before_check(w):
        w.close()
        with open(raw, 'r+b') as f:
            f.seek(raw.stat().st_size // 2)
            f.write(b'\x55' * 8)
        w.open(raw, sample_rate=1234., n_channels=8, dtype=arr.dtype)
    w = mtscomp_amd.Writer(before_check=before_check, codec=OracleCodec())
    w.open(raw, sample_rate=1234., n_channels=8, dtype=arr.dtype)
    with pytest.raises(RuntimeError):
        w.write(tmp_cfg / 'd.cbin', tmp_cfg / 'd.ch')


def test_reader_indexing_matches_numpy(tmp_cfg):
    # tests.py:246-342
    arr = (np.random.RandomState(1).randn(6997, 19) * 3000).astype(np.int16)
    r, _ = _write(tmp_cfg, arr)
    N = arr.shape[0]
    rs = np.random.RandomState(2)
    items = [slice(a, b, c) for a in (None, 0, 1, -1) for b in (None, 0, 1, -1) for c in (None, 2, 3, N // 2, N)]
    items += [slice(*map(int, t)) for t in rs.randint(-100, 2 * N, size=(100, 3)) if t[2] > 0]
    items += [(slice(None),), (slice(None), slice(1, -1, 2)), (slice(None), [1, 5, 3]), (slice(None), 1),
              (1, slice(None)), (2, 1), 0, 1, N - 2, N - 1, -1, -N]
    items += rs.randint(-N, N, size=50).tolist()
    for s in items:
        got, want = r[s], arr[s]
        assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), s
    table = [(-1, 2, 0, 0), (0, 0, 0, 0), (1233, 1233, 0, 0), (1233, 1234, 0, 1), (1234, 1234, 1, 1),
             (1234, 1235, 1, 1), (-10000, 10000, 0, 5), (1234, 10000, 1, 5), (6996, 10000, 5, 5), (6998, 10000, 5, 5)]
    for i0, i1, c0, c1 in table:
        assert r._chunks_for_interval(i0, i1) == (c0, c1)
    r.close()


def test_batched_decode_and_cache(tmp_cfg):
    arr = (np.random.RandomState(3).randn(6997, 5) * 300).astype(np.int16)
    r, codec = _write(tmp_cfg, arr, check_after_compress=False)
    r.set_cache_size(2)
    codec.calls.clear()
    d = r.decompress_chunks([0, 1, 2])          # tests.py:413-430
    assert sorted(d) == [0, 1, 2] and codec.calls == [('decompress', 3)]
    assert len(r._cache) == 2
    d = r.decompress_chunks([1, 2, 3], pool=r.start_thread_pool())
    r.stop_thread_pool()
    assert sorted(d) == [1, 2, 3] and codec.calls[-1] == ('decompress', 1)      # 1, 2 came from the cache
    codec.calls.clear()
    x = r[1000:4000]                            # 3-4 chunks in ONE codec call
    assert np.array_equal(x, arr[1000:4000]) and len(codec.calls) == 1
    r.close()


def test_corrupt_and_wrong_size_map_to_reference_exceptions(tmp_cfg):
    arr = (np.random.RandomState(4).randn(3000, 5) * 300).astype(np.int16)
    r, codec = _write(tmp_cfg, arr)
    r.close()
    meta = json.loads((tmp_cfg / 'data.ch').read_text())
    b = bytearray((tmp_cfg / 'data.cbin').read_bytes())
    b[meta['chunk_offsets'][1] + 30] ^= 0xff
    (tmp_cfg / 'data.cbin').write_bytes(bytes(b))
    r = mtscomp_amd.decompress(tmp_cfg / 'data.cbin', tmp_cfg / 'data.ch', codec=codec)
    assert np.array_equal(r[:1000], arr[:1000])
    with pytest.raises(IOError, match='Compressed chunk #1 is corrupted'):
        r[1300:1400]
    r.close()
    meta['chunk_bounds'][-1] -= 1            # valid stream, wrong expected size -> AssertionError (mtscomp.py:628)
    r = mtscomp_amd.Reader(codec=codec)
    r.open(tmp_cfg / 'data.cbin', meta)
    with pytest.raises(AssertionError):
        r[2500:2600]
    r.close()


@pytest.mark.parametrize('n_devices', [1, 2, 3])
def test_round_robin_sharding_of_hipcodec(n_devices, monkeypatch):
    """HipCodec shards chunk i -> device i mod G and reassembles in order (host logic only: the device
    calls are replaced by the oracle)."""
    from oracle import oracle as O
    seen = []

    def fake_compress(data, bounds, flags, level=6, device=0):
        seen.append((device, len(bounds) - 1))
        return [O.compress_chunk(np.ascontiguousarray(data[bounds[i]:bounds[i + 1]]), flags, level)
                for i in range(len(bounds) - 1)]
    monkeypatch.setattr(hip, 'require_device', lambda: n_devices)
    monkeypatch.setattr(hip, 'compress_chunks', fake_compress)
    codec = api.HipCodec()
    assert codec.devices == list(range(n_devices))
    x = (np.random.RandomState(5).randn(700, 6) * 100).astype(np.int16)
    chunks = [x[i * 100:(i + 1) * 100] for i in range(7)]
    got = codec.compress(chunks, 5, 6)
    assert got == [O.compress_chunk(c, 5, 6) for c in chunks]
    assert sorted(seen) == sorted((d, len(range(d, 7, n_devices))) for d in range(n_devices))


def test_3d_npy_and_options(tmp_cfg):
    # tests.py:433-448
    array = np.random.RandomState(6).randint(-5000, 5000, size=(20, 12, 13), dtype=np.int16)
    p = tmp_cfg / 't.npy'
    np.save(p, array)
    codec = OracleCodec()
    mtscomp_amd.compress(p, out=tmp_cfg / 't.cnpy', outmeta=tmp_cfg / 't.ch', sample_rate=np.prod(array.shape[1:]),
                         dtype=array.dtype, do_time_diff=False, codec=codec)
    d = mtscomp_amd.decompress(tmp_cfg / 't.cnpy', cmeta=tmp_cfg / 't.ch', codec=codec)
    assert np.array_equal(d[:, :].reshape(d.cmeta.shape), array)
    d.close()


@pytest.mark.parametrize('chunk_duration', [.01, .1, 1., 10.])
def test_chunk_durations(tmp_cfg, chunk_duration):
    arr = (np.random.RandomState(7).randn(6997, 3) * 300).astype(np.int16)
    r, _ = _write(tmp_cfg, arr, chunk_duration=chunk_duration)
    assert np.array_equal(r[:], arr)
    r.close()


def test_long_slices_go_batch_by_batch_and_short_preads_are_completed(tmp_cfg, monkeypatch):
    arr = (np.random.RandomState(8).randn(6997, 7) * 900).astype(np.int16)
    r, codec = _write(tmp_cfg, arr, check_after_compress=False)          # 6 chunks
    r.batch_size = 2                                                     # slices over more than 2 chunks are batched
    codec.calls.clear()
    assert np.array_equal(r[:], arr)
    assert [c for c in codec.calls if c[0] == 'decompress'] == [('decompress', 2)] * 3
    for s in (slice(3, -5, 3), slice(1234, 6000), slice(100, 6997, 1000)):
        assert np.array_equal(r[s], arr[s]), s
    # os.pread may return fewer bytes than asked for (it does above 2 GiB): the reader asks again
    real = os.pread
    monkeypatch.setattr(os, 'pread', lambda fd, n, off: real(fd, min(n, 1000), off))
    r._cache.clear()
    assert np.array_equal(r[500:3000], arr[500:3000])
    assert np.array_equal(r[:], arr)
    r.close()


def test_host_cache_is_safe_under_concurrent_slicing(tmp_cfg):
    # the reference's Reader is sliced from ThreadPool workers (mtscomp.py:648); the LRU of decoded chunks is shared
    from multiprocessing.pool import ThreadPool
    arr = (np.random.RandomState(9).randn(6997, 5) * 500).astype(np.int16)
    r, _ = _write(tmp_cfg, arr, check_after_compress=False)
    r.set_cache_size(2)
    rs = np.random.RandomState(1)
    jobs = [(int(a), int(a + n)) for a, n in zip(rs.randint(0, 6000, size=200), rs.randint(1, 900, size=200))]
    with ThreadPool(8) as pool:
        got = pool.map(lambda ab: r[ab[0]:ab[1]], jobs)
    assert all(np.array_equal(g, arr[a:b]) for (a, b), g in zip(jobs, got))
    assert len(r._cache) <= 2
    r.close()


def test_slices_through_the_decoded_chunk_cache_interface(tmp_cfg, monkeypatch):
    """Reader slices against a codec that offers the cache interface (CPU stand-in for mts_cache_*): only missing chunks are
    read and sent, a miss after the query is retried with all bytes, corrupt chunks map to IOError with their index.
    (Read-ahead off: the calls are counted.)"""
    from tests.codec_oracle import CachingOracleCodec
    monkeypatch.setattr(api, 'READ_AHEAD_MAX', 0)
    arr = (np.random.RandomState(10).randn(6997, 6) * 700).astype(np.int16)
    codec = CachingOracleCodec(capacity_chunks=3)
    r, _ = _write(tmp_cfg, arr, check_after_compress=False, codec=codec)             # 6 chunks of 1234 rows
    codec.calls.clear()
    assert np.array_equal(r[100:1300], arr[100:1300])                                # chunks 0 and 1: both sent
    assert np.array_equal(r[1300:2500:3], arr[1300:2500:3])                          # chunks 1 and 2: only 2 is sent
    assert codec.calls == [('cache_read', 2), ('cache_read', 1)]
    assert np.array_equal(r[-10:], arr[-10:]) and np.array_equal(r[3000], arr[3000])
    codec.drop_before_next_read = True                                               # evicted between query and read
    assert np.array_equal(r[2600:2700, 1:4], arr[2600:2700, 1:4])
    assert codec.calls[-2:] == [('cache_slices', 0), ('cache_slices', 1)]            # first try without bytes, then with
    assert np.array_equal(r[:], arr)                                                 # all six chunks: within DEVICE_CACHE_MAX_CHUNKS
    b = bytearray((tmp_cfg / 'data.cbin').read_bytes())
    b[r.chunk_offsets[4] + 20] ^= 0xff
    (tmp_cfg / 'data.cbin').write_bytes(bytes(b))
    r2 = mtscomp_amd.decompress(tmp_cfg / 'data.cbin', tmp_cfg / 'data.ch', codec=CachingOracleCodec())
    with pytest.raises(IOError, match='#4'):
        r2[4936:4950]
    assert np.array_equal(r2[0:100], arr[0:100])
    # several rectangles in one codec call (Reader.read_slices): chunks 0, 2, 3 and 5 -- two separate reads of missing bytes
    codec.calls.clear()
    items = [(slice(10, 50), slice(1, 5)), (slice(2500, 4000, 7), slice(None, None, 2)), slice(6990, None), (slice(3000, 3010), 4),
             (slice(40, 40), slice(0, 3))]
    got = r.read_slices(items)
    assert len(got) == len(items) and all(np.array_equal(g, arr[it]) and g.shape == arr[it].shape for g, it in zip(got, items))
    assert [c[0] for c in codec.calls] == ['cache_slices']
    # what the gather does not serve goes through __getitem__ (numpy semantics kept)
    assert np.array_equal(r[100:200, ::-1], arr[100:200, ::-1]) and np.array_equal(r[100:200, [1, 3]], arr[100:200, [1, 3]])
    assert r[200:100:-1].shape == arr[200:100:-1][0:0].shape          # (like the reference: a negative row step gives nothing)
    r.close(); r2.close()
    assert codec.caches == {}


def test_cold_slices_read_ahead_into_the_decoded_chunk_cache(tmp_cfg, monkeypatch):
    """A slice with missing chunks has the chunks right behind it decoded in the same codec call: one at first, one more
    every time a chunk read ahead is used (a sequential reader gets to READ_AHEAD_MAX), none when the slice's chunks are all
    resident, never past the end of the file or beyond a resident chunk; a damaged chunk ahead does not fail the read that
    did not ask for it, and the one that does gets the reference's IOError; the slices are numpy's throughout."""
    from tests.codec_oracle import CachingOracleCodec
    monkeypatch.setattr(api, 'READ_AHEAD_MAX', 3)
    arr = (np.random.RandomState(13).randn(14000, 4) * 500).astype(np.int16)
    codec = CachingOracleCodec(capacity_chunks=64)
    r, _ = _write(tmp_cfg, arr, check_after_compress=False, codec=codec)             # 12 chunks of 1234 rows (the last: 426)
    assert r.n_chunks == 12
    codec.calls.clear()
    assert np.array_equal(r[10:20], arr[10:20])                                      # chunk 0, and chunk 1 ahead
    assert codec.calls == [('cache_read', 2)] and sorted(codec.caches[1]) == [0, 1]
    assert np.array_equal(r[1300:1310], arr[1300:1310])                              # chunk 1: resident, nothing is read; the read-ahead was used
    assert codec.calls[-1] == ('cache_read', 0) and r._ra == 2
    assert np.array_equal(r[2500:2510], arr[2500:2510])                              # chunk 2 missing: 3 and 4 ahead
    assert codec.calls[-1] == ('cache_read', 3) and sorted(codec.caches[1]) == [0, 1, 2, 3, 4]
    assert np.array_equal(r[3800:5000:7], arr[3800:5000:7]) and r._ra == 3             # chunks 3 and 4: used
    assert np.array_equal(r[8700:8710], arr[8700:8710])                              # chunk 7 missing: 8, 9, 10 ahead
    assert codec.calls[-1] == ('cache_read', 4)
    assert np.array_equal(r[6200:6300], arr[6200:6300])                              # chunk 5 missing: 6 ahead, 7 is resident
    assert codec.calls[-1] == ('cache_read', 2)
    assert np.array_equal(r[13900:], arr[13900:])                                    # the last chunk: nothing behind it
    assert codec.calls[-1] == ('cache_read', 1)
    r.close()
    # damage in chunk 3: reading chunk 2 (which reads 3 ahead) succeeds, reading chunk 3 raises
    b = bytearray((tmp_cfg / 'data.cbin').read_bytes())
    b[r.chunk_offsets[3] + 20] ^= 0xff
    (tmp_cfg / 'data.cbin').write_bytes(bytes(b))
    r2 = mtscomp_amd.decompress(tmp_cfg / 'data.cbin', tmp_cfg / 'data.ch', codec=CachingOracleCodec(capacity_chunks=64))
    assert np.array_equal(r2[2500:2510], arr[2500:2510])
    with pytest.raises(IOError, match='#3'):
        r2[3800:3810]
    assert np.array_equal(r2[5000:5010], arr[5000:5010])
    r2.close()


def test_trim_cache_copies_only_what_pins_a_big_buffer(tmp_cfg):
    """Decoded chunks of one codec call are views of one buffer: while the cached views account for most of that buffer they
    stay views (no host copy per chunk, neighbours keep a common base); what is left of a big batch is copied so that the
    batch's buffer can go."""
    arr = (np.random.RandomState(4).randn(6997, 5) * 300).astype(np.int16)
    r, _ = _write(tmp_cfg, arr, check_after_compress=False)
    r.set_cache_size(3)
    r._cache.clear()
    big = np.zeros((4000, 5), dtype=np.int16)
    views = [big[i * 1000:(i + 1) * 1000] for i in range(4)]
    for i in (0, 1, 2):
        r._cache[i] = views[i]
    r._trim_cache()
    assert all(r._cache[i].base is big for i in (0, 1, 2))           # 3 of 4 quarters cached: views stay
    r.set_cache_size(1)
    r._trim_cache()
    (idx, kept), = r._cache.items()
    assert idx == 2 and kept.base is None and np.array_equal(kept, views[2])      # one quarter of the buffer left: copied
    r.close()


def test_leading_channel_reads_send_prefixes_and_fall_back(tmp_cfg):
    """Reader[rows, :k] with a codec that decodes leading channels from a prefix (mts_cache_read_slices_leading): the Reader reads
    a share of every cold chunk's bytes, asks again with the whole chunks when an entry holds too few channels (or a prefix fell
    short), and the rows come out as numpy's."""
    from tests.codec_oracle import LeadingOracleCodec
    rng = np.random.RandomState(8)
    arr = np.cumsum(rng.randint(-40, 41, size=(6000, 200)), axis=0).astype(np.int16)
    raw = tmp_cfg / 'lead.bin'
    arr.tofile(raw)
    codec = LeadingOracleCodec()
    mtscomp_amd.compress(raw, tmp_cfg / 'lead.cbin', tmp_cfg / 'lead.ch', sample_rate=1000., n_channels=200, dtype=np.int16, codec=codec,
                         check_after_compress=False)
    # default settings: whole chunks are sent (and checked) whatever the columns, as the reference does (mtscomp.py:618-621)
    r = mtscomp_amd.decompress(tmp_cfg / 'lead.cbin', tmp_cfg / 'lead.ch', codec=codec)
    sizes = [r.chunk_offsets[i + 1] - r.chunk_offsets[i] for i in range(r.n_chunks)]
    assert not r.partial_decode
    codec.bytes_given.clear()
    assert np.array_equal(r[500:2500, 0:4], arr[500:2500, 0:4])
    assert codec.bytes_given == [sum(sizes[0:3])]
    r.close()
    codec = LeadingOracleCodec()
    r = mtscomp_amd.decompress(tmp_cfg / 'lead.cbin', tmp_cfg / 'lead.ch', codec=codec, partial_decode=True)
    codec.bytes_given.clear()
    assert np.array_equal(r[500:2500, 0:4], arr[500:2500, 0:4])                  # chunks 0..2, cold: prefixes
    assert len(codec.bytes_given) >= 1 and codec.bytes_given[-1] <= sum(sizes[0:3])
    first_call = codec.bytes_given[0]
    assert first_call < sum(sizes[0:3])                                          # (a share of the bytes + the margin, not all of them)
    assert np.array_equal(r[600:700, 1:4:2], arr[600:700, 1:4:2])                # resident
    n_calls = len(codec.bytes_given)
    assert np.array_equal(r[500:2500, 0:12], arr[500:2500, 0:12])                # more channels than the entries hold: the query says so, ONE call with longer prefixes
    assert len(codec.bytes_given) == n_calls + 1 and first_call < codec.bytes_given[-1] < sum(sizes[0:3])
    assert np.array_equal(r[0:6000:7, 199], arr[0:6000:7, 199])                  # the last channel: whole chunks
    assert np.array_equal(r[100:200, 0:150], arr[100:200, 0:150])                # more than half of the channels: whole chunks
    # an empty request keeps its own columns: the leading-channel count is taken over all requests
    got = r.read_slices([(slice(3000, 3100), slice(0, 4)), (slice(5, 5), slice(None))])
    assert np.array_equal(got[0], arr[3000:3100, 0:4]) and got[1].shape == (0, 200)
    r.close()


def test_tofile_overwrite_makes_a_new_file_like_the_reference(tmp_cfg):
    """tofile(overwrite=True) unlinks the old file and writes a NEW one (mtscomp.py:711-717): a hard link to the old file and a
    memory map of it keep the old bytes, a read-only old file in a writable directory is no obstacle, the new file has the
    recording.  A name that is not a regular file is replaced as well."""
    from tests.codec_oracle import LaneOracleCodec
    arr = (np.random.RandomState(12).randn(5000, 5) * 300).astype(np.int16)
    codec = LaneOracleCodec(n_lanes=1, capacity_chunks=8)
    r, _ = _write(tmp_cfg, arr, check_after_compress=False, codec=codec)
    r.close()
    r = mtscomp_amd.decompress(tmp_cfg / 'data.cbin', tmp_cfg / 'data.ch', codec=codec, check_after_decompress=False)
    assert r.n_chunks > 2
    back, link = tmp_cfg / 'back.bin', tmp_cfg / 'link.bin'
    for old_len in (arr.nbytes + 12345, 100, arr.nbytes):
        back.write_bytes(b'\xa5' * old_len)
        os.link(back, link)
        os.chmod(back, 0o444)
        held = np.memmap(back, dtype=np.uint8, mode='r')
        ino = back.stat().st_ino
        r.tofile(back, overwrite=True)
        assert back.stat().st_ino != ino and back.stat().st_size == arr.nbytes
        assert np.array_equal(np.fromfile(back, dtype=np.int16).reshape(-1, 5), arr)
        assert link.stat().st_ino == ino and link.read_bytes() == b'\xa5' * old_len      # the hard link still has the old file
        assert held.shape[0] == old_len and bool((held == 0xa5).all())                        # and so does the map
        del held
        link.unlink()
    back.unlink()
    back.symlink_to(tmp_cfg / 'elsewhere.bin')                       # (dangling or not: the name is what gets the file)
    r.tofile(back, overwrite=True) if back.exists() else r.tofile(back)
    assert np.array_equal(np.fromfile(back, dtype=np.int16).reshape(-1, 5), arr)
    r.close()


def test_tofile_in_place_is_an_opt_in(tmp_cfg, monkeypatch):
    """tofile(overwrite=True, in_place=True) -- or MTSCOMP_TOFILE_IN_PLACE=1 -- writes over an existing regular file where it is:
    same inode, a longer old file cut to the new length, a shorter one grown; not the reference's behaviour, hence not the default."""
    from tests.codec_oracle import LaneOracleCodec
    arr = (np.random.RandomState(13).randn(5000, 5) * 300).astype(np.int16)
    codec = LaneOracleCodec(n_lanes=1, capacity_chunks=8)
    r, _ = _write(tmp_cfg, arr, check_after_compress=False, codec=codec)
    r.close()
    r = mtscomp_amd.decompress(tmp_cfg / 'data.cbin', tmp_cfg / 'data.ch', codec=codec, check_after_decompress=False)
    back = tmp_cfg / 'back.bin'
    for how, old_len in (('arg', arr.nbytes + 12345), ('arg', 100), ('env', arr.nbytes)):
        back.write_bytes(b'\xa5' * old_len)
        ino = back.stat().st_ino
        if how == 'env':
            monkeypatch.setenv('MTSCOMP_TOFILE_IN_PLACE', '1')
            r.tofile(back, overwrite=True)
        else:
            r.tofile(back, overwrite=True, in_place=True)
        assert back.stat().st_ino == ino and back.stat().st_size == arr.nbytes
        assert np.array_equal(np.fromfile(back, dtype=np.int16).reshape(-1, 5), arr)
    r.close()


def test_tofile_failure_waits_for_its_writers(tmp_cfg):
    """A corrupt chunk in the middle of a pipelined tofile: IOError naming the chunk, an empty output file, and no helper thread
    of the call still alive afterwards (the writers are joined BEFORE the descriptor is truncated and closed and the buffers go
    back to the pool)."""
    import threading
    from tests.codec_oracle import LaneOracleCodec
    arr = (np.random.RandomState(14).randn(9000, 5) * 300).astype(np.int16)
    codec = LaneOracleCodec(n_lanes=1, capacity_chunks=8)
    r, _ = _write(tmp_cfg, arr, check_after_compress=False, codec=codec)
    r.close()
    cbin = tmp_cfg / 'data.cbin'
    blob = bytearray(cbin.read_bytes())
    r = mtscomp_amd.decompress(cbin, tmp_cfg / 'data.ch', codec=codec, check_after_decompress=False)
    bad = r.n_chunks - 2
    off = r.chunk_offsets[bad] + (r.chunk_offsets[bad + 1] - r.chunk_offsets[bad]) // 2
    r.close()
    blob[off] ^= 0x5a
    blob[off + 1] ^= 0xff
    cbin.write_bytes(bytes(blob))
    r = mtscomp_amd.decompress(cbin, tmp_cfg / 'data.ch', codec=codec, check_after_decompress=False)
    before = {t.ident for t in threading.enumerate()}
    monkey_piece = api.TOFILE_PIECE_CHUNKS
    api.TOFILE_PIECE_CHUNKS = 1
    try:
        with pytest.raises(IOError, match='chunk #%d' % bad):
            r.tofile(tmp_cfg / 'back.bin', overwrite=True)
    finally:
        api.TOFILE_PIECE_CHUNKS = monkey_piece
    assert (tmp_cfg / 'back.bin').stat().st_size == 0
    left = [t for t in threading.enumerate() if t.ident not in before and t.is_alive()]
    assert not left, left
    r.close()


def test_read_slices_limits_are_per_lane_and_lanes_decode_leading_channels(tmp_cfg, monkeypatch):
    """Two lanes.  (a) Requests whose chunks all live on ONE lane (every other chunk) beyond that lane's per-call limit go
    through the host path instead of overfilling the lane -- the limits are per lane, not lanes x limit.  (b) With
    partial_decode=True the lanes are told how many leading channels the requests need (they were handed None before: whole
    chunks): fewer compressed bytes cross than the chunks have."""
    from tests.codec_oracle import LaneOracleCodec
    arr = (np.random.RandomState(15).randn(9000, 40) * 700).astype(np.int16)
    codec = LaneOracleCodec(n_lanes=2, leading=True, capacity_chunks=8)
    r, _ = _write(tmp_cfg, arr, check_after_compress=False, codec=codec)             # 8 chunks of 1234 rows
    r.close()
    r = mtscomp_amd.decompress(tmp_cfg / 'data.cbin', tmp_cfg / 'data.ch', codec=codec, check_after_decompress=False, partial_decode=True)
    assert r.n_chunks == 8 and r._n_lanes() == 2
    sizes = [r.chunk_offsets[k + 1] - r.chunk_offsets[k] for k in range(8)]
    # (b) leading channels through two lanes
    items = [(slice(100, 200), slice(0, 4)), (slice(1300, 1400), slice(1, 3))]        # chunk 0 (lane 0), chunk 1 (lane 1)
    got = r.read_slices(items)
    assert all(np.array_equal(g, arr[it]) for g, it in zip(got, items))
    assert len(codec.bytes_given) == 2 and all(b < sizes[k] for b, k in zip(sorted(codec.bytes_given), (0, 1)) if b) and \
        sum(codec.bytes_given) < sizes[0] + sizes[1]
    # (a) chunks 0, 2, 4 are all lane 0's: with a limit of two chunks per lane and call the device gather declines
    monkeypatch.setattr(api, 'DEVICE_CACHE_MAX_CHUNKS', 2)
    items = [(slice(10, 20), slice(None)), (slice(2500, 2510), slice(None)), (slice(5000, 5010), slice(None))]
    assert r.read_slices(items, _fallback=False) is None
    got = r.read_slices(items)
    assert all(np.array_equal(g, arr[it]) for g, it in zip(got, items))
    # the same number of chunks spread over both lanes is served: 2 per lane
    items = [(slice(10, 20), slice(None)), (slice(1300, 1310), slice(None)), (slice(2500, 2510), slice(None)), (slice(3800, 3810), slice(None))]
    got = r.read_slices(items, _fallback=False)
    assert got is not None and all(np.array_equal(g, arr[it]) for g, it in zip(got, items))
    r.close()


@pytest.mark.parametrize('n_lanes', [2, 3])
def test_reader_over_several_lanes(tmp_cfg, n_lanes):
    """A codec with several lanes (HipCodec: one per device): chunk k is read, decoded and kept by lane k mod lanes only; slices,
    rectangles that cross chunks (row and column steps included) and tofile give what numpy gives; errors name the chunk."""
    from tests.codec_oracle import LaneOracleCodec
    arr = (np.random.RandomState(11).randn(9000, 7) * 700).astype(np.int16)
    codec = LaneOracleCodec(n_lanes=n_lanes, capacity_chunks=8)
    r, _ = _write(tmp_cfg, arr, check_after_compress=False, codec=codec)             # 8 chunks of 1234 rows (the last: 362)
    r.close()
    r = mtscomp_amd.decompress(tmp_cfg / 'data.cbin', tmp_cfg / 'data.ch', codec=codec, check_after_decompress=False)
    assert r.n_chunks == 8
    for a, b, step in ((100, 1300, None), (1300, 5000, 3), (8990, 9000, None), (0, 9000, 7), (2468, 3702, None)):
        assert np.array_equal(r[a:b:step], arr[a:b:step])
    for lane, keys in codec.lane_keys.items():
        assert keys and all(k % n_lanes == lane for k in keys)
    assert sorted(k for keys in codec.lane_keys.values() for k in keys) == list(range(8))
    items = [(slice(10, 50), slice(1, 5)), (slice(1200, 6000, 7), slice(None, None, 2)), slice(8990, None), (slice(3000, 3010), 4),
             (slice(40, 40), slice(0, 3)), (slice(1233, 1236), slice(0, 7)), (slice(0, 9000, 1234), 0)]
    got = r.read_slices(items)
    assert all(np.array_equal(g, arr[it]) and g.shape == arr[it].shape for g, it in zip(got, items))
    assert np.array_equal(r[5:8000:11, 2:6:3], arr[5:8000:11, 2:6:3])
    # tofile: the pieces go to the lanes in turn, every lane decodes straight into its own buffers
    monkey_piece = api.TOFILE_PIECE_CHUNKS
    api.TOFILE_PIECE_CHUNKS = 2
    try:
        r.tofile(tmp_cfg / 'back.bin', overwrite=True)
    finally:
        api.TOFILE_PIECE_CHUNKS = monkey_piece
    assert np.array_equal(np.fromfile(tmp_cfg / 'back.bin', dtype=np.int16).reshape(-1, 7), arr)
    assert sorted(codec.lane_calls) == sorted((k % n_lanes, 2) for k in range(4))
    # a long slice (more chunks than a batch): batches go to the lanes in turn, straight into the result
    r.batch_chunks = r.batch_size = 1
    codec.lane_calls.clear()
    api_max = api.DEVICE_CACHE_MAX_CHUNKS
    api.DEVICE_CACHE_MAX_CHUNKS = 1
    try:
        assert np.array_equal(r[600:8000], arr[600:8000])
    finally:
        api.DEVICE_CACHE_MAX_CHUNKS = api_max
    assert sorted(codec.lane_calls) == sorted((j % n_lanes, 1) for j in range(7))
    r.close()
    assert codec.caches == {}
    # damage in chunk 5: the slice that touches it raises with its index, from whichever lane found it
    b = bytearray((tmp_cfg / 'data.cbin').read_bytes())
    b[r.chunk_offsets[5] + 20] ^= 0xff
    (tmp_cfg / 'data.cbin').write_bytes(bytes(b))
    r2 = mtscomp_amd.decompress(tmp_cfg / 'data.cbin', tmp_cfg / 'data.ch', codec=LaneOracleCodec(n_lanes=n_lanes))
    with pytest.raises(IOError, match='#5'):
        r2[4000:8000]
    with pytest.raises(IOError, match='#5'):
        r2.read_slices([(slice(4000, 8000, 5), slice(0, 3))])
    assert np.array_equal(r2[0:3000], arr[0:3000])
    r2.close()


def test_a_failed_read_leaves_the_readers_pinned_buffer_unlocked(tmp_cfg, monkeypatch):
    """A short read of a truncated .cbin inside the page-locked read path raises the reference's AssertionError -- and the next
    slice on the same Reader gets the same answer instead of waiting for a lock nobody releases (round-4 advisor finding)."""
    from tests.codec_oracle import CachingOracleCodec

    class Pinned:
        def __init__(self, n):
            self.nbytes, self.ptr = n, 1
            self.array = np.zeros(n, dtype=np.uint8)

        def free(self):
            self.ptr = 0
    codec = CachingOracleCodec()
    codec.host_buffer = Pinned
    arr = (np.random.RandomState(12).randn(5000, 4) * 300).astype(np.int16)
    r, _ = _write(tmp_cfg, arr, check_after_compress=False, codec=codec)
    assert np.array_equal(r[10:20], arr[10:20])                      # (the pinned path works)
    size = (tmp_cfg / 'data.cbin').stat().st_size
    os.truncate(tmp_cfg / 'data.cbin', size - 100)
    for _ in range(2):
        with pytest.raises(AssertionError):
            r[4900:5000]
        assert not r._pin_lock.locked()
    r.close()
