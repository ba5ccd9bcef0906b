"""BASELINE.json's configurations at their stated sizes on the GPU box (what one GPU can run of them):

  configs[1]  385 ch x 60 s through mtscomp_amd.compress(): the WHOLE .cbin (all 60 chunks) against the reference's ThreadPool +
              zlib path restated (oracle.ref_compress_chunk), the .ch text as the reference would write it, the file back
              (reference contract: mtscomp.py:474-495, tests.py:403-410)
  configs[2]  600 s file, 1000 windows Reader[s:s+30000] at splitmix starts, EVERY window against the generator
              (mtscomp.py:798-856; LRU of decoded chunks :582-588 -- here the device cache, default and a small one)
  configs[3]  385 ch x 3600 s sharded over 8 GPUs: ONE rank's shard at its size -- chunks 0, 8, 16, ... = 450 chunks, 10.4 GB raw --
              through the device-resident entry points, full round trip, every 25th chunk against zlib.compress
              (mtscomp.py:399-423, :474-483; the other seven shards are the same code on other chunk ids)
  configs[4]  1024 ch, 0.25 s chunks, 240 chunks, levels 1 / 6 / 9: EVERY chunk against zlib.compress(stream, level); and ONE RANK'S SHARD of
              the 600 s / 2400-chunk form (300 chunks, 4.6 GB) at the three levels

The recordings live in HBM through the library's own allocator (hip.DevBuffer -> mts_dev_alloc / mts_dev_copy): this process
initialises one HIP runtime, whatever else it imports and in whatever order (test_import_order_* pins both orders with torch).
"""
import ctypes as C
import hashlib
import json
import os
import shutil
import tempfile
import zlib
from multiprocessing.dummy import Pool as ThreadPool
from pathlib import Path

import numpy as np
import pytest

import bench
import mtscomp_amd
from mtscomp_amd import hip
from oracle import oracle as O

pytestmark = pytest.mark.gpu
RATE = 30000

@pytest.fixture
def shm(monkeypatch):
    tmp = Path(tempfile.mkdtemp(prefix='mtstest_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None))
    monkeypatch.setenv('HOME', str(tmp))
    monkeypatch.setattr(mtscomp_amd.api, 'CONFIG_PATH', tmp / '.mtscomp', raising=False)
    yield tmp
    shutil.rmtree(tmp, ignore_errors=True)


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_long))


def _synth_host(t0, t1, nc):
    return bench.synth_host(hip, 0, t0, t1, nc)


def test_config1_whole_recording_byte_identical(shm):
    nc, seconds = 385, 60
    x = _synth_host(0, seconds * RATE, nc)
    raw, out, outmeta, back = shm / 'data.bin', shm / 'data.cbin', shm / 'data.ch', shm / 'back.bin'
    x.tofile(raw)
    ratio = mtscomp_amd.compress(raw, out, outmeta, sample_rate=float(RATE), n_channels=nc, dtype=np.int16)      # (check_after_compress on, like the default)
    # the reference's path on the host cores: every chunk through numpy diff / tobytes('F') / zlib.compress
    with ThreadPool(min(64, os.cpu_count() or 1)) as pool:
        want = pool.map(lambda i: O.ref_compress_chunk(x[i * RATE:(i + 1) * RATE]), range(seconds))
    cbin = out.read_bytes()
    offs = [0] + [int(v) for v in np.cumsum([len(c) for c in want])]
    got = [cbin[offs[i]:offs[i + 1]] for i in range(seconds)]
    assert len(cbin) == offs[-1]
    assert [i for i in range(seconds) if got[i] != want[i]] == []
    assert hashlib.sha1(cbin).hexdigest() == hashlib.sha1(b''.join(want)).hexdigest()
    expected_header = {
        'version': '1.0', 'algorithm': 'zlib', 'comp_level': -1, 'do_time_diff': True, 'do_spatial_diff': False, 'dtype': 'int16',
        'n_channels': nc, 'sample_rate': float(RATE), 'chunk_bounds': list(range(0, seconds * RATE + 1, RATE)), 'chunk_offsets': offs,
        'chunk_order': 'F', 'sha1_compressed': hashlib.sha1(cbin).hexdigest(), 'sha1_uncompressed': hashlib.sha1(x).hexdigest(),
        'shape': [seconds * RATE, nc]}
    assert outmeta.read_text() == json.dumps(expected_header, indent=2, sort_keys=True)      # the .ch as mtscomp.py:494-495 writes it
    assert abs(ratio - len(cbin) / x.nbytes) < 1e-12
    r = mtscomp_amd.decompress(out, outmeta, back)                                             # tests.py:403-410: file back, byte for byte
    r.close()
    assert hashlib.sha1(back.read_bytes()).hexdigest() == expected_header['sha1_uncompressed']


@pytest.mark.parametrize('cache_gb', [None, '1'])          # default (32 GiB: the whole decoded file stays in HBM) / 1 GiB (43 chunks: evicts all the time)
def test_config2_random_windows_600s(shm, monkeypatch, cache_gb):
    nc, seconds = 385, 600
    n_windows = 1000 if cache_gb is None else 150
    if cache_gb is not None:
        monkeypatch.setenv('MTSCOMP_DEVICE_CACHE_GB', cache_gb)
    n_samples, _ = bench.build_synth_file(hip, 0, seconds, shm, nc)
    r = mtscomp_amd.decompress(shm / 'data.cbin', shm / 'data.ch')
    assert r.shape == (n_samples, nc)
    chk, got_d = hip.DevBuffer(RATE * nc * 2), hip.DevBuffer(RATE * nc * 2)
    bad = []
    for k, s in enumerate(bench.window_starts(n_samples, n_windows)):
        got = r[s:s + RATE]
        hip.dev_synth_int16(chk, 0, s, s + RATE, nc, 0)
        if got.shape != (RATE, nc):
            bad.append((k, s))
            continue
        got_d.upload(got)
        if got_d.diff(chk)[0]:
            bad.append((k, s))
    assert bad == []
    # the same windows again from what is resident, a few by sha1 against the first read's generator
    for s in bench.window_starts(n_samples, 8):
        assert hashlib.sha1(r[s:s + RATE]).hexdigest() == hashlib.sha1(_synth_host(s, s + RATE, nc)).hexdigest()
    r.close()


def test_config3_one_rank_shard_of_the_3600s_recording():
    """configs[3]: rank 0's shard of the 8-GPU run at its real size: chunks 0, 8, 16, ... of the 3600 s recording."""
    nc, world, seconds = 385, 8, 3600
    ids = bench.shard_ids(0, world, seconds)
    n = len(ids)
    assert n == 450
    chunk_bytes = RATE * nc * 2
    raw = hip.DevBuffer(n * chunk_bytes)
    for k, g in enumerate(ids):
        hip.dev_synth_int16(raw, k * chunk_bytes, g * RATE, (g + 1) * RATE, nc, 0)
    cb = (hip.compress_bound(chunk_bytes) + 255) // 256 * 256
    cbuf, back = hip.DevBuffer(n * cb), hip.DevBuffer(n * chunk_bytes)
    bounds = np.arange(n + 1, dtype=np.int64) * RATE
    slots = np.arange(n, dtype=np.int64) * cb
    sizes = np.zeros(n, dtype=np.int64)
    nrows = np.full(n, RATE, dtype=np.int64)
    ooffs = np.arange(n, dtype=np.int64) * chunk_bytes
    status = np.zeros(n, dtype=np.int32)
    flags = hip.make_flags(True, False, 'F')
    hip.dev_compress_chunks(raw, nc, 2, bounds, flags, 6, cbuf, slots, sizes)
    assert (sizes > 0).all() and 0.30 < sizes.sum() / (n * chunk_bytes) < 0.42
    hip.dev_decompress_chunks(cbuf, slots, sizes, nrows, nc, 2, flags, back, ooffs, status)
    assert not status.any()
    assert back.diff(raw) == (0, -1)
    picks = list(range(0, n, 25))
    with ThreadPool(min(32, os.cpu_count() or 1)) as pool:
        want = pool.map(lambda k: O.ref_compress_chunk(raw.download(k * chunk_bytes, chunk_bytes, np.int16).reshape(RATE, nc)), picks)
    diff = [ids[k] for k, w in zip(picks, want) if cbuf.download(int(slots[k]), int(sizes[k])).tobytes() != w]
    assert diff == []
    # the host-side gather of the sizes (the path's only exchange) puts this shard's chunks where the file has them
    offsets = np.concatenate(([0], np.cumsum(sizes)))
    assert offsets[-1] == sizes.sum() and list(bench.gather_chunk_offsets(sizes, 0, 1)) == list(offsets)
    for b in (raw, cbuf, back):
        b.free()


def test_config4_one_rank_shard_of_the_600s_recording():
    """configs[4]: rank 0's shard of the 8-GPU stress run at its real size -- 1024 ch, chunk = 0.25 s, 600 s = 2400 chunks, of which this
    rank owns chunks 0, 8, 16, ...: 300 chunks of 15.36 MB -- at levels 1, 6 and 9: full round trip on the device, every 20th chunk of
    the shard against zlib.compress(stream, level)."""
    nc, rows, world = 1024, 7500, 8
    ids = bench.shard_ids(0, world, 2400)
    n = len(ids)
    assert n == 300
    chunk_bytes = rows * nc * 2
    raw = hip.DevBuffer(n * chunk_bytes)
    for k, g in enumerate(ids):
        hip.dev_synth_int16(raw, k * chunk_bytes, g * rows, (g + 1) * rows, nc, 0)
    cb = (hip.compress_bound(chunk_bytes) + 255) // 256 * 256
    cbuf, back = hip.DevBuffer(n * cb), hip.DevBuffer(n * chunk_bytes)
    bounds = np.arange(n + 1, dtype=np.int64) * rows
    slots = np.arange(n, dtype=np.int64) * cb
    sizes = np.zeros(n, dtype=np.int64)
    nrows = np.full(n, rows, dtype=np.int64)
    ooffs = np.arange(n, dtype=np.int64) * chunk_bytes
    status = np.zeros(n, dtype=np.int32)
    flags = hip.make_flags(True, False, 'F')
    picks = list(range(0, n, 20))
    with ThreadPool(min(32, os.cpu_count() or 1)) as pool:
        streams = pool.map(lambda k: O.delta_transpose(raw.download(k * chunk_bytes, chunk_bytes, np.int16).reshape(rows, nc), flags).tobytes(), picks)
        for level in (1, 6, 9):
            hip.dev_compress_chunks(raw, nc, 2, bounds, flags, level, cbuf, slots, sizes)
            want = pool.map(lambda j: zlib.compress(streams[j], level), range(len(picks)))
            diff = [ids[k] for k, w in zip(picks, want) if cbuf.download(int(slots[k]), int(sizes[k])).tobytes() != w]
            assert diff == [], (level, diff)
            hip.dev_decompress_chunks(cbuf, slots, sizes, nrows, nc, 2, flags, back, ooffs, status)
            assert not status.any()
            assert back.diff(raw) == (0, -1), level
    for b in (raw, cbuf, back):
        b.free()


def test_config4_stress_shape_every_chunk_levels_1_6_9():
    nc, rows, n = 1024, 7500, 240
    raw = hip.DevBuffer(n * rows * nc * 2)
    hip.dev_synth_int16(raw, 0, 0, n * rows, nc, 0)
    cb = (hip.compress_bound(rows * nc * 2) + 255) // 256 * 256
    cbuf, back = hip.DevBuffer(n * cb), hip.DevBuffer(raw.nbytes)
    bounds = np.arange(n + 1, dtype=np.int64) * rows
    slots = np.arange(n, dtype=np.int64) * cb
    sizes = np.zeros(n, dtype=np.int64)
    nrows = np.full(n, rows, dtype=np.int64)
    ooffs = np.arange(n, dtype=np.int64) * rows * nc * 2
    status = np.zeros(n, dtype=np.int32)
    flags = hip.make_flags(True, False, 'F')
    x = raw.download(dtype=np.int16).reshape(n * rows, nc)
    zeros = np.zeros(1 << 24, dtype=np.uint8)
    with ThreadPool(min(64, os.cpu_count() or 1)) as pool:
        streams = pool.map(lambda k: O.delta_transpose(x[k * rows:(k + 1) * rows], flags).tobytes(), range(n))
        for level in (1, 6, 9):
            hip.dev_compress_chunks(raw, nc, 2, bounds, flags, level, cbuf, slots, sizes)
            host = cbuf.download()
            want = pool.map(lambda k: zlib.compress(streams[k], level), range(n))
            diff = [k for k in range(n) if host[k * cb:k * cb + int(sizes[k])].tobytes() != want[k]]
            assert diff == [], (level, diff[:8])
            for o in range(0, back.nbytes, zeros.nbytes):                       # (what a decode that wrote nothing would leave)
                back.upload(zeros[:min(zeros.nbytes, back.nbytes - o)], o)
            hip.dev_decompress_chunks(cbuf, slots, sizes, nrows, nc, 2, flags, back, ooffs, status)
            assert not status.any()
            assert back.diff(raw) == (0, -1), level


_ORDER_CHILD = r'''
import sys
sys.path.insert(0, %r)
%s
import numpy as np
from mtscomp_amd.synth import synth_int16
x = synth_int16(0, 900, 64, 0)
got = hip.compress_chunks(x, [0, 900], hip.make_flags(), 6, device=0)
st, arrs = hip.decompress_chunks(got, [900], 64, 'int16', hip.make_flags(), device=0)
assert st == [0] and np.array_equal(arrs[0], x)
b = hip.DevBuffer(x.nbytes); b.upload(x); assert np.array_equal(b.download(dtype=np.int16).reshape(x.shape), x)
maps = open('/proc/self/maps').read()
print('RUNTIMES', len({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}))
print('OK')
'''


@pytest.mark.parametrize('order', ['torch_first', 'library_first'])
def test_import_order_with_torch(order):
    """`import mtscomp_amd` before or after `import torch`: the codec works either way.  torch ships its own HIP runtime libraries;
    the package never asks torch for device memory or streams, so whichever runtime torch loaded is not the one that opens the device
    for these kernels -- unless torch came first, in which case the loader gives both the same one."""
    import subprocess
    import sys
    first = 'import torch\nfrom mtscomp_amd import hip\nhip.lib()' if order == 'torch_first' else 'from mtscomp_amd import hip\nhip.lib()\nimport torch'
    first += '\nimport torch.distributed  # what bench.py uses torch for'
    r = subprocess.run([sys.executable, '-c', _ORDER_CHILD % (str(Path(__file__).resolve().parent.parent), first)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'OK' in r.stdout, (r.stdout[-400:], r.stderr[-800:])
    if order == 'torch_first':
        assert 'RUNTIMES 1' in r.stdout
