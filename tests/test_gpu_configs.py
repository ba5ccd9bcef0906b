"""BASELINE.json's configurations at their stated sizes on the GPU box (what one GPU can run of them):

  configs[1]  385 ch x 60 s through mtscomp_amd.compress(): the WHOLE .cbin (all 60 chunks) against the reference's ThreadPool +
              zlib path restated (oracle.ref_compress_chunk), the .ch text as the reference would write it, the file back
              (reference contract: mtscomp.py:474-495, tests.py:403-410)
  configs[2]  600 s file, 1000 windows Reader[s:s+30000] at splitmix starts, EVERY window against the generator
              (mtscomp.py:798-856; LRU of decoded chunks :582-588 -- here the device cache, default and a small one)
  configs[4]  1024 ch, 0.25 s chunks, 240 chunks, levels 1 / 6 / 9: EVERY chunk against zlib.compress(stream, level)
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

# These tests hold their recordings in torch tensors.  torch brings its own HIP / HSA runtime libraries: they must be the first
# ones this process initialises (a second runtime opening the device after libmtscomp_hip.so's finds no GPU), so the device is
# initialised here, at collection time, before any test has called into the library.
if os.path.exists('/dev/kfd'):
    import torch
    torch.cuda.init()


@pytest.fixture
def shm(monkeypatch):
    tmp = Path(tempfile.mkdtemp(prefix='mtstest_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None))
    monkeypatch.setenv('HOME', str(tmp))
    monkeypatch.setattr(mtscomp_amd.api, 'CONFIG_PATH', tmp / '.mtscomp', raising=False)
    yield tmp
    shutil.rmtree(tmp, ignore_errors=True)


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_long))


def _synth_host(torch, t0, t1, nc):
    buf = torch.empty((t1 - t0, nc), dtype=torch.int16, device='cuda')
    assert hip.lib().mts_dev_synth_int16(0, None, C.c_void_p(buf.data_ptr()), t0, t1, nc, 0) == 0
    torch.cuda.synchronize()
    return buf.cpu().numpy()


def test_config1_whole_recording_byte_identical(shm):
    nc, seconds = 385, 60
    x = np.concatenate([_synth_host(torch, s * RATE, (s + 1) * RATE, nc) for s in range(seconds)])
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
    L = hip.lib()
    n_samples, _ = bench.build_synth_file(torch, hip, L, 0, seconds, shm, nc)
    r = mtscomp_amd.decompress(shm / 'data.cbin', shm / 'data.ch')
    assert r.shape == (n_samples, nc)
    chk = torch.empty((RATE, nc), dtype=torch.int16, device='cuda')
    bad = []
    for k, s in enumerate(bench.window_starts(n_samples, n_windows)):
        got = r[s:s + RATE]
        assert L.mts_dev_synth_int16(0, None, C.c_void_p(chk.data_ptr()), s, s + RATE, nc, 0) == 0
        if got.shape != (RATE, nc) or not torch.equal(torch.from_numpy(got).cuda(), chk):
            bad.append((k, s))
    assert bad == []
    # the same windows again from what is resident, a few by sha1 against the first read's generator
    for s in bench.window_starts(n_samples, 8):
        assert L.mts_dev_synth_int16(0, None, C.c_void_p(chk.data_ptr()), s, s + RATE, nc, 0) == 0
        assert hashlib.sha1(r[s:s + RATE]).hexdigest() == hashlib.sha1(chk.cpu().numpy()).hexdigest()
    r.close()


def test_config4_stress_shape_every_chunk_levels_1_6_9():
    nc, rows, n = 1024, 7500, 240
    L = hip.lib()
    raw = torch.empty((n * rows, nc), dtype=torch.int16, device='cuda')
    for k in range(n):
        assert L.mts_dev_synth_int16(0, None, C.c_void_p(raw[k * rows:].data_ptr()), k * rows, (k + 1) * rows, nc, 0) == 0
    cb = (hip.compress_bound(rows * nc * 2) + 255) // 256 * 256
    cbuf = torch.empty(n * cb, dtype=torch.uint8, device='cuda')
    back = torch.empty_like(raw)
    bounds = np.arange(n + 1, dtype=np.int64) * rows
    slots = np.arange(n, dtype=np.int64) * cb
    sizes = np.zeros(n, dtype=np.int64)
    nrows = np.full(n, rows, dtype=np.int64)
    ooffs = np.arange(n, dtype=np.int64) * rows * nc * 2
    status = np.zeros(n, dtype=np.int32)
    flags = hip.make_flags(True, False, 'F')
    x = raw.cpu().numpy()
    with ThreadPool(min(64, os.cpu_count() or 1)) as pool:
        streams = pool.map(lambda k: O.delta_transpose(x[k * rows:(k + 1) * rows], flags).tobytes(), range(n))
        for level in (1, 6, 9):
            rc = L.mts_dev_compress_chunks(0, None, C.c_void_p(raw.data_ptr()), nc, 2, _lp(bounds), n, flags, level, C.c_void_p(cbuf.data_ptr()),
                                           _lp(slots), _lp(sizes))
            assert rc == 0, L.mts_last_error()
            host = cbuf.cpu().numpy()
            want = pool.map(lambda k: zlib.compress(streams[k], level), range(n))
            diff = [k for k in range(n) if host[k * cb:k * cb + int(sizes[k])].tobytes() != want[k]]
            assert diff == [], (level, diff[:8])
            back.zero_()
            rc = L.mts_dev_decompress_chunks(0, None, C.c_void_p(cbuf.data_ptr()), _lp(slots), _lp(sizes), _lp(nrows), n, nc, 2, flags,
                                             C.c_void_p(back.data_ptr()), _lp(ooffs), status.ctypes.data_as(C.POINTER(C.c_int)))
            assert rc == 0 and not status.any()
            assert torch.equal(back, raw), level
