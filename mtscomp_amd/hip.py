"""ctypes binding of libmtscomp_hip.so (include/mtscomp_hip.h).

This is the only compute path of the package: there is no CPU fallback.  Loading fails loudly when
the shared library has not been built (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C mtscomp_amd/csrc``) and every compute call raises ``HipError`` when no MI355X is visible.
"""
import ctypes as C
import os
import warnings
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent

# (MTSCOMP_HIP_LIB: another build of the same library, for A/B measurements -- tools/ab_stage_times.py; said out loud when used)
LIB_PATH = Path(os.environ.get('MTSCOMP_HIP_LIB') or _HERE / 'libmtscomp_hip.so')
if os.environ.get('MTSCOMP_HIP_LIB'):
    warnings.warn('mtscomp_amd: MTSCOMP_HIP_LIB replaces the in-tree library with %s' % LIB_PATH, RuntimeWarning, stacklevel=2)

FLAG_TIME_DIFF = 1
FLAG_SPATIAL_DIFF = 2
FLAG_ORDER_F = 4
FLAG_FLOAT = 8

CHUNK_OK = 0
CHUNK_CORRUPT = -1
CHUNK_BADSIZE = -2

E_NODEV = -2
E_UNSUPPORTED = -5
E_MISS = -7


class HipError(RuntimeError):
    def __init__(self, code, what, detail=''):
        self.code = code
        super().__init__('%s failed: %s (%d)%s' % (what, _strerror(code), code, (': ' + detail) if detail else ''))


_lib = None


def _strerror(code):
    try:
        return lib().mts_strerror(code).decode()
    except Exception:  # pragma: no cover
        return '?'


def lib():
    """The loaded shared library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            '%s is missing: build the HIP extension first (make -C mtscomp_amd/csrc). '
            'mtscomp_amd has no CPU code path.' % LIB_PATH)
    L = C.CDLL(str(LIB_PATH))
    vp, lp, ip = C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_int)
    L.mts_version.restype = C.c_int
    L.mts_device_count.restype = C.c_int
    L.mts_strerror.restype = C.c_char_p
    L.mts_strerror.argtypes = [C.c_int]
    L.mts_last_error.restype = C.c_char_p
    L.mts_compress_bound.restype = C.c_long
    L.mts_compress_bound.argtypes = [C.c_long]
    L.mts_delta_transpose.argtypes = [C.c_int, vp, C.c_long, C.c_int, C.c_int, C.c_int, vp]
    L.mts_cumsum_transpose.argtypes = [C.c_int, vp, C.c_long, C.c_int, C.c_int, C.c_int, vp]
    L.mts_compress_chunks.argtypes = [C.c_int, vp, C.c_int, C.c_int, lp, C.c_int, C.c_int, C.c_int, vp, lp, lp]
    L.mts_decompress_chunks.argtypes = [C.c_int, vp, lp, lp, lp, C.c_int, C.c_int, C.c_int, C.c_int, vp, lp, ip]
    L.mts_dev_compress_chunks.argtypes = [C.c_int, vp, vp, C.c_int, C.c_int, lp, C.c_int, C.c_int, C.c_int, vp, lp, lp]
    L.mts_dev_decompress_chunks.argtypes = [C.c_int, vp, vp, lp, lp, lp, C.c_int, C.c_int, C.c_int, C.c_int, vp, lp, ip]
    L.mts_dev_synth_int16.argtypes = [C.c_int, vp, vp, C.c_long, C.c_long, C.c_int, C.c_long]
    L.mts_host_alloc.argtypes = [C.c_long, C.POINTER(vp)]
    L.mts_host_free.argtypes = [vp]
    L.mts_dev_alloc.argtypes = [C.c_int, C.c_long, C.POINTER(vp)]
    L.mts_dev_free.argtypes = [C.c_int, vp]
    L.mts_dev_copy.argtypes = [C.c_int, vp, vp, vp, C.c_long, C.c_int]
    L.mts_dev_sync.argtypes = [C.c_int]
    L.mts_dev_compare.argtypes = [C.c_int, vp, vp, vp, C.c_long, lp, lp]
    L.mts_last_stage_times.restype = C.c_int
    L.mts_last_stage_times.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    L.mts_debug_match_tables.argtypes = [C.c_int, vp, C.c_long, C.c_int, vp, vp]
    L.mts_debug_tokens.argtypes = [C.c_int, vp, C.c_long, C.c_int, vp, lp]
    L.mts_debug_deflate.argtypes = [C.c_int, vp, C.c_long, C.c_int, vp, C.c_long, lp]
    L.mts_debug_inflate.argtypes = [C.c_int, vp, C.c_long, vp, C.c_long, lp, ip]
    L.mts_cache_create.argtypes = [C.c_int, C.c_long, lp]
    L.mts_cache_destroy.argtypes = [C.c_long]
    L.mts_cache_query.argtypes = [C.c_long, lp, C.c_int, ip]
    L.mts_cache_read_rows.argtypes = [C.c_long, C.c_int, lp, vp, lp, lp, lp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long,
                                      vp, ip]
    L.mts_cache_read_slices.argtypes = [C.c_long, C.c_int, lp, vp, lp, lp, lp, C.c_int, C.c_int, C.c_int, C.c_int, lp, vp, lp,
                                        C.c_long, ip]
    L.mts_cache_read_slices_leading.argtypes = [C.c_long, C.c_int, lp, vp, lp, lp, lp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, lp, vp, lp,
                                                C.c_long, ip]
    L.mts_release.restype = None
    _lib = L
    return L


EXPORTS = ['mts_version', 'mts_device_count', 'mts_strerror', 'mts_last_error', 'mts_compress_bound',
           'mts_delta_transpose', 'mts_cumsum_transpose', 'mts_compress_chunks', 'mts_decompress_chunks',
           'mts_dev_compress_chunks', 'mts_dev_decompress_chunks', 'mts_dev_synth_int16',
           'mts_host_alloc', 'mts_host_free', 'mts_dev_alloc', 'mts_dev_free', 'mts_dev_copy', 'mts_dev_sync', 'mts_dev_compare',
           'mts_last_stage_times', 'mts_debug_match_tables', 'mts_debug_tokens', 'mts_debug_deflate',
           'mts_debug_inflate', 'mts_release', 'mts_cache_create', 'mts_cache_destroy', 'mts_cache_query',
           'mts_cache_read_rows', 'mts_cache_read_slices', 'mts_cache_read_slices_leading']


def _check(rc, what):
    if rc != 0:
        raise HipError(rc, what, lib().mts_last_error().decode())


def device_count():
    return int(lib().mts_device_count())


def require_device():
    n = device_count()
    if n <= 0:
        raise HipError(E_NODEV, 'mtscomp_amd', 'no MI355X (gfx950) device visible; there is no CPU fallback')
    return n


def compress_bound(n):
    return int(lib().mts_compress_bound(int(n)))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _longs(seq):
    return np.ascontiguousarray(np.asarray(seq, dtype=np.int64))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_long))


def make_flags(do_time_diff=True, do_spatial_diff=False, chunk_order='F'):
    return ((FLAG_TIME_DIFF if do_time_diff else 0) | (FLAG_SPATIAL_DIFF if do_spatial_diff else 0) |
            (FLAG_ORDER_F if chunk_order == 'F' else 0))


def check_dtype(dtype):
    dtype = np.dtype(dtype)
    if not ((dtype.kind in 'iu' and dtype.itemsize in (1, 2, 4, 8)) or (dtype.kind == 'f' and dtype.itemsize in (4, 8))):
        raise NotImplementedError(
            'the MI355X codec handles integer dtypes of 1/2/4/8 bytes and float32/float64; got %s' % dtype)
    return dtype


def _dflags(flags, dtype):
    """flags as the C ABI wants them: the float bit comes from the dtype."""
    return (int(flags) & ~FLAG_FLOAT) | (FLAG_FLOAT if np.dtype(dtype).kind == 'f' else 0)


# ------------------------------------------------------------------------------------------------
# host-buffer entry points
# ------------------------------------------------------------------------------------------------
def delta_transpose(chunk, flags, device=0):
    chunk = np.ascontiguousarray(chunk)
    check_dtype(chunk.dtype)
    nt, nc = chunk.shape
    out = np.empty(chunk.nbytes, dtype=np.uint8)
    _check(lib().mts_delta_transpose(device, _ptr(chunk), nt, nc, chunk.itemsize, _dflags(flags, chunk.dtype), _ptr(out)),
           'mts_delta_transpose')
    return out


def cumsum_transpose(stream, nt, nc, dtype, flags, device=0):
    dtype = check_dtype(dtype)
    stream = np.ascontiguousarray(np.frombuffer(stream, dtype=np.uint8) if not isinstance(stream, np.ndarray)
                                  else stream.view(np.uint8).ravel())
    assert stream.size == nt * nc * dtype.itemsize
    out = np.empty((nt, nc), dtype=dtype)
    _check(lib().mts_cumsum_transpose(device, _ptr(stream), nt, nc, dtype.itemsize, _dflags(flags, dtype), _ptr(out)),
           'mts_cumsum_transpose')
    return out


def compress_chunks(data, chunk_bounds, flags, level=6, device=0):
    """data: C-contiguous (rows, n_channels) array whose row 0 is chunk_bounds[0].
    Returns the list of zlib streams, one per chunk."""
    data = np.ascontiguousarray(data)
    check_dtype(data.dtype)
    b = _longs(chunk_bounds)
    n_chunks = len(b) - 1
    assert data.shape[0] == b[-1] - b[0]
    row = data.shape[1] * data.itemsize
    bounds = [(compress_bound(int(b[i + 1] - b[i]) * row) + 15) // 16 * 16 for i in range(n_chunks)]
    slots = _longs(np.concatenate(([0], np.cumsum(bounds)))[:-1]) if n_chunks else _longs([])
    out = np.empty(int(sum(bounds)) + 16, dtype=np.uint8)
    sizes = np.zeros(max(n_chunks, 1), dtype=np.int64)
    _check(lib().mts_compress_chunks(device, _ptr(data), data.shape[1], data.itemsize, _lp(b), n_chunks,
                                     _dflags(flags, data.dtype), level, _ptr(out), _lp(slots), _lp(sizes)), 'mts_compress_chunks')
    return [out[int(slots[i]):int(slots[i]) + int(sizes[i])].tobytes() for i in range(n_chunks)]


def decompress_chunks(cbufs, n_rows, n_channels, dtype, flags, device=0, out=None):
    """cbufs: list of bytes-like compressed chunks, or (buffer, offsets, lengths) for chunks that already sit in
    one buffer.  Returns (status list, list of arrays or None).  The arrays are views of ONE output buffer, back to
    back in the order given, so consecutive chunks can be joined without copying.  `out`: a C-contiguous array of
    exactly the decoded size to decode into (the views are then views of it)."""
    dtype = check_dtype(dtype)
    if isinstance(cbufs, tuple):
        buf, offs, lens = cbufs
        n = len(lens)
        if n == 0:
            return [], []
        cdata = np.frombuffer(buf, dtype=np.uint8)
        offs, lens = _longs(offs), _longs(lens)
        if int(offs[-1] + lens[-1]) + 16 > cdata.size:          # the kernels may read a few bytes past a stream
            cdata = np.concatenate((cdata, np.zeros(16, dtype=np.uint8)))
    else:
        n = len(cbufs)
        if n == 0:
            return [], []
        lens = _longs([len(c) for c in cbufs])
        offs = _longs(np.concatenate(([0], np.cumsum(lens)))[:-1])
        cdata = np.frombuffer(b''.join(bytes(c) for c in cbufs) + b'\0' * 16, dtype=np.uint8)
    rows = _longs(n_rows)
    sizes = rows * (n_channels * dtype.itemsize)
    ooffs = _longs(np.concatenate(([0], np.cumsum(sizes)))[:-1])
    if out is None:
        out = np.empty(int(ooffs[-1] + sizes[-1]) + 256, dtype=np.uint8)
    else:
        assert out.flags.c_contiguous and out.nbytes == int(ooffs[-1] + sizes[-1])
        out = out.reshape(-1).view(np.uint8)
    status = np.zeros(n, dtype=np.int32)
    _check(lib().mts_decompress_chunks(device, _ptr(cdata), _lp(offs), _lp(lens), _lp(rows), n, n_channels,
                                       dtype.itemsize, _dflags(flags, dtype), _ptr(out), _lp(ooffs),
                                       status.ctypes.data_as(C.POINTER(C.c_int))), 'mts_decompress_chunks')
    arrays = []
    for i in range(n):
        if status[i] == CHUNK_OK:
            arrays.append(out[int(ooffs[i]):int(ooffs[i] + sizes[i])].view(dtype).reshape(int(rows[i]), n_channels))
        else:
            arrays.append(None)
    return [int(s) for s in status], arrays


# ------------------------------------------------------------------------------------------------
# decoded-chunk cache on the device (Reader random access)
# ------------------------------------------------------------------------------------------------
def cache_create(capacity_bytes, device=0):
    cid = C.c_long(0)
    _check(lib().mts_cache_create(device, int(capacity_bytes), C.byref(cid)), 'mts_cache_create')
    return int(cid.value)


def cache_destroy(cache_id):
    if _lib is not None:
        _lib.mts_cache_destroy(int(cache_id))


def cache_query(cache_id, keys):
    """Per key: 0 = not resident, else the number of channels the resident entry holds."""
    keys = _longs(keys)
    present = np.zeros(keys.size, dtype=np.int32)
    _check(lib().mts_cache_query(int(cache_id), _lp(keys), int(keys.size), present.ctypes.data_as(C.POINTER(C.c_int))),
           'mts_cache_query')
    return present


def cache_read_rows(cache_id, keys, cdata, offs, lens, n_rows, n_channels, dtype, flags, row_begin, row_end, out=None):
    """Rows [row_begin, row_end) of the concatenation of the chunks `keys` (file order).  Chunks with lens[i] == 0 must be
    resident (HipError with code E_MISS otherwise).  Returns (status list, (row_end - row_begin, n_channels) array); `out`: a
    C-contiguous array of exactly that shape to fill instead of a new one."""
    dtype = check_dtype(dtype)
    keys, offs, lens, rows = _longs(keys), _longs(offs), _longs(lens), _longs(n_rows)
    n = int(keys.size)
    cdata = np.frombuffer(cdata, dtype=np.uint8) if len(cdata) else np.zeros(16, dtype=np.uint8)
    if n and int((offs + lens).max()) + 16 > cdata.size:          # the kernels may read a few bytes past a stream
        cdata = np.concatenate((cdata, np.zeros(16, dtype=np.uint8)))
    if out is None:
        out = np.empty((int(row_end - row_begin), n_channels), dtype=dtype)
    else:
        assert out.flags.c_contiguous and out.dtype == dtype and out.shape == (int(row_end - row_begin), n_channels)
    status = np.zeros(n, dtype=np.int32)
    _check(lib().mts_cache_read_rows(int(cache_id), n, _lp(keys), _ptr(cdata), _lp(offs), _lp(lens), _lp(rows), n_channels,
                                     dtype.itemsize, _dflags(flags, dtype), int(row_begin), int(row_end), _ptr(out),
                                     status.ctypes.data_as(C.POINTER(C.c_int))), 'mts_cache_read_rows')
    return [int(x) for x in status], out


def cache_read_slices(cache_id, keys, cdata, offs, lens, n_rows, n_channels, dtype, flags, requests, n_leading=None):
    """Several rectangles of the concatenation of the chunks `keys` in one call: requests = [(row_begin, row_end, row_step,
    col_begin, col_end, col_step), ...] (steps >= 1).  The pieces are gathered on the device and come back in one copy.
    Returns (status list, list of 2-D arrays).  n_leading: the requests only touch channels below it, chunks that are not
    resident are decoded up to there only and `lens` may be prefixes of their bytes (mts_cache_read_slices_leading)."""
    dtype = check_dtype(dtype)
    keys, offs, lens, rows = _longs(keys), _longs(offs), _longs(lens), _longs(n_rows)
    n = int(keys.size)
    cdata = np.frombuffer(cdata, dtype=np.uint8) if len(cdata) else np.zeros(16, dtype=np.uint8)
    if n and int((offs + lens).max()) + 16 > cdata.size:          # the kernels may read a few bytes past a stream
        cdata = np.concatenate((cdata, np.zeros(16, dtype=np.uint8)))
    req = _longs(np.asarray(requests, dtype=np.int64).reshape(-1, 6))
    shapes = [(int(-(-(q[1] - q[0]) // q[2])), int(-(-(q[4] - q[3]) // q[5]))) for q in req.reshape(-1, 6)]
    sizes = [(a * b * dtype.itemsize + 255) // 256 * 256 for a, b in shapes]
    out_offs = _longs(np.concatenate(([0], np.cumsum(sizes)))[:-1]) if shapes else _longs([])
    out = np.empty(int(sum(sizes)) + 8, dtype=np.uint8)
    status = np.zeros(max(n, 1), dtype=np.int32)
    _check(lib().mts_cache_read_slices_leading(int(cache_id), n, _lp(keys), _ptr(cdata), _lp(offs), _lp(lens), _lp(rows), n_channels,
                                               dtype.itemsize, _dflags(flags, dtype), int(n_leading or n_channels), len(shapes), _lp(req),
                                               _ptr(out), _lp(out_offs), int(sum(sizes)), status.ctypes.data_as(C.POINTER(C.c_int))),
           'mts_cache_read_slices_leading')
    arrays = [out[int(o):int(o) + a * b * dtype.itemsize].view(dtype).reshape(a, b) for o, (a, b) in zip(out_offs, shapes)]
    return [int(x) for x in status[:n]], arrays


# ------------------------------------------------------------------------------------------------
# device-resident recordings (bench.py, the tests at BASELINE's sizes): memory held through the library -- no second HIP runtime
# ------------------------------------------------------------------------------------------------
class HostBuffer:
    """Page-locked host memory (mts_host_alloc) as a numpy array: `.array` (uint8, nbytes).  Copies between it and the device are
    DMA transfers without a staging copy.  Keep the object alive while the array (or views of it) is in use."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _check(lib().mts_host_alloc(self.nbytes, C.byref(p)), 'mts_host_alloc')
        self.ptr = p.value or 0
        self.array = np.ctypeslib.as_array((C.c_ubyte * max(self.nbytes, 1)).from_address(self.ptr))[:self.nbytes]

    def free(self):
        if self.ptr:
            self.array = None
            ptr, self.ptr = self.ptr, 0
            _check(lib().mts_host_free(C.c_void_p(ptr)), 'mts_host_free')

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class PinnedPool:
    """Page-locked buffers kept between calls.  hipHostMalloc pins its pages one by one (a few hundred MB take a good part of a
    second) and hipHostFree unpins them again: a Reader.tofile that allocated its two piece buffers per call spent more time on
    that than on the file.  take() hands out an idle buffer of at least the size asked for (the smallest that fits, grown by
    a quarter when a new one has to be made); give() returns it; at most `keep_bytes` stay idle, the largest first
    (MTSCOMP_PINNED_KEEP_MB, default 2048; 0 keeps nothing).  clear() frees the idle ones: HipCodec.close() and release() call it."""

    def __init__(self, keep_bytes=None):
        import threading
        if keep_bytes is None:
            keep_bytes = int(os.environ.get('MTSCOMP_PINNED_KEEP_MB', 2048)) << 20
        self.keep_bytes = int(keep_bytes)
        self._idle = []
        self._lock = threading.Lock()

    def take(self, nbytes):
        nbytes = int(nbytes)
        with self._lock:
            fits = [b for b in self._idle if b.nbytes >= nbytes]
            if fits:
                best = min(fits, key=lambda b: b.nbytes)
                self._idle.remove(best)
                return best
        return HostBuffer(nbytes + nbytes // 4)

    def give(self, buf):
        if buf is None or not buf.ptr:
            return
        drop = []
        with self._lock:
            self._idle.append(buf)
            self._idle.sort(key=lambda b: -b.nbytes)
            while sum(b.nbytes for b in self._idle) > self.keep_bytes and len(self._idle) > 1:
                drop.append(self._idle.pop())
            if self._idle and (self._idle[0].nbytes > self.keep_bytes or self.keep_bytes <= 0):
                drop.append(self._idle.pop(0))
        for b in drop:
            b.free()

    def clear(self):
        with self._lock:
            idle, self._idle = self._idle, []
        for b in idle:
            b.free()


pinned_pool = PinnedPool()


def release():
    """Give back what the process keeps between calls: the idle page-locked buffers of `pinned_pool`, then the library's
    per-device workspaces (mts_release)."""
    pinned_pool.clear()
    if _lib is not None:
        _lib.mts_release()


class DevBuffer:
    """`nbytes` of HBM on `device`, allocated, copied and freed by libmtscomp_hip.so (mts_dev_alloc / mts_dev_copy / mts_dev_free)."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = int(device), int(nbytes)
        p = C.c_void_p()
        _check(lib().mts_dev_alloc(self.device, self.nbytes, C.byref(p)), 'mts_dev_alloc')
        self.ptr = p.value or 0

    def at(self, offset=0):
        assert 0 <= offset <= self.nbytes
        return C.c_void_p(self.ptr + int(offset))

    def upload(self, arr, offset=0):
        a = np.ascontiguousarray(arr)
        assert offset + a.nbytes <= self.nbytes
        _check(lib().mts_dev_copy(self.device, None, self.at(offset), _ptr(a), a.nbytes, 0), 'mts_dev_copy')

    def download(self, offset=0, nbytes=None, dtype=np.uint8):
        nbytes = self.nbytes - offset if nbytes is None else int(nbytes)
        assert offset + nbytes <= self.nbytes
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _check(lib().mts_dev_copy(self.device, None, _ptr(out), self.at(offset), out.nbytes, 1), 'mts_dev_copy')
        return out

    def diff(self, other, nbytes=None):
        """(bytes that differ, first such offset or -1) between this buffer and `other` (compared on the device)."""
        n, first = C.c_long(0), C.c_long(-1)
        nbytes = min(self.nbytes, other.nbytes) if nbytes is None else int(nbytes)
        _check(lib().mts_dev_compare(self.device, None, self.at(), other.at(), nbytes, C.byref(n), C.byref(first)), 'mts_dev_compare')
        return int(n.value), int(first.value)

    def free(self):
        if self.ptr:
            ptr, self.ptr = self.ptr, 0
            _check(lib().mts_dev_free(self.device, C.c_void_p(ptr)), 'mts_dev_free')

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def dev_sync(device=0):
    _check(lib().mts_dev_sync(int(device)), 'mts_dev_sync')


def dev_synth_int16(buf, offset, t0, t1, n_channels, seed=0):
    """Rows [t0, t1) of the synthetic recording written at `offset` of a DevBuffer."""
    assert offset + (t1 - t0) * n_channels * 2 <= buf.nbytes
    _check(lib().mts_dev_synth_int16(buf.device, None, buf.at(offset), int(t0), int(t1), int(n_channels), int(seed)), 'mts_dev_synth_int16')


def dev_compress_chunks(raw, n_channels, itemsize, bounds, flags, level, out, slots, sizes):
    """mts_dev_compress_chunks on DevBuffers: `bounds` rows (int64 array, n + 1), `slots` byte offsets into `out`, `sizes` filled."""
    _check(lib().mts_dev_compress_chunks(raw.device, None, raw.at(), n_channels, itemsize, _lp(bounds), len(bounds) - 1, int(flags), int(level),
                                         out.at(), _lp(slots), _lp(sizes)), 'mts_dev_compress_chunks')


def dev_decompress_chunks(cbuf, offs, lens, rows, n_channels, itemsize, flags, out, out_offs, status):
    _check(lib().mts_dev_decompress_chunks(cbuf.device, None, cbuf.at(), _lp(offs), _lp(lens), _lp(rows), len(rows), n_channels, itemsize, int(flags),
                                           out.at(), _lp(out_offs), status.ctypes.data_as(C.POINTER(C.c_int))), 'mts_dev_decompress_chunks')


def last_stage_times(device=0):
    names = (C.c_char_p * 32)()
    ms = (C.c_float * 32)()
    n = lib().mts_last_stage_times(device, names, ms, 32)
    return [(names[i].decode(), float(ms[i])) for i in range(n)]


# ------------------------------------------------------------------------------------------------
# debug taps (GPU parity tests)
# ------------------------------------------------------------------------------------------------
def _u8(data):
    return np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) \
        else np.ascontiguousarray(data.view(np.uint8).ravel())


def debug_match_tables(data, level=6, device=0):
    a = _u8(data)
    tf = np.zeros(max(a.size, 1), dtype=np.uint32)
    tq = np.zeros(max(a.size, 1), dtype=np.uint32)
    _check(lib().mts_debug_match_tables(device, _ptr(a), a.size, level, _ptr(tf), _ptr(tq)), 'mts_debug_match_tables')
    return tf[:a.size], tq[:a.size]


def debug_tokens(data, level=6, device=0):
    a = _u8(data)
    toks = np.zeros((a.size + 1, 2), dtype=np.uint16)
    n = C.c_long(0)
    _check(lib().mts_debug_tokens(device, _ptr(a), a.size, level, _ptr(toks), C.byref(n)), 'mts_debug_tokens')
    return toks[:n.value].copy()


def debug_deflate(data, level=6, device=0):
    a = _u8(data)
    cap = compress_bound(a.size) + 64
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_long(0)
    _check(lib().mts_debug_deflate(device, _ptr(a), a.size, level, _ptr(out), cap, C.byref(n)), 'mts_debug_deflate')
    return out[:n.value].tobytes()


def debug_inflate(zbytes, expect_len, device=0):
    """Returns (status, bytes)."""
    z = _u8(zbytes)
    out = np.zeros(max(expect_len, 1), dtype=np.uint8)
    n, st = C.c_long(0), C.c_int(0)
    _check(lib().mts_debug_inflate(device, _ptr(z), z.size, _ptr(out), expect_len, C.byref(n), C.byref(st)),
           'mts_debug_inflate')
    return int(st.value), out[:n.value].tobytes()
