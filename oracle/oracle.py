"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module; the product package (``mtscomp_amd``) never does.

Two checkers live here:

* ``libmtsc_oracle.so`` (``mtsc_oracle.c``): plain-C restatement of the per-chunk path
  (delta/transposes, zlib 1.2.11 deflate at levels 1-9, RFC 1950/1951 inflate, adler32) with
  stage-by-stage reports (tokens, candidate tables, block layout).
* ``ref_compress_chunk`` / ``ref_decompress_chunk``: the reference's own statement sequence
  (``/root/reference/mtscomp.py:375-397`` and ``:602-635``) on numpy + the stdlib ``zlib`` module,
  i.e. the very libz the reference calls.  This is also what ``bench.py`` times as ``cpu_baseline``.
"""
import ctypes as C
import os
import subprocess
import zlib
from multiprocessing.dummy import Pool as ThreadPool
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / 'libmtsc_oracle.so'

FLAG_TIME_DIFF = 1
FLAG_SPATIAL_DIFF = 2
FLAG_ORDER_F = 4


def build(force=False):
    src = _HERE / 'mtsc_oracle.c'
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(['make', '-C', str(_HERE), '-s', '-B'])
    return _SO


class BlockInfo(C.Structure):
    _fields_ = [('tok_start', C.c_long), ('ntok', C.c_long), ('in_start', C.c_long),
                ('in_len', C.c_long), ('opt_len', C.c_long), ('static_len', C.c_long),
                ('btype', C.c_int), ('last', C.c_int), ('bit_start', C.c_long),
                ('bit_end', C.c_long)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_SO))
        u8p, lp = C.POINTER(C.c_uint8), C.POINTER(C.c_long)
        L.orc_adler32.restype = C.c_uint32
        L.orc_adler32.argtypes = [C.c_void_p, C.c_long]
        L.orc_delta_transpose.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_void_p]
        L.orc_cumsum_transpose.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_void_p]
        L.orc_deflate.restype = C.c_long
        L.orc_deflate.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_void_p,
                                  C.c_void_p, lp, C.c_void_p, C.c_long, lp]
        L.orc_compress_bound.restype = C.c_long
        L.orc_compress_bound.argtypes = [C.c_long]
        L.orc_match_tables.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_parse_tables.restype = C.c_long
        L.orc_parse_tables.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
        L.orc_inflate.restype = C.c_long
        L.orc_inflate.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, lp]
        L.orc_compress_chunk.restype = C.c_long
        L.orc_compress_chunk.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_long]
        L.orc_decompress_chunk.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_int,
                                           C.c_int, C.c_void_p]
        L.orc_slides_at.restype = C.c_long
        L.orc_slides_at.argtypes = [C.c_long, C.c_long]
        _ = u8p
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _bytes_arr(b):
    a = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b
    return np.ascontiguousarray(a.view(np.uint8).ravel())


def make_flags(do_time_diff=True, do_spatial_diff=False, chunk_order='F'):
    return ((FLAG_TIME_DIFF if do_time_diff else 0) | (FLAG_SPATIAL_DIFF if do_spatial_diff else 0) |
            (FLAG_ORDER_F if chunk_order == 'F' else 0))


# ------------------------------------------------------------------------------------------------
# C oracle
# ------------------------------------------------------------------------------------------------
def _dflags(flags, dtype):
    """the float bit (8) of the C functions' flags comes from the dtype"""
    return (int(flags) & ~8) | (8 if np.dtype(dtype).kind == 'f' else 0)


def adler32(data):
    a = _bytes_arr(data)
    return int(lib().orc_adler32(_ptr(a), a.size))


def delta_transpose(chunk, flags):
    """chunk: (nt, nc) C-contiguous integer array -> uint8 stream."""
    chunk = np.ascontiguousarray(chunk)
    nt, nc = chunk.shape
    out = np.empty(chunk.nbytes, dtype=np.uint8)
    rc = lib().orc_delta_transpose(_ptr(chunk), nt, nc, chunk.itemsize, _dflags(flags, chunk.dtype), _ptr(out))
    assert rc == 0, rc
    return out


def cumsum_transpose(stream, nt, nc, dtype, flags):
    stream = _bytes_arr(stream)
    dtype = np.dtype(dtype)
    out = np.empty((nt, nc), dtype=dtype)
    rc = lib().orc_cumsum_transpose(_ptr(stream), nt, nc, dtype.itemsize, _dflags(flags, dtype), _ptr(out))
    assert rc == 0, rc
    return out


def deflate(data, level=6, report=False):
    """zlib.compress(data, level) restated.  With report=True also returns tokens/blocks."""
    a = _bytes_arr(data)
    n = a.size
    cap = int(lib().orc_compress_bound(n)) + 64
    out = np.empty(cap, dtype=np.uint8)
    if not report:
        r = lib().orc_deflate(_ptr(a), n, level, _ptr(out), cap, None, None, None, None, 0, None)
        assert r >= 0, r
        return out[:r].tobytes()
    toks = np.empty((n + 1, 2), dtype=np.uint16)
    tokpos = np.empty(n + 1, dtype=np.int64)
    nblk_cap = n // 16383 + 4
    binfo = (BlockInfo * nblk_cap)()
    ntok, nblk = C.c_long(0), C.c_long(0)
    r = lib().orc_deflate(_ptr(a), n, level, _ptr(out), cap, _ptr(toks), _ptr(tokpos),
                          C.byref(ntok), C.cast(binfo, C.c_void_p), nblk_cap, C.byref(nblk))
    assert r >= 0, r
    blocks = [{f: getattr(binfo[i], f) for f, _ in BlockInfo._fields_} for i in range(nblk.value)]
    return out[:r].tobytes(), toks[:ntok.value].copy(), tokpos[:ntok.value].copy(), blocks


def match_tables(data, level=6):
    a = _bytes_arr(data)
    tf = np.zeros(max(a.size, 1), dtype=np.uint32)
    tq = np.zeros(max(a.size, 1), dtype=np.uint32)
    rc = lib().orc_match_tables(_ptr(a), a.size, level, _ptr(tf), _ptr(tq))
    assert rc == 0, rc
    return tf[:a.size], tq[:a.size]


def parse_tables(data, tf, tq, level=6):
    a = _bytes_arr(data)
    toks = np.empty((a.size + 1, 2), dtype=np.uint16)
    tokpos = np.empty(a.size + 1, dtype=np.int64)
    tf = np.ascontiguousarray(tf, dtype=np.uint32)
    tq = np.ascontiguousarray(tq, dtype=np.uint32)
    n = lib().orc_parse_tables(_ptr(a), a.size, level, _ptr(tf), _ptr(tq), _ptr(toks), _ptr(tokpos))
    assert n >= 0, n
    return toks[:n].copy(), tokpos[:n].copy()


def inflate(data, max_out):
    """Returns (bytes, consumed) or raises ValueError(code)."""
    a = _bytes_arr(data)
    out = np.empty(max(max_out, 1), dtype=np.uint8)
    used = C.c_long(0)
    r = lib().orc_inflate(_ptr(a), a.size, _ptr(out), max_out, C.byref(used))
    if r < 0:
        raise ValueError(int(r))
    return out[:r].tobytes(), used.value


def compress_chunk(chunk, flags, level=6):
    chunk = np.ascontiguousarray(chunk)
    nt, nc = chunk.shape
    cap = int(lib().orc_compress_bound(chunk.nbytes)) + 64
    out = np.empty(cap, dtype=np.uint8)
    r = lib().orc_compress_chunk(_ptr(chunk), nt, nc, chunk.itemsize, _dflags(flags, chunk.dtype), level, _ptr(out), cap)
    assert r >= 0, r
    return out[:r].tobytes()


def decompress_chunk(cbuf, nt, nc, dtype, flags):
    """Returns (status, array).  status 0 ok, <0 corrupt, 1 wrong size."""
    a = _bytes_arr(cbuf)
    dtype = np.dtype(dtype)
    out = np.empty((nt, nc), dtype=dtype)
    rc = lib().orc_decompress_chunk(_ptr(a), a.size, nt, nc, dtype.itemsize, _dflags(flags, dtype), _ptr(out))
    return int(rc), out


def slides_at(q, n):
    return int(lib().orc_slides_at(q, n))


# ------------------------------------------------------------------------------------------------
# The reference's statement sequence on numpy + stdlib zlib (the libz the reference calls)
# ------------------------------------------------------------------------------------------------
def ref_diff_along_axis(chunk, axis=None):
    """/root/reference/mtscomp.py:143-159."""
    if axis is None:
        return chunk
    d = np.diff(chunk, axis=axis)
    if axis == 0:
        return np.concatenate((chunk[0, :][np.newaxis, :], d), axis=axis)
    return np.concatenate((chunk[:, 0][:, np.newaxis], d), axis=axis)


def ref_cumsum_along_axis(chunk, axis=None):
    """/root/reference/mtscomp.py:162-169."""
    if axis is None:
        return chunk
    out = np.empty_like(chunk)
    np.cumsum(chunk, axis=axis, out=out)
    return out


def ref_compress_chunk(chunk, do_time_diff=True, do_spatial_diff=False, chunk_order='F', level=None):
    """/root/reference/mtscomp.py:375-397 (level=None = what the reference does: zlib default)."""
    d = ref_diff_along_axis(chunk, axis=0 if do_time_diff else None)
    d = ref_diff_along_axis(d, axis=1 if do_spatial_diff else None)
    b = d.tobytes(order=chunk_order)
    return zlib.compress(b) if level is None else zlib.compress(b, level)


def ref_decompress_chunk(cbuf, nt, nc, dtype, do_time_diff=True, do_spatial_diff=False, chunk_order='F'):
    """/root/reference/mtscomp.py:618-635."""
    buf = zlib.decompress(cbuf)
    chunk = np.frombuffer(buf, np.dtype(dtype))
    assert chunk.size == nt * nc
    chunk = chunk.reshape((nt, nc), order=chunk_order)
    c = ref_cumsum_along_axis(chunk, axis=1 if do_spatial_diff else None)
    c = ref_cumsum_along_axis(c, axis=0 if do_time_diff else None)
    return np.ascontiguousarray(c)


def ref_compress_array(data, chunk_bounds, n_threads=1, **kw):
    """The reference's batch scheduler (mtscomp.py:399-423, 461-483) over an in-memory array:
    batches of n_threads chunks through a thread pool.  Returns the list of compressed chunks."""
    ids = list(range(len(chunk_bounds) - 1))

    def one(i):
        return ref_compress_chunk(data[chunk_bounds[i]:chunk_bounds[i + 1]], **kw)
    if n_threads == 1:
        return [one(i) for i in ids]
    out = []
    with ThreadPool(n_threads) as pool:
        for b0 in range(0, len(ids), n_threads):
            out.extend(pool.map(one, ids[b0:b0 + n_threads]))
    return out


def ref_decompress_array(cchunks, chunk_bounds, nc, dtype, n_threads=1, **kw):
    """mtscomp.py:645-650, 720-734 over in-memory compressed chunks."""
    ids = list(range(len(chunk_bounds) - 1))

    def one(i):
        return ref_decompress_chunk(cchunks[i], chunk_bounds[i + 1] - chunk_bounds[i], nc, dtype, **kw)
    if n_threads == 1:
        return [one(i) for i in ids]
    out = []
    with ThreadPool(n_threads) as pool:
        for b0 in range(0, len(ids), n_threads):
            out.extend(pool.map(one, ids[b0:b0 + n_threads]))
    return out


if __name__ == '__main__':
    build(force=True)
    print(_SO, os.path.getsize(_SO))
