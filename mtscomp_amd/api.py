"""Host side of the drop-in: mtscomp's ``compress()/decompress()/Writer/Reader`` API and ``.cbin/.ch``
on-disk format, with the per-chunk codec running on MI355X through ``libmtscomp_hip.so``.

The interface mirrors /root/reference/mtscomp.py (names, argument meaning, error behaviour); every
public item cites the reference lines it stands in for.  What changed is below the interface:

* ``Writer.compress_batch`` hands a whole batch of chunks to ``mts_compress_chunks`` (one C-ABI call
  per device) instead of ``pool.map(self._compress_chunk, ...)`` (mtscomp.py:399-423);
* ``Reader.decompress_chunks`` / ``__getitem__`` hand all missing chunks to ``mts_decompress_chunks``
  instead of ``pool.map(self._decompress_chunk, ...)`` (mtscomp.py:645-650, 810-812);
* chunks are sharded round-robin over the visible GPUs (chunk i -> device i mod G); the only
  cross-device step is the host-side gather of compressed sizes into ``chunk_offsets``.

There is no CPU implementation of the codec in this package: the codec object is ``HipCodec`` and it
raises if the HIP library or a gfx950 device is missing.
"""
import bisect
import hashlib
import json
import logging
import multiprocessing as mp
import os
import threading
import os.path as op
from collections import OrderedDict
from multiprocessing.dummy import Pool as ThreadPool
from pathlib import Path
from threading import Lock

import numpy as np

from . import hip

__version__ = '0.1.0'
FORMAT_VERSION = '1.0'          # mtscomp.py:41

__all__ = ('load_raw_data', 'Writer', 'Reader', 'compress', 'decompress', 'check', 'diff_along_axis',
           'cumsum_along_axis', 'read_config', 'write_config', 'add_default_handler', 'HipCodec')

# mtscomp.py:46-57 -- same keys, same defaults; the .ch header only ever sees the reference's keys.
DEFAULT_CONFIG = tuple(dict(
    algorithm='zlib',
    cache_size=10,
    check_after_compress=True,
    check_after_decompress=True,
    chunk_duration=1.,
    chunk_order='F',
    comp_level=-1,              # stored in the header, never passed to the codec (mtscomp.py:394)
    do_spatial_diff=False,
    do_time_diff=True,
    n_threads=mp.cpu_count(),   # kept for compatibility; the device path does not use host threads
).items())

CHECK_ATOL = 1e-16              # mtscomp.py:59
CRITICAL_ERROR_URL = "https://github.com/int-brain-lab/mtscomp/issues/new?title=Critical+error"
DEFAULT_BATCH_CHUNKS = 64       # chunks handed to one device call
TOFILE_PIECE_CHUNKS = 8         # chunks per piece of Reader.tofile (decode of one piece under the file writes of the one before)
TOFILE_WRITERS = 4              # threads writing a piece: a fresh tmpfs file takes ~6.5 GB/s whatever their number (its pages are allocated and zeroed under the inode's lock), an existing file written over in place 6-9 GB/s from four (round 5: 2 against 4 writers, new files 2.8-6.2 / 4.0-5.1 GB/s from call to call, over existing files 6.3-8.9 / 6.7-9.0)
DEFAULT_DEVICE_CACHE_GB = 32    # decoded chunks a Reader may keep in HBM for slicing (allocated as touched; env MTSCOMP_DEVICE_CACHE_GB, 0 = off)
DEVICE_CACHE_MAX_CHUNKS = 8     # longer slices are streamed through the host path instead of the cache
READ_AHEAD_MAX = int(os.environ.get('MTSCOMP_READ_AHEAD', 4))   # chunks a cold slice may decode ahead of itself into the device cache (0 = off)
MAP_CDATA = os.environ.get('MTSCOMP_MAP_CDATA', '1') not in ('', '0')      # slices read their compressed bytes out of a mapping of the .cbin (0: preadv into page-locked memory)
PREAD_THREADS = int(os.environ.get('MTSCOMP_PREAD_THREADS', 8))      # threads that read the compressed bytes of a slice's missing chunks (a few MB and more)

logger = logging.getLogger('mtscomp_amd')
logger.setLevel(logging.INFO)
logger.addHandler(logging.NullHandler())

_seek_lock = Lock()


def add_default_handler(level='INFO', logger=logger):
    """Attach a stream handler (mtscomp.py:89-96)."""
    handler = logging.StreamHandler()
    handler.setLevel(level)
    handler.setFormatter(logging.Formatter('%(asctime)s [%(levelname).1s] %(message)s', '%H:%M:%S'))
    logger.addHandler(handler)


class Bunch(dict):
    """dict with attribute access (mtscomp.py:99-104)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


def _clip(x, lo, hi):
    return max(lo, min(hi, x))


# ------------------------------------------------------------------------------------------------
# config (mtscomp.py:176-209)
# ------------------------------------------------------------------------------------------------
def config_path():
    return (Path('~') / '.mtscomp').expanduser()


CONFIG_PATH = config_path()


def read_config(**kwargs):
    """Defaults < ~/.mtscomp < kwargs; None values are ignored (mtscomp.py:186-200)."""
    params = dict(DEFAULT_CONFIG)
    user = {}
    if CONFIG_PATH.exists():
        with CONFIG_PATH.open('r') as f:
            user = json.load(f)
    for layer in (user, kwargs):
        params.update({k: v for k, v in layer.items() if v is not None})
    return Bunch(params)


def write_config(**kwargs):
    """Merge kwargs into the user's configuration file and return what it now holds (mtscomp.py:203-209)."""
    merged = read_config(**kwargs)
    CONFIG_PATH.parent.mkdir(parents=True, exist_ok=True)
    CONFIG_PATH.write_text(json.dumps(merged, indent=2, sort_keys=True))
    return merged


# ------------------------------------------------------------------------------------------------
# raw I/O (mtscomp.py:115-140)
# ------------------------------------------------------------------------------------------------
def load_raw_data(path=None, n_channels=None, dtype=None, offset=None, mmap=True):
    """A flat binary file as an (n_samples, n_channels) array, memory mapped unless mmap=False (mtscomp.py:115-140).
    Contract: AssertionError for a missing file or dtype, ValueError when the size is no whole number of rows, an empty
    (0, n_channels) array for an empty file."""
    path = Path(path)
    assert path.exists(), "File %s does not exist." % path
    assert dtype, "The data type must be provided."
    n_channels, offset = n_channels or 1, offset or 0
    f_size = path.stat().st_size
    n_samples, rest = divmod(f_size - offset, np.dtype(dtype).itemsize * n_channels)
    if rest:
        raise ValueError(
            "The file size (%d bytes) is incompatible with the specified parameters "
            "(n_channels=%d, dtype=%s, offset=%d)" % (f_size, n_channels, dtype, offset))
    if n_samples <= 0:
        return np.zeros((0, n_channels), dtype=dtype)
    if not mmap:
        if offset > 0:  # pragma: no cover
            raise NotImplementedError()
        return np.fromfile(str(path), dtype).reshape((n_samples, n_channels))
    return np.memmap(str(path), dtype=dtype, shape=(n_samples, n_channels), offset=offset)


# ------------------------------------------------------------------------------------------------
# the codec (device side)
# ------------------------------------------------------------------------------------------------
def _one_block(chunks):
    """The chunks as one (rows, n_channels) array: a view when they already lie back to back in memory (consecutive
    slices of one memmap, the Writer's case), a concatenation otherwise."""
    if len(chunks) == 1:
        return chunks[0]
    first = chunks[0]
    if first.ndim == 2 and first.size and all(c.flags['C_CONTIGUOUS'] and c.dtype == first.dtype and c.shape[1:] == first.shape[1:]
                                              for c in chunks):
        addr = first.__array_interface__['data'][0]
        for c in chunks:
            if c.__array_interface__['data'][0] != addr:
                break
            addr += c.nbytes
        else:
            rows = sum(c.shape[0] for c in chunks)
            # (as_strided keeps `first`, hence the buffer under all the chunks, alive)
            return np.lib.stride_tricks.as_strided(first, shape=(rows,) + first.shape[1:], writeable=False)
    return np.concatenate(chunks, axis=0)


class HipCodec:
    """Per-chunk codec on MI355X.  ``devices``: list of device indices (default: all visible)."""

    name = 'hip'
    takes_ranges = True          # decompress() also accepts (buffer, offsets, lengths)

    def __init__(self, devices=None):
        n = hip.require_device()
        self.devices = list(range(n)) if devices is None else [int(d) for d in devices]
        assert self.devices and all(0 <= d < n for d in self.devices), "invalid device list"
        self._pool = None                       # one host thread per entry of `devices`, kept for the codec's lifetime
        self._pool_lock = threading.Lock()

    # -- transforms alone (diff_along_axis / cumsum_along_axis)
    def delta(self, arr, flags):
        return hip.delta_transpose(arr, flags, device=self.devices[0])

    def cumsum(self, stream, nt, nc, dtype, flags):
        return hip.cumsum_transpose(stream, nt, nc, dtype, flags, device=self.devices[0])

    def _shards(self, n):
        g = len(self.devices)
        return [list(range(k, n, g)) for k in range(g)] if g > 1 and n > 1 else [list(range(n))]

    def compress(self, chunks, flags, level=6):
        """chunks: list of (n_t, n_c) arrays -> list of zlib streams (order preserved)."""
        n = len(chunks)
        out = [None] * n
        shards = [s for s in self._shards(n) if s]

        def run(k):
            ids = shards[k]
            rows = [chunks[i].shape[0] for i in ids]
            data = _one_block([chunks[i] for i in ids])
            bounds = np.concatenate(([0], np.cumsum(rows)))
            res = hip.compress_chunks(data, bounds, flags, level, device=self.devices[k % len(self.devices)])
            for i, b in zip(ids, res):
                out[i] = b
        self.run_lanes(run, len(shards))
        return out

    takes_out = True

    def decompress(self, cbufs, n_rows, n_channels, dtype, flags, out=None, lane=None):
        """-> (status list, list of (n_rows[i], n_channels) arrays or None).  `out`: optional array of exactly the decoded
        size, filled in place when one device does the whole call: a codec with one device, or `lane` = the entry of
        `devices` (modulo their number) that is to take all of it -- Reader.tofile hands its pieces to the devices in turn."""
        one = self.devices[lane % len(self.devices)] if lane is not None else self.devices[0] if len(self.devices) == 1 else None
        if isinstance(cbufs, tuple):                        # (buffer, offsets, lengths): chunks already in one buffer
            buf, offs, lens = cbufs
            if one is not None:
                return hip.decompress_chunks(cbufs, n_rows, n_channels, dtype, flags, device=one, out=out)
            mv = memoryview(buf)
            cbufs = [mv[o:o + l] for o, l in zip(offs, lens)]
        elif one is not None:
            return hip.decompress_chunks(cbufs, n_rows, n_channels, dtype, flags, device=one, out=out)
        n = len(cbufs)
        status, arrays = [0] * n, [None] * n
        shards = [s for s in self._shards(n) if s]

        def run(k):
            ids = shards[k]
            st, arrs = hip.decompress_chunks([cbufs[i] for i in ids], [n_rows[i] for i in ids], n_channels, dtype,
                                             flags, device=self.devices[k % len(self.devices)])
            for i, s, a in zip(ids, st, arrs):
                status[i], arrays[i] = s, a
        self.run_lanes(run, len(shards))
        return status, arrays

    # -- decoded-chunk cache in HBM (Reader random access): one cache per lane, chunk k lives on lane k mod n_lanes
    device_cache = True
    leading_channels = True      # cache_read_slices(..., n_leading=): chunks decoded up to the leading channels a request needs

    @property
    def n_lanes(self):
        return len(self.devices)

    def cache_create(self, capacity_bytes, lane=0):
        return hip.cache_create(capacity_bytes, device=self.devices[lane % len(self.devices)])

    # page-locked host memory (Reader.tofile's pieces, the compressed bytes of slices): from a pool that outlives the call
    host_buffer = staticmethod(hip.HostBuffer)
    host_buffer_take = staticmethod(hip.pinned_pool.take)
    host_buffer_give = staticmethod(hip.pinned_pool.give)

    cache_destroy = staticmethod(hip.cache_destroy)
    cache_query = staticmethod(hip.cache_query)
    cache_read_rows = staticmethod(hip.cache_read_rows)
    cache_read_slices = staticmethod(hip.cache_read_slices)

    def run_lanes(self, fn, n):
        """fn(0) ... fn(n - 1), each on a host thread of its own (ctypes releases the GIL): the threads are the codec's, made on
        first use and kept -- a ThreadPool per call cost a thread start and a join per device and call."""
        if n <= 1:
            if n:
                fn(0)
            return
        with self._pool_lock:
            if self._pool is None or self._pool_size < n:
                if self._pool is not None:
                    self._pool.close()
                self._pool_size = max(n, len(self.devices))
                self._pool = ThreadPool(self._pool_size)
            pool = self._pool
        pool.map(fn, range(n), chunksize=1)

    def close(self):
        """Ends the lane threads and frees the page-locked buffers the process-wide pool keeps idle for Reader.tofile and the
        slice reads (hip.pinned_pool: up to MTSCOMP_PINNED_KEEP_MB, default 2 GiB, would stay pinned otherwise)."""
        with self._pool_lock:
            if self._pool is not None:
                self._pool.close()
                self._pool = None
        hip.pinned_pool.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_default_codec = None


def get_codec():
    """The process-wide device codec (created on first use; raises without library/GPU)."""
    global _default_codec
    if _default_codec is None:
        _default_codec = HipCodec()
    return _default_codec


def set_codec(codec):
    """Install a codec object (tests inject the CPU oracle here to exercise the host logic without a
    GPU; the package itself never constructs anything but ``HipCodec``)."""
    global _default_codec
    _default_codec = codec


def _int_flags(do_time_diff, do_spatial_diff, chunk_order):
    return hip.make_flags(bool(do_time_diff), bool(do_spatial_diff), chunk_order)


def diff_along_axis(chunk, axis=None):
    """np.diff along `axis` keeping the first row/column (mtscomp.py:143-159)."""
    if axis is None:
        return chunk
    assert 0 <= axis < chunk.ndim
    hip.check_dtype(chunk.dtype)
    flags = hip.FLAG_TIME_DIFF if axis == 0 else hip.FLAG_SPATIAL_DIFF
    chunk = np.ascontiguousarray(chunk)
    out = get_codec().delta(chunk, flags)          # order 'C': same layout as the input
    return out.view(chunk.dtype).reshape(chunk.shape)


def cumsum_along_axis(chunk, axis=None):
    """Inverse of diff_along_axis (mtscomp.py:162-169)."""
    if axis is None:
        return chunk
    assert 0 <= axis < chunk.ndim
    hip.check_dtype(chunk.dtype)
    flags = hip.FLAG_TIME_DIFF if axis == 0 else hip.FLAG_SPATIAL_DIFF
    chunk = np.ascontiguousarray(chunk)
    return get_codec().cumsum(chunk, chunk.shape[0], chunk.shape[1], chunk.dtype, flags)


# ------------------------------------------------------------------------------------------------
# Writer (mtscomp.py:216-511)
# ------------------------------------------------------------------------------------------------
class Writer:
    """Compress a raw data file chunk by chunk.

    Keyword arguments as in the reference (chunk_duration, algorithm, comp_level, do_time_diff,
    do_spatial_diff, n_threads, before_check, check_after_compress, chunk_order) plus
    ``batch_chunks`` (chunks per device call) and ``codec`` (defaults to the MI355X codec).
    """

    # configuration keys a Writer keeps as attributes of the same name
    _CONFIG_ATTRS = ('chunk_duration', 'algorithm', 'comp_level', 'do_time_diff', 'do_spatial_diff', 'n_threads',
                     'check_after_compress', 'chunk_order')

    def __init__(self, before_check=None, codec=None, **kwargs):
        self.pool = None
        self.quiet = kwargs.pop('quiet', False)
        self.config = read_config(**kwargs)
        for key in self._CONFIG_ATTRS:
            setattr(self, key, self.config[key])
        assert self.algorithm == 'zlib', "Only zlib is currently supported."
        self.before_check = before_check if before_check is not None else (lambda x: None)
        self.batch_chunks = int(self.config.get('batch_chunks', None) or DEFAULT_BATCH_CHUNKS)
        self._codec = codec

    @property
    def codec(self):
        return self._codec or get_codec()

    def _need(self, given, key, message):
        """A parameter from the call or the configuration; ValueError(message) when neither has it."""
        value = given or self.config.get(key, None)
        if not value:
            raise ValueError(message)
        return value

    def open(self, data_path, sample_rate=None, n_channels=None, dtype=None, offset=None, mmap=True):
        """Open the raw file and lay out the chunks (mtscomp.py:257-322).  A .npy file brings its own shape and dtype
        (arrays of 3 and more dimensions are flattened to (-1, last axis), the header keeps the original shape); a flat file
        needs n_channels and dtype.  ValueError for a missing sample rate / n_channels / dtype, AssertionError for no data."""
        self.data_path = Path(data_path)
        sample_rate = self._need(sample_rate, 'sample_rate', "Please provide a sample rate (-s option in the command-line).")
        if str(data_path).endswith('.npy'):
            data = np.load(data_path, mmap_mode='r')
            self.shape = data.shape
            self.data = data if data.ndim < 3 else np.reshape(data, (-1, data.shape[-1]))
            self.dtype = self.data.dtype
            n_channels = self.data.shape[1]
        else:
            n_channels = self._need(n_channels, 'n_channels', "Please provide n_channels (-n option in the command-line).")
            self.dtype = np.dtype(self._need(dtype, 'dtype', "Please provide a dtype (-d option in the command-line)."))
            self.data = load_raw_data(data_path, n_channels=n_channels, dtype=self.dtype)
            self.shape = self.data.shape
        assert sample_rate > 0 and n_channels > 0
        assert self.data.ndim == 2
        self.sample_rate = float(sample_rate)
        self.n_samples, self.n_channels = self.data.shape
        assert self.n_samples > 0 and self.n_channels > 0
        assert self.n_channels == n_channels
        self.file_size = self.data.size * self.data.itemsize
        logger.info("Opening %s, duration %.1fs, %d channels.", data_path, self.n_samples / self.sample_rate, self.n_channels)
        self._compute_chunk_bounds()
        self.sha1_compressed, self.sha1_uncompressed = hashlib.sha1(), hashlib.sha1()

    def _compute_chunk_bounds(self):
        """mtscomp.py:324-339 (np.round: half to even)."""
        chunk_size = int(np.round(self.chunk_duration * self.sample_rate))
        bounds = list(range(0, self.n_samples, chunk_size))
        if bounds[-1] < self.n_samples:
            bounds.append(self.n_samples)
        self.chunk_bounds = bounds
        self.n_chunks = len(bounds) - 1
        assert bounds[0] == 0 and bounds[-1] == self.n_samples
        # A batch is what one round of device calls processes (the reference: one chunk per thread).
        self.batch_size = max(1, self.batch_chunks * max(1, len(getattr(self.codec, 'devices', [0]))))
        self.n_batches = int(np.ceil(self.n_chunks / self.batch_size))

    def get_cmeta(self):
        """The .ch header (mtscomp.py:341-358): same keys, same value types."""
        return {
            'version': FORMAT_VERSION,
            'algorithm': self.algorithm,
            'comp_level': self.comp_level,
            'do_time_diff': self.do_time_diff,
            'do_spatial_diff': self.do_spatial_diff,
            'dtype': str(np.dtype(self.dtype)),
            'n_channels': self.n_channels,
            'sample_rate': self.sample_rate,
            'chunk_bounds': self.chunk_bounds,
            'chunk_offsets': self.chunk_offsets,
            'chunk_order': self.chunk_order,
            'sha1_compressed': self.sha1_compressed.hexdigest(),
            'sha1_uncompressed': self.sha1_uncompressed.hexdigest(),
            'shape': self.shape,
        }

    def get_chunk(self, chunk_idx):
        """mtscomp.py:360-373."""
        assert 0 <= chunk_idx <= self.n_chunks - 1
        return self.data[self.chunk_bounds[chunk_idx]:self.chunk_bounds[chunk_idx + 1], :]

    def _compress_chunk(self, chunk_idx):
        """(chunk_idx, (raw chunk, compressed bytes)) -- mtscomp.py:375-397; a device batch of one."""
        return chunk_idx, self.compress_batch(chunk_idx, chunk_idx + 1)[chunk_idx]

    def compress_batch(self, first_chunk, last_chunk):
        """Chunks [first_chunk, last_chunk) -> {idx: (raw chunk, compressed bytes)}
        (mtscomp.py:399-423; the per-chunk work of :375-397 happens on the device)."""
        assert 0 <= first_chunk < last_chunk <= self.n_chunks
        ids = list(range(first_chunk, last_chunk))
        chunks = [self.get_chunk(i) for i in ids]
        flags = _int_flags(self.do_time_diff, self.do_spatial_diff, self.chunk_order)
        # the reference never forwards comp_level to zlib (mtscomp.py:394): level 6 always
        cbufs = self.codec.compress(chunks, flags, 6)
        return {i: (c, b) for i, c, b in zip(ids, chunks, cbufs)}

    def write(self, out, outmeta):
        """Write .cbin + .ch; returns csize / raw size (mtscomp.py:425-507)."""
        if not out:
            out = self.data_path.with_suffix('.c' + self.data_path.suffix[1:])
        if not outmeta:
            outmeta = self.data_path.with_suffix('.ch')
        Path(out).parent.mkdir(exist_ok=True, parents=True)
        self.chunk_offsets = [0]
        logger.info("Starting compression on %s.", getattr(self.codec, 'name', 'codec'))
        # The two SHA-1s and the file write are host work in file order (mtscomp.py:477-483).  One SHA-1 over the whole raw
        # file (~1.2 GB/s per core) is the slowest thing in a file-to-file run, so it gets its own thread from the start
        # (it needs nothing from the devices: hashing chunk after chunk is hashing the file); a second thread writes and
        # hashes the compressed chunks of the previous batch while the devices work on the next one (hashlib and ctypes
        # release the GIL).
        state = {'offset': 0}

        def hash_raw():
            for idx in range(self.n_chunks):
                self.sha1_uncompressed.update(np.ascontiguousarray(self.get_chunk(idx)))

        def consume(done):
            for idx in sorted(done.keys()):                     # strictly in file order
                _, cbuf = done[idx]
                fb.write(cbuf)
                state['offset'] += len(cbuf)
                self.chunk_offsets.append(state['offset'])
                self.sha1_compressed.update(cbuf)

        with open(out, 'wb') as fb, ThreadPool(2) as sink:
            raw_job = sink.apply_async(hash_raw)
            pending = None
            for batch in range(self.n_batches):
                first = self.batch_size * batch
                last = min(self.batch_size * (batch + 1), self.n_chunks)
                done = self.compress_batch(first, last)
                assert set(done.keys()) <= set(range(first, last))
                if pending is not None:
                    pending.get()                               # keep at most one batch in flight behind the devices
                pending = sink.apply_async(consume, (done,))
            if pending is not None:
                pending.get()
            raw_job.get()
            csize = fb.tell()
        assert self.chunk_offsets[-1] == csize
        ratio = csize / self.file_size
        logger.info("Wrote %s (%.1f GB, -%.3f%%).", out, csize / 1024 ** 3, 100 - 100 * ratio)
        with open(outmeta, 'w') as f:
            json.dump(self.get_cmeta(), f, indent=2, sort_keys=True)
        if self.check_after_compress:
            self.before_check(self)
            try:
                check(self.data, out, outmeta, codec=self._codec)
            except AssertionError:
                raise RuntimeError(
                    "CRITICAL ERROR: automatic check failed when compressing the data. "
                    "Report immediately to " + CRITICAL_ERROR_URL)
            logger.debug("Automatic integrity check after compression PASSED.")
        return ratio

    def close(self):
        """mtscomp.py:509-511."""
        mm = getattr(self.data, '_mmap', None)
        if mm is not None:
            mm.close()


# ------------------------------------------------------------------------------------------------
# Reader (mtscomp.py:514-859)
# ------------------------------------------------------------------------------------------------
def _join_rows(chunks):
    """np.concatenate(chunks, axis=0) -- without the copy when the chunks already lie back to back in memory
    (the arrays of one codec call do)."""
    if len(chunks) == 1:
        return chunks[0]
    addr = [c.__array_interface__['data'][0] for c in chunks]
    if all(c.flags.c_contiguous and c.dtype == chunks[0].dtype and c.shape[1] == chunks[0].shape[1] for c in chunks) and \
            all(addr[k] + chunks[k].nbytes == addr[k + 1] for k in range(len(chunks) - 1)) and \
            all(c.base is not None and c.base is chunks[0].base for c in chunks):
        total = sum(c.shape[0] for c in chunks)
        return np.lib.stride_tricks.as_strided(chunks[0], shape=(total, chunks[0].shape[1]))
    return np.concatenate(chunks, axis=0)


def _unlink_lazily(path):
    """path.unlink() whose page-cache work happens off the caller's thread: the name is gone on return (a new file of that name
    is a new file), the old file's pages -- a few hundred thousand of them for a recording on a RAM-backed file system, freed one
    by one when its last reference goes -- are released by a thread that only closes a descriptor."""
    try:
        fd = os.open(str(path), os.O_RDONLY)
    except OSError:
        Path(path).unlink()
        return
    try:
        Path(path).unlink()
    finally:
        threading.Thread(target=os.close, args=(fd,), daemon=True).start()


class Reader:
    """NumPy-style read access to a compressed file; chunks are decoded on the device."""

    def __init__(self, codec=None, **kwargs):
        self.pool = None
        self.cdata = None
        self.quiet = kwargs.pop('quiet', False)
        self.config = read_config(**kwargs)
        self.cache_size = self.config.cache_size
        self.check_after_decompress = self.config.check_after_decompress
        self.batch_chunks = int(self.config.get('batch_chunks', None) or DEFAULT_BATCH_CHUNKS)
        # partial_decode=True (off by default): Reader[rows, :k] may inflate only the prefix of a chunk's stream its leading
        # channels need.  The reference inflates -- and adler32-checks -- the whole chunk on every read (mtscomp.py:618-621);
        # a prefix decode cannot, so it is the caller's explicit choice.
        self.partial_decode = bool(self.config.get('partial_decode', False))
        self._codec = codec
        self._cache = OrderedDict()
        self._dev_caches = None                                   # decoded-chunk caches in HBM, one per lane of the codec (chunk k -> lane k mod lanes)
        self._dev_cache_lock = threading.Lock()
        self._cache_lock = threading.RLock()                      # the host LRU is shared by the threads that slice (mtscomp.py:648)
        self._dev_cache_bytes = int(float(os.environ.get('MTSCOMP_DEVICE_CACHE_GB', DEFAULT_DEVICE_CACHE_GB)) * 2 ** 30)
        self._pin = None                                          # page-locked buffer the compressed bytes of slices are read into
        self._ra, self._ra_calls, self._ra_pending = 1, 0, {}     # read-ahead of the device cache (see _read_ahead)
        self._ra_lock = threading.Lock()                          # (slices may come from several threads)
        self._pin_lock = threading.Lock()
        self._map_lock = threading.Lock()
        self._cmap = self._cmap_view = None
        self._cmap_size = -1
        self._io_pool = None

    @property
    def codec(self):
        return self._codec or get_codec()

    # header fields a Reader keeps as attributes of the same name
    _HEADER_ATTRS = ('n_channels', 'sample_rate', 'chunk_offsets', 'chunk_bounds', 'chunk_order')

    def open(self, cdata, cmeta=None):
        """Attach a compressed file (path or open binary file object) and its header: a dict, a path, or -- by default --
        the .ch file next to a .cbin path (mtscomp.py:536-580)."""
        header = cmeta if cmeta is not None else Path(cdata).with_suffix('.ch')
        if not isinstance(header, dict):
            header = json.loads(Path(header).read_text())
        assert isinstance(header, dict)
        self.cmeta = Bunch(header)
        for key in self._HEADER_ATTRS:
            setattr(self, key, self.cmeta[key])
        self.dtype = np.dtype(self.cmeta.dtype)
        self.n_chunks = len(self.chunk_bounds) - 1
        self.n_samples = self.chunk_bounds[-1]
        self.shape, self.ndim = (self.n_samples, self.n_channels), 2
        self.batch_size = max(1, self.batch_chunks * max(1, len(getattr(self.codec, 'devices', [0]))))
        self.n_batches = int(np.ceil(self.n_chunks / self.batch_size))
        if isinstance(cdata, (str, Path)):
            if Path(cdata).suffix in ('.bin', '.dat'):  # pragma: no cover
                logger.error("File to decompress has unexpected extension %s.", Path(cdata).suffix)
            cdata = open(cdata, 'rb')
        self.cdata = cdata
        self.set_cache_size()

    def set_cache_size(self, cache_size=None):
        """LRU size for decoded chunks (mtscomp.py:582-588)."""
        cache_size = cache_size or self.cache_size
        assert cache_size > 0
        with self._cache_lock:
            self.cache_size = cache_size
            while len(self._cache) > self.cache_size:
                self._cache.popitem(last=False)

    def iter_chunks(self, first_chunk=0, last_chunk=None):
        """Yield (chunk_idx, chunk_start, chunk_length) (mtscomp.py:590-600)."""
        last_chunk = last_chunk if last_chunk is not None else self.n_chunks - 1
        for idx in range(first_chunk, last_chunk + 1):
            yield idx, self.chunk_offsets[idx], self.chunk_offsets[idx + 1] - self.chunk_offsets[idx]

    def _pread(self, length, start):
        if hasattr(os, 'pread'):
            buf = os.pread(self.cdata.fileno(), length, start)       # thread safe (mtscomp.py:609)
            if len(buf) < length:                                    # one pread returns at most 2 GiB - 4 KiB on Linux
                parts = [buf]
                got = len(buf)
                while got < length:
                    more = os.pread(self.cdata.fileno(), length - got, start + got)
                    if not more:
                        break
                    parts.append(more)
                    got += len(more)
                buf = b''.join(parts)
        else:  # pragma: no cover
            with _seek_lock:
                self.cdata.seek(start)
                buf = self.cdata.read(length)
        assert len(buf) == length
        return buf

    def _map_range(self, length, start):
        """`length` bytes at `start` of the compressed file as a uint8 view of a read-only MAPPING of the file: no read call and no
        copy on this side -- the library's host threads copy the bytes out of the page cache into its page-locked pieces while
        the DMA of the piece before runs (mts_cache_*: staged copies).  A cold window's two chunks were 1.1 ms of preadv on eight
        Python threads plus 0.65 ms of DMA before (round 6).  None when the file cannot be mapped, or the range is not inside it
        (the caller's pread then fails the way the reference's does, mtscomp.py:609-612)."""
        if not MAP_CDATA or length <= 0:
            return None
        try:
            with self._map_lock:
                size = os.fstat(self.cdata.fileno()).st_size
                if start + length > size:
                    return None
                if self._cmap is None or self._cmap_size != size:
                    import mmap
                    self._cmap_view = None
                    if self._cmap is not None:
                        try:
                            self._cmap.close()
                        except (BufferError, ValueError):
                            pass                                     # (a view of the old mapping is still out: it goes with its last user)
                    self._cmap = mmap.mmap(self.cdata.fileno(), 0, access=mmap.ACCESS_READ)
                    self._cmap_size = size
                    self._cmap_view = np.frombuffer(self._cmap, dtype=np.uint8)
                return self._cmap_view[start:min(start + length + 16, size)]      # (16 spare bytes where the file has them: the wrapper wants them behind the last chunk and would copy the range to get them)
        except (OSError, ValueError, AttributeError):
            return None

    def _pread_pinned(self, length, start):
        """`length` bytes at `start` of the compressed file as a uint8 view of a page-locked buffer (the codec's: the bytes go to
        the device by DMA from where the file system put them -- no fresh pages to fault in for every read, no staging copy);
        reads of a few MB and more are split over PREAD_THREADS threads.  The caller holds self._pin_lock while the view is in use.
        None when the codec has no such buffers (the caller reads into a bytes object then)."""
        alloc = getattr(self.codec, 'host_buffer_take', None) or getattr(self.codec, 'host_buffer', None)
        if alloc is None or not hasattr(os, 'preadv'):
            return None
        if self._pin is None or self._pin.nbytes < length + 64:
            try:
                new = alloc(max(int((length + 64) * 1.5), 16 << 20))
            except Exception:  # noqa: BLE001
                return None
            if self._pin is not None:
                (getattr(self.codec, 'host_buffer_give', None) or (lambda h: h.free()))(self._pin)
            self._pin = new
        mv = memoryview(self._pin.array)
        fd = self.cdata.fileno()

        def part(a, b):
            while a < b:
                got = os.preadv(fd, [mv[a:b]], start + a)
                if got <= 0:
                    break
                a += got
            return a
        if length >= (4 << 20):
            nt = PREAD_THREADS
            if self._io_pool is None:
                self._io_pool = ThreadPool(nt)
            per = (length + nt - 1) // nt
            ends = self._io_pool.starmap(part, [(k * per, min((k + 1) * per, length)) for k in range(nt) if k * per < length])
            assert all(e == min((k + 1) * per, length) for k, e in enumerate(ends))
        else:
            assert part(0, length) == length
        self._pin.array[length:length + 16] = 0                  # (the kernels may read a few bytes past a stream)
        return self._pin.array[:length + 16]

    def _flags(self):
        return _int_flags(self.cmeta.do_time_diff, self.cmeta.do_spatial_diff, self.chunk_order)

    def _decode(self, triples):
        """Decode chunks [(idx, start, length)] not in the cache with ONE codec call; returns
        {idx: array}.  Error mapping as mtscomp.py:618-628."""
        with self._cache_lock:
            return self._decode_locked(triples)

    def _decode_locked(self, triples):
        todo = [t for t in triples if t[0] not in self._cache]
        result = {}
        if todo:
            rows = [self.chunk_bounds[i + 1] - self.chunk_bounds[i] for (i, _, _) in todo]
            consecutive = all(todo[k][1] + todo[k][2] == todo[k + 1][1] for k in range(len(todo) - 1))
            if consecutive and len(todo) > 1 and getattr(self.codec, 'takes_ranges', False):
                base = todo[0][1]                              # one read for the whole byte range
                buf = self._pread(todo[-1][1] + todo[-1][2] - base, base)
                cbufs = (buf, [t[1] - base for t in todo], [t[2] for t in todo])
            else:
                cbufs = [self._pread(length, start) for (_, start, length) in todo]
            status, arrays = self.codec.decompress(cbufs, rows, self.n_channels, self.dtype, self._flags())
            for (idx, _, _), st, arr in zip(todo, status, arrays):
                if st == hip.CHUNK_BADSIZE:
                    raise AssertionError("Chunk #%d does not have the expected size." % idx)
                if st != hip.CHUNK_OK:
                    raise IOError("Compressed chunk #%d is corrupted." % idx)
                result[idx] = arr
        for idx, _, _ in triples:
            if idx in self._cache:
                self._cache.move_to_end(idx)
                result[idx] = self._cache[idx]
        for idx, arr in result.items():
            self._cache[idx] = arr
            self._cache.move_to_end(idx)
        self._trim_cache()
        return result

    def _trim_cache(self):
        """Back to cache_size entries.  The arrays of one codec call are views of one buffer: entries that would keep a
        buffer several times the size of everything cached from it alive (what is left of a big batch) are replaced by
        copies of themselves; views that still account for most of their buffer stay views (no copy, and neighbours keep
        sharing a base, which lets a slice over several cached chunks be joined without copying)."""
        while len(self._cache) > self.cache_size:
            self._cache.popitem(last=False)
        roots = {}
        for idx, arr in self._cache.items():
            root = arr
            while isinstance(root.base, np.ndarray):
                root = root.base
            if root is not arr:
                entry = roots.setdefault(id(root), [root, 0, []])
                entry[1] += arr.nbytes
                entry[2].append(idx)
        for root, cached_bytes, ids in roots.values():
            if root.nbytes > 2 * cached_bytes:
                for idx in ids:
                    self._cache[idx] = self._cache[idx].copy()

    # -- the decoded-chunk caches in HBM: one per lane of the codec (a HipCodec has a lane per device), chunk k on lane k mod lanes
    def _n_lanes(self):
        return max(1, int(getattr(self.codec, 'n_lanes', 1)))

    def _cache_for(self, lane):
        """The cache id of `lane` (created on first use; every lane gets the capacity the Reader was given)."""
        with self._dev_cache_lock:                             # slices may come from several threads (mtscomp.py:422, :648)
            if self._dev_caches is None:
                self._dev_caches = [None] * self._n_lanes()
            if self._dev_caches[lane] is None:
                if len(self._dev_caches) > 1:
                    self._dev_caches[lane] = self.codec.cache_create(self._dev_cache_bytes, lane)
                else:
                    self._dev_caches[lane] = self.codec.cache_create(self._dev_cache_bytes)
            return self._dev_caches[lane]

    @property
    def _dev_cache(self):
        """Lane 0's cache id, None before its first use."""
        return self._dev_caches[0] if self._dev_caches else None

    @staticmethod
    def _raise_for(status_by_chunk):
        """The reference's errors for the first chunk (in file order) that did not decode (mtscomp.py:618-628)."""
        for k in sorted(status_by_chunk):
            st = status_by_chunk[k]
            if st == hip.CHUNK_BADSIZE:
                raise AssertionError("Chunk #%d does not have the expected size." % k)
            if st != hip.CHUNK_OK:
                raise IOError("Compressed chunk #%d is corrupted." % k)

    def _slice_from_device_cache(self, first, last, i0, i1):
        """Rows [i0, i1) -- inside chunks first..last -- through the codec's decoded-chunk cache in HBM: chunks that are
        not resident are read from the file and decoded in one batch (and stay on the device), and only the requested
        rows come back.  None when the slice should take the host path (no device cache, slice too large for it, or the
        chunks are already in the host LRU of read_chunk)."""
        n = last - first + 1
        if not getattr(self.codec, 'device_cache', False) or self._dev_cache_bytes <= 0 or n > DEVICE_CACHE_MAX_CHUNKS:
            return None
        rows = [self.chunk_bounds[i + 1] - self.chunk_bounds[i] for i in range(first, last + 1)]
        if 2 * sum(rows) * self.n_channels * self.dtype.itemsize > self._dev_cache_bytes:
            return None
        if all(i in self._cache for i in range(first, last + 1)):
            return None
        lanes = self._n_lanes()
        if lanes > 1:
            return self._slice_from_lane_caches(first, last, i0, i1, lanes)
        cache = self._cache_for(0)
        keys = list(range(first, last + 1))
        a, b = i0 - self.chunk_bounds[first], i1 - self.chunk_bounds[first]
        present = [int(p) >= self.n_channels for p in self.codec.cache_query(cache, keys)]      # (entries of leading channels only do not count)
        ahead = self._read_ahead(cache, keys, present)              # chunks behind the slice decoded along with its missing ones
        if ahead:
            keys, present = keys + ahead, present + [False] * len(ahead)
            rows = rows + [self.chunk_bounds[k + 1] - self.chunk_bounds[k] for k in ahead]
            n += len(ahead)
        for attempt in range(2):
            need = [k for k, p in zip(keys, present) if not p]
            offs, lens = [0] * n, [0] * n
            base = self.chunk_offsets[need[0]] if need else 0      # one read from the first to the last missing chunk
            nbytes = self.chunk_offsets[need[-1] + 1] - base if need else 0
            for k in need:
                offs[k - first] = self.chunk_offsets[k] - base
                lens[k - first] = self.chunk_offsets[k + 1] - self.chunk_offsets[k]
            # (the page-locked buffer is the Reader's: held from the read until the codec has taken the bytes, released whatever
            #  either of them raises -- a short read of a truncated file is the reference's AssertionError, not a lock left behind)
            buf = self._map_range(nbytes, base) if need else b''     # (a view of the file's mapping: nothing to lock, nothing read here)
            if buf is not None:
                try:
                    status, out = self.codec.cache_read_rows(cache, keys, buf, offs, lens, rows, self.n_channels,
                                                             self.dtype, self._flags(), a, b)
                    break
                except hip.HipError as e:
                    if e.code != hip.E_MISS or attempt:
                        raise
                    present = [False] * n                          # dropped since the query: send everything
                    continue
            with self._pin_lock:
                buf = self._pread_pinned(nbytes, base) if need else b''
                if buf is None:
                    buf = self._pread(nbytes, base)
                try:
                    status, out = self.codec.cache_read_rows(cache, keys, buf, offs, lens, rows, self.n_channels,
                                                             self.dtype, self._flags(), a, b)
                    break
                except hip.HipError as e:
                    if e.code != hip.E_MISS or attempt:
                        raise
                    present = [False] * n                          # dropped since the query: send everything
        with self._ra_lock:
            for k, st in zip(keys[n - len(ahead):], status[n - len(ahead):]) if ahead else ():
                if st == 0:
                    self._ra_pending[k] = True                     # (a damaged chunk ahead is reported when somebody reads it)
        self._raise_for(dict(zip(keys[:n - len(ahead)], status[:n - len(ahead)])))
        return out

    def _read_ahead(self, cache, keys, present):
        """Chunks right behind a slice that are decoded in the same device call as the slice's missing chunks -- a batch of one
        or two chunks costs the device nearly what a batch of four does (the inflate stages of a small batch are one dependency
        chain each), so a reader that comes to them later (the next window of a sequential reader; sooner or later every chunk,
        for random windows over a recording that fits the cache) finds them decoded.  Adaptive like a file system's read-ahead:
        starts at one chunk, one more (up to READ_AHEAD_MAX) whenever a chunk read ahead is used, one less whenever one leaves
        the list of pending ones unused (the list is half the cache long at most); at none, one chunk is tried every 32nd time.
        Nothing is read ahead when all of a slice's chunks are resident."""
        with self._ra_lock:
            for k, p in zip(keys, present):
                if p and self._ra_pending.pop(k, None):
                    self._ra = min(self._ra + 1, READ_AHEAD_MAX)
            if all(present) or READ_AHEAD_MAX <= 0:
                return []
            self._ra_calls += 1
            ra = self._ra if self._ra > 0 else (1 if self._ra_calls % 32 == 0 else 0)
        chunk_bytes = max(1, (self.chunk_bounds[1] - self.chunk_bounds[0]) * self.n_channels * self.dtype.itemsize)
        cap = max(8, min(self._dev_cache_bytes // chunk_bytes // 2, 1024))
        ra = min(ra, max(0, DEVICE_CACHE_MAX_CHUNKS - len(keys)), int(self._dev_cache_bytes // chunk_bytes // 4))
        cand = list(range(keys[-1] + 1, min(keys[-1] + 1 + ra, self.n_chunks)))
        if not cand:
            return []
        ahead = []
        for k, p in zip(cand, self.codec.cache_query(cache, cand)):
            if int(p) >= self.n_channels:
                break                                              # (one read covers the missing chunks: it ends at the first resident one)
            ahead.append(k)
        with self._ra_lock:
            while self._ra_pending and len(self._ra_pending) + len(ahead) > cap:      # the oldest pending ones have had their chance
                self._ra_pending.pop(next(iter(self._ra_pending)))
                self._ra = max(self._ra - 1, 0)
        return ahead

    def _slice_from_lane_caches(self, first, last, i0, i1, lanes):
        """The same over several lanes: every lane reads, decodes and keeps its own chunks (k mod lanes) and copies the rows of
        [i0, i1) they hold straight into their place in the result, all lanes at once."""
        out = np.empty((i1 - i0, self.n_channels), dtype=self.dtype)
        keys = list(range(first, last + 1))
        status = {}

        def run(g):
            cache = self._cache_for(g)
            for k in keys[(g - first) % lanes::lanes]:
                c0, c1 = self.chunk_bounds[k], self.chunk_bounds[k + 1]
                lo, hi = max(i0, c0), min(i1, c1)
                present = int(self.codec.cache_query(cache, [k])[0]) >= self.n_channels
                for attempt in range(2):
                    length = 0 if present else self.chunk_offsets[k + 1] - self.chunk_offsets[k]
                    buf = self._pread(length, self.chunk_offsets[k]) if length else b''
                    try:
                        st, _ = self.codec.cache_read_rows(cache, [k], buf, [0], [length], [c1 - c0], self.n_channels, self.dtype,
                                                           self._flags(), lo - c0, hi - c0, out=out[lo - i0:hi - i0])
                        break
                    except hip.HipError as e:
                        if e.code != hip.E_MISS or attempt:
                            raise
                        present = False                            # dropped since the query: send the bytes
                status[k] = st[0]
        owners = sorted({k % lanes for k in keys})
        self.codec.run_lanes(lambda j: run(owners[j]), len(owners))
        self._raise_for(status)
        return out

    def _gather_request(self, item):
        """(i0, i1, row_step, c0, c1, col_step, squeeze) when the index is a rectangle the device gather serves, else None."""
        if isinstance(item, slice):
            item = (item, slice(None))
        if not (isinstance(item, tuple) and len(item) == 2 and isinstance(item[0], slice)):
            return None
        rs = 1 if item[0].step is None else item[0].step
        if not isinstance(rs, (int, np.integer)) or rs < 1:
            return None
        cols, squeeze = item[1], False
        if isinstance(cols, (int, np.integer)):
            c = int(cols) + (self.n_channels if cols < 0 else 0)
            if not 0 <= c < self.n_channels:
                return None                                    # (numpy raises IndexError: let it)
            c0, c1, cs, squeeze = c, c + 1, 1, True
        elif isinstance(cols, slice):
            c0, c1, cs = cols.indices(self.n_channels)
            if cs < 1:
                return None
            c1 = max(c0, c1)
        else:
            return None
        i0 = self._validate_index(item[0].start, 0)
        i1 = self._validate_index(item[0].stop, self.n_samples)
        return i0, max(i0, i1), int(rs), c0, c1, cs, squeeze

    def read_slices(self, items, _fallback=True):
        """Several index expressions ``r[rows, columns]`` / ``r[rows]`` in ONE device call per lane: the chunks they touch are
        decoded (or found in the decoded-chunk cache in HBM), the requested rows and columns are gathered on the device and only
        they cross the bus (the reference decodes whole chunks and slices on the host, mtscomp.py:835-842).  Returns the list
        of arrays; items the device gather does not serve go through ``__getitem__`` one by one."""
        reqs = [self._gather_request(it) for it in items]
        usable = getattr(self.codec, 'device_cache', False) and hasattr(self.codec, 'cache_read_slices') and \
            self._dev_cache_bytes > 0 and all(r is not None for r in reqs)
        spans = []
        lanes = self._n_lanes()
        if usable:
            for i0, i1, _, _, _, _, _ in reqs:
                if i1 <= i0:
                    spans.append(None)
                    continue
                first, last = self._chunks_for_interval(i0, i1)
                if last > first and self.chunk_bounds[last] >= i1:
                    last -= 1
                spans.append((first, last))
            keys = sorted({k for sp in spans if sp for k in range(sp[0], sp[1] + 1)})
            rows = [self.chunk_bounds[k + 1] - self.chunk_bounds[k] for k in keys]
            # the limits are PER LANE (chunk k lives on lane k mod lanes: every other chunk with two lanes lands on one of them)
            per_lane = [[r for k, r in zip(keys, rows) if k % lanes == g] for g in range(lanes)]
            usable = len(keys) > 0 and all(len(rs) <= DEVICE_CACHE_MAX_CHUNKS and
                                           2 * sum(rs) * self.n_channels * self.dtype.itemsize <= self._dev_cache_bytes for rs in per_lane)
        if not usable:
            return [self[it] for it in items] if _fallback else None
        # Requests that stay within the leading channels of channel-major chunks need only a prefix of every chunk's stream:
        # the codec inflates a chunk that is not resident just that far, from a prefix of its compressed bytes (the channels
        # compress about equally: the share of the bytes plus a margin; if that falls short the codec says so and the whole
        # chunk is sent).  The reference inflates whole chunks and drops the columns on the host (mtscomp.py:835-842).
        # Only with partial_decode=True: such a read cannot check the chunk's adler32 (the reference always does, mtscomp.py:618-621).
        n_lead = max([r[4] for r in reqs] or [self.n_channels])     # (over ALL requests: the codec checks every one against it)
        leading = self.partial_decode and bool(getattr(self.codec, 'leading_channels', False)) and self.chunk_order == 'F' and \
            self.dtype.kind in 'iu' and 0 < 2 * n_lead <= self.n_channels
        if lanes == 1:
            where = {k: j for j, k in enumerate(keys)}
            cum = np.concatenate(([0], np.cumsum(rows)))
            requests = []
            for (i0, i1, rs, c0, c1, cs, _), sp in zip(reqs, spans):
                a = int(cum[where[sp[0]]]) + i0 - self.chunk_bounds[sp[0]] if sp else 0
                requests.append((a, a + (i1 - i0), rs, c0, c1, cs))
            status, arrays = self._lane_read_slices(self._cache_for(0), keys, requests, n_lead if leading else None)
            self._raise_for(dict(zip(keys, status)))
            return [a[:, 0] if r[6] else a for a, r in zip(arrays, reqs)]
        # several lanes: a request is cut where it crosses from one chunk into the next; every lane gathers the pieces that lie in
        # its chunks (k mod lanes) in one call, all lanes at once; the pieces of a request are joined on the host
        lane_keys = {g: [k for k in keys if k % lanes == g] for g in sorted({k % lanes for k in keys})}
        lane_reqs = {g: [] for g in lane_keys}                      # per lane: (request index, piece index, request tuple)
        n_pieces = []
        for q, ((i0, i1, rs, c0, c1, cs, _), sp) in enumerate(zip(reqs, spans)):
            pieces = 0
            for k in (range(sp[0], sp[1] + 1) if sp else ()):
                b0, b1 = self.chunk_bounds[k], self.chunk_bounds[k + 1]
                lo, hi = max(i0, b0), min(i1, b1)
                start = i0 + -(-(lo - i0) // rs) * rs               # the first row of the request's grid inside the chunk
                if start >= hi:
                    continue
                g = k % lanes
                before = sum(self.chunk_bounds[j + 1] - self.chunk_bounds[j] for j in lane_keys[g] if j < k)      # rows of the lane's chunks in front
                lane_reqs[g].append((q, pieces, (before + start - b0, before + hi - b0, rs, c0, c1, cs)))
                pieces += 1
            n_pieces.append(pieces)
        parts = [[None] * n for n in n_pieces]
        status = {}
        owners = list(lane_keys)

        def run(j):
            g = owners[j]
            st, arrays = self._lane_read_slices(self._cache_for(g), lane_keys[g], [r for _, _, r in lane_reqs[g]], n_lead if leading else None)
            status.update(zip(lane_keys[g], st))
            for (q, piece, _), arr in zip(lane_reqs[g], arrays):
                parts[q][piece] = arr
        self.codec.run_lanes(run, len(owners))
        self._raise_for(status)
        out = []
        for q, r in enumerate(reqs):
            ncol = len(range(r[3], r[4], r[5]))
            arr = parts[q][0] if len(parts[q]) == 1 else np.concatenate(parts[q], axis=0) if parts[q] else np.zeros((0, ncol), dtype=self.dtype)
            out.append(arr[:, 0] if r[6] else arr)
        return out

    def _lane_read_slices(self, cache, keys, requests, n_lead):
        """One cache_read_slices call on one lane's cache: `requests` index the concatenation of the chunks `keys`; the bytes of the
        chunks that are not resident are read first (one read per run of neighbours).  n_lead: the requests stay below that many
        leading channels and a chunk that is not resident is decoded only that far, from a prefix of its bytes; if that falls short
        -- or an entry was dropped between the query and the call -- everything is sent once more, whole.  -> (status, arrays)."""
        where = {k: j for j, k in enumerate(keys)}
        rows = [self.chunk_bounds[k + 1] - self.chunk_bounds[k] for k in keys]
        held = self.codec.cache_query(cache, keys)                  # channels every resident entry holds (0: not resident)
        present = [int(h) >= (n_lead or self.n_channels) for h in held]      # bytes are sent for entries that are too narrow only
        for attempt in range(2):
            leading = bool(n_lead) and attempt == 0
            offs, lens, parts, at = [0] * len(keys), [0] * len(keys), [], 0
            need = [k for k, p in zip(keys, present) if not p]
            j = 0
            while j < len(need):                                # one read per run of neighbouring missing chunks
                e = j
                while not leading and e + 1 < len(need) and need[e + 1] == need[e] + 1:
                    e += 1
                base = self.chunk_offsets[need[j]]
                nbytes = self.chunk_offsets[need[e] + 1] - base
                if leading:                                     # (one chunk: a prefix of its bytes)
                    nbytes = min(nbytes, int(nbytes * n_lead / self.n_channels * 1.3) + 32768)
                parts.append(self._pread(nbytes, base))
                for k in need[j:e + 1]:
                    offs[where[k]] = at + self.chunk_offsets[k] - base
                    lens[where[k]] = min(self.chunk_offsets[k + 1] - self.chunk_offsets[k], nbytes)
                at += len(parts[-1])
                j = e + 1
            try:
                # the second attempt is a plain whole-chunk decode (every check runs: a chunk the prefix decoder cannot follow, or
                # one damaged inside the prefix, gets the reference's verdict instead of a miss)
                extra = {'n_leading': n_lead} if leading else {}
                return self.codec.cache_read_slices(cache, keys, b''.join(parts), offs, lens, rows,
                                                    self.n_channels, self.dtype, self._flags(), requests, **extra)
            except hip.HipError as e:
                if e.code != hip.E_MISS or attempt:
                    raise
                present = [False] * len(keys)                  # dropped since the query (or a prefix fell short): send everything

    def _read_range(self, b0, b1):
        """The compressed bytes of chunks b0 .. b1-1 in one read."""
        base = self.chunk_offsets[b0]
        return self._pread(self.chunk_offsets[b1] - base, base)

    def _decode_into(self, b0, b1, dst, buf=None, lane=None):
        """Chunks b0 .. b1-1 (consecutive in the file) decoded straight into `dst` (their rows, C-contiguous): one read (or
        the bytes read ahead by the caller), one codec call (on `lane` of the codec, if given), no copy on the host and
        nothing left in the chunk cache."""
        base = self.chunk_offsets[b0]
        if buf is None:
            buf = self._read_range(b0, b1)
        offs = [self.chunk_offsets[i] - base for i in range(b0, b1)]
        lens = [self.chunk_offsets[i + 1] - self.chunk_offsets[i] for i in range(b0, b1)]
        rows = [self.chunk_bounds[i + 1] - self.chunk_bounds[i] for i in range(b0, b1)]
        extra = {} if lane is None else {'lane': lane}
        status, _ = self.codec.decompress((buf, offs, lens), rows, self.n_channels, self.dtype, self._flags(), out=dst, **extra)
        self._raise_for(dict(zip(range(b0, b1), status)))

    def read_chunk(self, chunk_idx, chunk_start, chunk_length):
        """One decoded chunk, (n_samples_chunk, n_channels), C-contiguous (mtscomp.py:602-635)."""
        return self._decode([(chunk_idx, chunk_start, chunk_length)])[chunk_idx]

    def _decompress_chunk(self, chunk_idx):
        """mtscomp.py:637-643."""
        assert 0 <= chunk_idx <= self.n_chunks - 1
        start = self.chunk_offsets[chunk_idx]
        return chunk_idx, self.read_chunk(chunk_idx, start, self.chunk_offsets[chunk_idx + 1] - start)

    def decompress_chunks(self, chunk_ids, pool=None):
        """{chunk_idx: array} for the requested chunks, decoded in one device batch
        (mtscomp.py:645-650; `pool` is accepted for compatibility and not needed)."""
        ids = list(chunk_ids)
        triples = [(i, self.chunk_offsets[i], self.chunk_offsets[i + 1] - self.chunk_offsets[i]) for i in ids]
        with self._cache_lock:
            keep = self.cache_size
            if len(ids) > keep:
                self.cache_size = len(ids)          # a batch must not evict itself
            try:
                out = self._decode(triples)
            finally:
                self.cache_size = keep
                self._trim_cache()
        assert set(out.keys()) == set(ids)
        return out

    def _validate_index(self, i, value_for_none=0):
        """A slice bound as a sample index in [0, n_samples]: None -> the default, negative -> from the end (mtscomp.py:652-659)."""
        if i is None:
            return int(_clip(value_for_none, 0, self.n_samples))
        return int(_clip(i + self.n_samples if i < 0 else i, 0, self.n_samples))

    def _chunks_for_interval(self, i0, i1):
        """First and last chunk to load for samples [i0, i1] -- i1 is treated as inclusive, exactly
        like the reference (mtscomp.py:661-684; table pinned by tests.py:308-339)."""
        i0 = _clip(i0, 0, self.n_samples - 1)
        i1 = _clip(i1, i0, self.n_samples - 1)
        first = _clip(bisect.bisect_right(self.chunk_bounds, i0) - 1, 0, self.n_chunks - 1)
        last = _clip(bisect.bisect_right(self.chunk_bounds, i1, lo=first) - 1, 0, self.n_chunks - 1)
        assert 0 <= first <= last <= self.n_chunks - 1
        return first, last

    def start_thread_pool(self):
        """Kept for API compatibility (mtscomp.py:686-692); the device path does not need it."""
        if self.pool:  # pragma: no cover
            return self.pool
        self.pool = ThreadPool(max(1, min(4, int(self.config.n_threads or 1))))
        return self.pool

    def stop_thread_pool(self):
        """mtscomp.py:694-699."""
        if self.pool:
            self.pool.close()
            self.pool.join()
        self.pool = None

    def tofile(self, out, overwrite=False, in_place=None):
        """Decompress the whole file to a flat binary file (mtscomp.py:701-743).  With `overwrite` an existing file is unlinked
        first and a NEW file written, as the reference does (mtscomp.py:711-715): hard links to the old file, memory maps of it and
        its owner / mode are left alone.  `in_place=True` (or MTSCOMP_TOFILE_IN_PLACE=1) is an opt-in that is NOT the reference's
        behaviour: an existing regular file is written over where it is (same inode, cut to the new length at the end) -- faster
        on a RAM-backed file system, whose pages are there already; readers of the old file see their data change."""
        if out is None:
            out = Path(self.cdata.name).with_suffix('.bin')
        out = Path(out)
        if not overwrite and out.exists():  # pragma: no cover
            raise ValueError("The output file %s already exists, use --overwrite or specify another "
                             "output path." % out)
        if in_place is None:
            in_place = os.environ.get('MTSCOMP_TOFILE_IN_PLACE', '0') not in ('', '0')
        direct = getattr(self.codec, 'takes_out', False)
        pipelined = direct and self.n_chunks > 1
        keep_inode = bool(in_place) and pipelined and out.is_file() and not out.is_symlink()
        if overwrite and out.exists() and not keep_inode:
            _unlink_lazily(out)
        if pipelined:
            dsize = self._tofile_pipelined(out, keep_inode)
        else:
            with open(out, 'wb') as fb:
                for batch in range(self.n_batches):
                    first = self.batch_size * batch
                    last = min(self.batch_size * (batch + 1), self.n_chunks)
                    chunks = self.decompress_chunks(range(first, last))
                    for idx in sorted(chunks.keys()):
                        fb.write(chunks[idx])
                dsize = fb.tell()
        assert dsize == self.chunk_bounds[-1] * self.n_channels * self.dtype.itemsize
        logger.info("Wrote %s (%.1f GB).", out, dsize / 1024 ** 3)
        if self.check_after_decompress:
            decompressed = load_raw_data(out, n_channels=self.n_channels, dtype=self.dtype)
            check(decompressed, self.cdata, self.cmeta, codec=self._codec)

    def _tofile_pipelined(self, out, keep_inode=False):
        """The file written piece by piece.  Per lane of the codec (a HipCodec has one per device; piece k goes to lane k mod
        lanes) three things are in flight: the compressed bytes of the lane's next piece being read, a piece on the device
        (decoded straight into one of the lane's two host buffers), the piece before being written by a few threads (pwrite on
        disjoint ranges; file writes, reads and ctypes calls all release the GIL).  A piece is a fraction of a device batch so
        that even a one-batch file overlaps its copies with its writes.  The host buffers are page-locked when the codec has
        such memory -- the decoded rows arrive by DMA, without a copy out of a staging piece -- and come from a pool that outlives
        the call (pinning a few hundred MB takes longer than writing them).  Returns the size of the file."""
        lanes = self._n_lanes() if getattr(self.codec, 'takes_out', False) else 1
        piece = max(1, min(self.batch_size, TOFILE_PIECE_CHUNKS))
        starts = list(range(0, self.n_chunks, piece))
        lanes = max(1, min(lanes, len(starts)))
        row_bytes = self.n_channels * self.dtype.itemsize
        max_rows = max(self.chunk_bounds[min(b0 + piece, self.n_chunks)] - self.chunk_bounds[b0] for b0 in starts)
        n_bufs = min(2 * lanes, len(starts))
        take, give = getattr(self.codec, 'host_buffer_take', None), getattr(self.codec, 'host_buffer_give', None)
        max_cbytes = max(self.chunk_offsets[min(b0 + piece, self.n_chunks)] - self.chunk_offsets[b0] for b0 in starts)
        pinned, pinned_in = [], []
        if take is not None and give is not None and hasattr(os, 'preadv'):
            try:
                for _ in range(n_bufs):
                    pinned.append(take(max_rows * row_bytes))
                for _ in range(n_bufs):                             # the compressed bytes of a piece are read into page-locked memory too:
                    pinned_in.append(take(max_cbytes + 64))          # no fresh pages to fault in per piece, and the DMA reads them where they are
            except Exception:  # noqa: BLE001  (no page-locked memory to be had: pageable buffers do)
                for h in pinned + pinned_in:
                    give(h)
                pinned, pinned_in = [], []
        if pinned:
            bufs = [h.array[:max_rows * row_bytes].view(self.dtype).reshape(max_rows, self.n_channels) for h in pinned]
        else:
            bufs = [np.empty((max_rows, self.n_channels), dtype=self.dtype) for _ in range(n_bufs)]
        n_writers = max(1, int(os.environ.get('MTSCOMP_TOFILE_WRITERS', TOFILE_WRITERS)))
        # a new file (the reference's open(out, 'wb') after its unlink) unless tofile(in_place=True) found a file to write over:
        # that one keeps its pages and is cut to the new length when everything is written
        fd = os.open(str(out), os.O_WRONLY | os.O_CREAT | (0 if keep_inode else os.O_TRUNC), 0o644)
        total_bytes = self.chunk_bounds[-1] * row_bytes

        def write_piece(arr, offset):
            mv = memoryview(arr).cast('B')
            per = (len(mv) + n_writers - 1) // n_writers
            per = max(1 << 20, (per + 4095) & ~4095)

            def one(k):
                a, b = k * per, min((k + 1) * per, len(mv))
                while a < b:
                    a += os.pwrite(fd, mv[a:b], offset + a)
            parts = [k for k in range(n_writers) if k * per < len(mv)]
            if len(parts) == 1:
                one(0)
            else:
                wpool.map(one, parts, chunksize=1)

        def piece_range(k):
            return starts[k], min(starts[k] + piece, self.n_chunks)

        def read_piece(k, into):
            """The compressed bytes of piece k: into the page-locked buffer `into` (-> a view of it, 16 spare bytes behind the last
            chunk zeroed: the kernels may read a few bytes past a stream), or as a bytes object."""
            b0, b1 = piece_range(k)
            if into is None:
                return self._read_range(b0, b1)
            base, n = self.chunk_offsets[b0], self.chunk_offsets[b1] - self.chunk_offsets[b0]
            mv, a = memoryview(into.array), 0
            while a < n:
                got = os.preadv(self.cdata.fileno(), [mv[a:n]], base + a)
                if got <= 0:
                    break
                a += got
            assert a == n
            into.array[n:n + 16] = 0
            return into.array[:n + 16]

        def lane_loop(g):
            mine = list(range(g, len(starts), lanes))            # this lane's pieces
            my_bufs = bufs[g::lanes]
            my_in = pinned_in[g::lanes] or [None]                   # (the read ahead fills the one the device is not reading)
            aux = ThreadPool(2)                                     # (one thread reads ahead, one hands pieces to the writers)
            try:
                nxt = aux.apply_async(read_piece, (mine[0], my_in[0]))
                pending = [None] * len(my_bufs)
                for j, k in enumerate(mine):
                    b0, b1 = piece_range(k)
                    buf = nxt.get()
                    nxt = aux.apply_async(read_piece, (mine[j + 1], my_in[(j + 1) % len(my_in)])) if j + 1 < len(mine) else None
                    slot = j % len(my_bufs)
                    if pending[slot] is not None:
                        pending[slot].get()                         # (the write that used this buffer two pieces ago)
                    rows = self.chunk_bounds[b1] - self.chunk_bounds[b0]
                    dst = my_bufs[slot][:rows]
                    self._decode_into(b0, b1, dst, buf, lane=g if lanes > 1 else None)
                    pending[slot] = aux.apply_async(write_piece, (dst, self.chunk_bounds[b0] * row_bytes))
                for p in pending:
                    if p is not None:
                        p.get()
            finally:
                # also on the way out of a failure (a corrupt chunk raises in _decode_into): a read or a write still in flight is
                # waited for -- ThreadPool's context manager terminate()s without joining, and the caller is about to truncate and
                # close the descriptor and hand the page-locked buffers back to the pool
                aux.close()
                aux.join()

        wpool = ThreadPool(n_writers)
        try:
            try:
                if lanes == 1:
                    lane_loop(0)
                else:
                    self.codec.run_lanes(lane_loop, lanes)
            finally:
                wpool.close()                                       # (every lane has joined its own helpers by now: nothing writes any more)
                wpool.join()
            if os.fstat(fd).st_size > total_bytes:
                os.ftruncate(fd, total_bytes)                       # (what a longer file of that name had behind)
            return os.fstat(fd).st_size
        except BaseException:
            try:
                os.ftruncate(fd, 0)                                 # (a failed write leaves no mixture of the old file and the new)
            except OSError:
                pass
            raise
        finally:
            os.close(fd)
            del bufs
            for h in pinned + pinned_in:
                give(h)

    def close(self):
        """mtscomp.py:745-748."""
        with self._dev_cache_lock:
            caches, self._dev_caches = self._dev_caches or [], None
        for cache in caches:
            if cache is not None:
                try:
                    self.codec.cache_destroy(cache)
                except Exception:  # pragma: no cover
                    pass
        with self._map_lock:
            self._cmap_view = None
            if self._cmap is not None:
                try:
                    self._cmap.close()
                except (BufferError, ValueError):                  # (a view is still out somewhere: the mapping goes with it)
                    pass
                self._cmap = None
        with self._pin_lock:                                      # (a slice on another thread may be reading into the buffer)
            if self._io_pool is not None:
                self._io_pool.close()
                self._io_pool = None
            if self._pin is not None:
                pin, self._pin = self._pin, None
                give = getattr(self.codec, 'host_buffer_give', None)
                (give or (lambda h: h.free()))(pin)
        if self.cdata:
            self.cdata.close()

    def chop(self, n_chunks, out=None):
        """Write the first n_chunks chunks as a new .cbin/.ch pair without decoding anything: the byte range up to
        chunk_offsets[n_chunks], a header cut to match, sha1s dropped and 'chopped' set (mtscomp.py:750-796)."""
        assert n_chunks > 0
        if n_chunks >= self.n_chunks:  # pragma: no cover
            logger.warning("Cannot chop more chunks than there are in the original file.")
            return
        assert out is not None, "The output path must be specified."
        out = Path(out)
        outmeta = out.with_suffix('.ch')
        assert out.suffix == '.cbin'
        for target in (out, outmeta):
            if target.exists():  # pragma: no cover
                raise IOError("File %s already exists." % target)
        out.parent.mkdir(parents=True, exist_ok=True)
        n_bytes = self.chunk_offsets[n_chunks]
        out.write_bytes(self._pread(n_bytes, 0))
        header = dict(self.cmeta)
        header.update(chunk_bounds=self.cmeta['chunk_bounds'][:n_chunks + 1], chunk_offsets=self.cmeta['chunk_offsets'][:n_chunks + 1],
                      sha1_compressed=None, sha1_uncompressed=None, chopped=True)
        assert header['chunk_offsets'][-1] == n_bytes
        outmeta.write_text(json.dumps(header, indent=2, sort_keys=True))

    def __getitem__(self, item):
        """NumPy-style slicing (mtscomp.py:798-856).  All chunks a slice touches are decoded in one
        device batch."""
        fallback = np.zeros((0, self.n_channels), dtype=self.dtype)
        if isinstance(item, slice):
            i0 = self._validate_index(item.start, 0)
            i1 = self._validate_index(item.stop, self.n_samples)
            if i1 <= i0:
                return fallback
            first, last = self._chunks_for_interval(i0, i1)
            # the reference also loads the chunk that starts exactly at i1 (inclusive stop); its rows
            # are never part of the result, so it is not decoded here
            if last > first and self.chunk_bounds[last] >= i1:
                last -= 1
            rows = self._slice_from_device_cache(first, last, i0, i1)
            if rows is not None:
                out = rows[0:i1 - i0:item.step, :]             # (like arr[a:b:step] in the reference: a negative step gives nothing)
                assert out.shape[0] == len(range(i0, i1, item.step or 1))
                return out
            if last - first + 1 > self.batch_size:
                # a long slice: batch after batch straight into the result (one codec call for everything would need the
                # whole slice in device memory, and the chunk cache could not hold it anyway)
                r0 = self.chunk_bounds[first]
                whole = np.empty((self.chunk_bounds[last + 1] - r0, self.n_channels), dtype=self.dtype)   # whole chunks first..last
                direct = getattr(self.codec, 'takes_out', False)
                per_call = max(1, self.batch_chunks) if direct else self.batch_size      # (direct: a call is one lane's, i.e. one device's)
                starts = list(range(first, last + 1, per_call))
                lanes = max(1, min(self._n_lanes(), len(starts))) if direct else 1

                def lane_loop(g):
                    mine = starts[g::lanes]
                    ahead = ThreadPool(1) if direct and len(mine) > 1 else None      # the next batch's bytes are read while this one is on the device
                    nxt = ahead.apply_async(self._read_range, (mine[0], min(mine[0] + per_call, last + 1))) if ahead else None
                    for k, b0 in enumerate(mine):
                        b1 = min(b0 + per_call, last + 1)
                        dst = whole[self.chunk_bounds[b0] - r0:self.chunk_bounds[b1] - r0]
                        if direct:
                            buf = nxt.get() if nxt is not None else None
                            nxt = ahead.apply_async(self._read_range, (mine[k + 1], min(mine[k + 1] + per_call, last + 1))) \
                                if ahead and k + 1 < len(mine) else None
                            self._decode_into(b0, b1, dst, buf, lane=g if lanes > 1 else None)
                        else:
                            chunks = self.decompress_chunks(range(b0, b1))
                            for idx in range(b0, b1):
                                dst[self.chunk_bounds[idx] - self.chunk_bounds[b0]:self.chunk_bounds[idx + 1] - self.chunk_bounds[b0]] = chunks[idx]
                            del chunks
                    if ahead:
                        ahead.close()
                if lanes == 1:
                    lane_loop(0)
                else:
                    self.codec.run_lanes(lane_loop, lanes)
                out = whole[i0 - r0:i1 - r0:item.step, :]
                assert out.shape[0] == len(range(i0, i1, item.step or 1))
                return out
            triples = list(self.iter_chunks(first, last))
            with self._cache_lock:
                keep = self.cache_size
                if len(triples) > keep:
                    self.cache_size = len(triples)
                try:
                    decoded = self._decode(triples)
                finally:
                    self.cache_size = keep
                    self._trim_cache()
            chunks = [decoded[i] for i in range(first, last + 1)]
            arr = _join_rows(chunks)
            a = i0 - self.chunk_bounds[first]
            b = i1 - self.chunk_bounds[first]
            assert 0 <= a <= b <= arr.shape[0]
            out = arr[a:b:item.step, :]
            assert out.shape[0] == len(range(i0, i1, item.step or 1))
            return out
        elif isinstance(item, tuple):
            if len(item) == 1:
                return self[item[0]]
            elif len(item) == 2 and np.isscalar(item[0]):
                return self[item[0]][item[1]]
            elif len(item) == 2:
                got = self.read_slices([item], _fallback=False)
                if got is not None:
                    return got[0]
                return self[item[0]][:, item[1]]
        elif isinstance(item, (int, np.integer)):
            item = int(item)
            if item < 0:
                item += self.n_samples * (-(item // self.n_samples))
                assert 0 <= item < self.n_samples
            if not 0 <= item < self.n_samples:  # pragma: no cover
                raise IndexError("index %d is out of bounds for axis 0 with size %d" % (item, self.n_samples))
            return self[item:item + 1][0]
        elif isinstance(item, (list, np.ndarray)):  # pragma: no cover
            raise NotImplementedError("Indexing with multiple values is currently unsupported.")
        return fallback  # pragma: no cover

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass


# ------------------------------------------------------------------------------------------------
# high-level API (mtscomp.py:862-997)
# ------------------------------------------------------------------------------------------------
def check(data, out, outmeta, codec=None):
    """Decompress everything and compare with `data` (mtscomp.py:866-888)."""
    unc = decompress(out, outmeta, codec=codec)
    try:
        for first in range(0, unc.n_chunks, unc.batch_size):
            ids = range(first, min(first + unc.batch_size, unc.n_chunks))
            chunks = unc.decompress_chunks(ids)
            for idx in ids:
                chunk = chunks[idx]
                expected = data[unc.chunk_bounds[idx]:unc.chunk_bounds[idx + 1]]
                assert chunk.dtype == expected.dtype
                assert chunk.shape == expected.shape
                if np.issubdtype(chunk.dtype, np.integer):
                    assert np.array_equal(chunk, expected)
                else:
                    assert np.allclose(chunk, expected, atol=CHECK_ATOL)
    finally:
        unc.close()


def compress(path, out=None, outmeta=None, sample_rate=None, n_channels=None, dtype=None, codec=None, **kwargs):
    """Compress a flat binary (or .npy) file into `out` (.cbin) + `outmeta` (.ch); returns the
    compression ratio, as the reference does (mtscomp.py:891-958)."""
    w = Writer(codec=codec, **kwargs)
    w.open(path, sample_rate=sample_rate, n_channels=n_channels, dtype=dtype)
    ratio = w.write(out, outmeta)
    w.close()
    return ratio


def decompress(cdata, cmeta=None, out=None, write_output=False, overwrite=False, codec=None, **kwargs):
    """Open a compressed dataset; optionally write it out decompressed (mtscomp.py:961-997)."""
    if out:
        write_output = True
    r = Reader(codec=codec, **kwargs)
    r.open(cdata, cmeta)
    if write_output:
        r.tofile(out, overwrite=overwrite)
    return r
