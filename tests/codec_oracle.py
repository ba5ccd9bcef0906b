"""Test-only codec: the CPU oracle behind the codec interface of mtscomp_amd.api.

Lets the `-m "not gpu"` suite exercise the HOST logic (file format, header JSON, slicing, caching, error
mapping, batching/sharding) without a GPU.  The package never imports this; it only ever builds HipCodec."""
import numpy as np

from oracle import oracle as O


class OracleCodec:
    name = 'oracle'

    def __init__(self, n_devices=1, use_c=True):
        self.devices = list(range(n_devices))
        self.use_c = use_c
        self.calls = []          # (kind, n_chunks) per codec call

    @staticmethod
    def _unflags(flags):
        return bool(flags & 1), bool(flags & 2), 'F' if flags & 4 else 'C'

    def compress(self, chunks, flags, level=6):
        self.calls.append(('compress', len(chunks)))
        out = []
        for c in chunks:
            c = np.ascontiguousarray(c)
            if self.use_c and c.dtype.kind in 'iuf':
                out.append(O.compress_chunk(c, flags, level))
            else:
                out.append(O.ref_compress_chunk(c, *self._unflags(flags)))
        return out

    def decompress(self, cbufs, n_rows, n_channels, dtype, flags):
        self.calls.append(('decompress', len(cbufs)))
        status, arrays = [], []
        dtype = np.dtype(dtype)
        for cb, nr in zip(cbufs, n_rows):
            if self.use_c and dtype.kind in 'iuf':
                rc, arr = O.decompress_chunk(cb, nr, n_channels, dtype, flags)
                rc = 0 if rc == 0 else (-2 if rc == 1 else -1)
            else:
                try:
                    arr = O.ref_decompress_chunk(cb, nr, n_channels, dtype, *self._unflags(flags))
                    rc = 0
                except AssertionError:
                    rc, arr = -2, None
                except Exception:
                    rc, arr = -1, None
            status.append(rc)
            arrays.append(arr if rc == 0 else None)
        return status, arrays

    def delta(self, arr, flags):
        return O.delta_transpose(arr, flags)

    def cumsum(self, stream, nt, nc, dtype, flags):
        return O.cumsum_transpose(stream, nt, nc, dtype, flags)


class CachingOracleCodec(OracleCodec):
    """OracleCodec + the decoded-chunk cache interface of HipCodec (mts_cache_* in the C ABI), kept in a dict: lets the CPU
    suite drive Reader._slice_from_device_cache (reads of the missing byte ranges, the retry after a miss, error mapping)."""
    device_cache = True

    def __init__(self, capacity_chunks=3, **kw):
        super().__init__(**kw)
        self.capacity_chunks = capacity_chunks
        self.caches = {}
        self.drop_before_next_read = False          # simulate an eviction between query and read

    def cache_create(self, capacity_bytes):
        cid = len(self.caches) + 1
        self.caches[cid] = {}
        return cid

    def cache_destroy(self, cid):
        self.caches.pop(cid, None)

    def cache_query(self, cid, keys):
        return np.array([self.caches[cid][k].shape[1] if k in self.caches[cid] else 0 for k in keys], dtype=np.int32)

    def _ensure(self, cid, keys, cdata, offs, lens, n_rows, n_channels, dtype, flags, what):
        from mtscomp_amd import hip
        cache = self.caches[cid]
        if self.drop_before_next_read:
            self.drop_before_next_read = False
            cache.clear()
        self.calls.append((what, sum(1 for n in lens if n)))
        status = []
        for k, o, n, nr in zip(keys, offs, lens, n_rows):
            if k in cache:
                status.append(0)
                continue
            if not n:
                raise hip.HipError(hip.E_MISS, 'mts_cache_read_rows', 'chunk key %d is not resident' % k)
            st, arrs = self.decompress([bytes(cdata[o:o + n])], [nr], n_channels, dtype, flags)
            self.calls.pop()                         # (the inner decompress call is not a codec call of its own)
            status.append(st[0])
            if st[0] == 0:
                cache[k] = arrs[0]
        return cache, status

    def cache_read_slices(self, cid, keys, cdata, offs, lens, n_rows, n_channels, dtype, flags, requests):
        """mts_cache_read_slices restated: rectangles of the concatenation of the listed chunks."""
        cache, status = self._ensure(cid, keys, cdata, offs, lens, n_rows, n_channels, dtype, flags, 'cache_slices')
        whole = np.concatenate([cache[k] if st == 0 else np.zeros((nr, n_channels), dtype=dtype)
                                for k, nr, st in zip(keys, n_rows, status)], axis=0)
        arrays = [np.ascontiguousarray(whole[rb:re:rs, cb:ce:cs]) for rb, re, rs, cb, ce, cs in requests]
        while len(cache) > self.capacity_chunks:
            cache.pop(next(iter(cache)))
        return status, arrays

    def cache_read_rows(self, cid, keys, cdata, offs, lens, n_rows, n_channels, dtype, flags, row_begin, row_end):
        from mtscomp_amd import hip
        cache = self.caches[cid]
        if self.drop_before_next_read:
            self.drop_before_next_read = False
            cache.clear()
        self.calls.append(('cache_read', sum(1 for n in lens if n)))
        status = []
        for k, o, n, nr in zip(keys, offs, lens, n_rows):
            if k in cache:
                status.append(0)
                continue
            if not n:
                raise hip.HipError(hip.E_MISS, 'mts_cache_read_rows', 'chunk key %d is not resident' % k)
            st, arrs = self.decompress([bytes(cdata[o:o + n])], [nr], n_channels, dtype, flags)
            self.calls.pop()                         # (the inner decompress call is not a codec call of its own)
            status.append(st[0])
            if st[0] == 0:
                cache[k] = arrs[0]
        out = np.empty((row_end - row_begin, n_channels), dtype=dtype)
        r0 = 0
        for k, nr, st in zip(keys, n_rows, status):
            lo, hi = max(row_begin, r0), min(row_end, r0 + nr)
            if lo < hi and st == 0:
                out[lo - row_begin:hi - row_begin] = cache[k][lo - r0:hi - r0]
            r0 += nr
        while len(cache) > self.capacity_chunks:     # least recently inserted goes first
            cache.pop(next(iter(cache)))
        return status, out


class LeadingOracleCodec(CachingOracleCodec):
    """CachingOracleCodec + mts_cache_read_slices_leading restated (stdlib zlib's streaming inflate on a prefix of the bytes): lets
    the CPU suite drive the Reader's prefix reads, its retry with whole chunks and the entries of leading channels."""
    leading_channels = True

    def __init__(self, **kw):
        super().__init__(**kw)
        self.cols = {}                 # (cache id, key) -> channels the entry holds
        self.bytes_given = []          # per call: compressed bytes handed over

    def cache_read_slices(self, cid, keys, cdata, offs, lens, n_rows, n_channels, dtype, flags, requests, n_leading=None):
        import zlib
        from mtscomp_amd import hip
        dtype = np.dtype(dtype)
        n_lead = n_leading or n_channels
        assert all(ce <= n_lead for _, _, _, _, ce, _ in requests)
        cache = self.caches[cid]
        self.calls.append(('cache_slices', sum(1 for n in lens if n)))
        self.bytes_given.append(int(sum(lens)))
        status = []
        for k, o, n, nr in zip(keys, offs, lens, n_rows):
            if k in cache and self.cols.get((cid, k), 1 << 20) >= n_lead:
                status.append(0)
                continue
            if not n:
                raise hip.HipError(hip.E_MISS, 'mts_cache_read_slices', 'chunk key %d is not resident with the channels asked for' % k)
            need = nr * n_lead * dtype.itemsize
            if n_lead < n_channels:
                assert flags & 4 and dtype.kind in 'iu'
                d = zlib.decompressobj()
                stream = d.decompress(bytes(cdata[o:o + n]), need)           # (a prefix of the stream from a prefix of the bytes)
                if len(stream) < need:
                    raise hip.HipError(hip.E_MISS, 'mts_cache_read_slices', 'the bytes given do not reach the leading channels')
                arr = O.cumsum_transpose(np.frombuffer(stream[:need], dtype=np.uint8), nr, n_lead, dtype, flags)
                st = 0
            else:
                sts, arrs = self.decompress([bytes(cdata[o:o + n])], [nr], n_channels, dtype, flags)
                self.calls.pop()
                st, arr = sts[0], arrs[0]
            status.append(st)
            if st == 0:
                cache[k] = arr
                self.cols[(cid, k)] = n_lead
        whole = np.concatenate([cache[k][:, :n_lead] if st == 0 else np.zeros((nr, n_lead), dtype=dtype)
                                for k, nr, st in zip(keys, n_rows, status)], axis=0)
        arrays = [np.ascontiguousarray(whole[rb:re:rs, cb:ce:cs]) for rb, re, rs, cb, ce, cs in requests]
        return status, arrays

    def cache_query(self, cid, keys):
        return np.array([self.cols.get((cid, k), 1 << 20) if k in self.caches[cid] else 0 for k in keys], dtype=np.int32)      # (entries made by cache_read_rows are whole chunks)


class LaneOracleCodec(LeadingOracleCodec):
    """CachingOracleCodec with several lanes, the way HipCodec has one per device: a cache per lane (cache_create(capacity, lane)),
    run_lanes() on host threads, decompress(..., out=, lane=) for Reader.tofile's pieces.  Records which lane served what, so the
    CPU suite can check that chunk k is read, decoded and kept on lane k mod n_lanes and nowhere else."""
    takes_out = True
    takes_ranges = True
    leading_channels = False                      # (LaneOracleCodec(leading=True): the lanes decode leading channels only, like HipCodec)

    def __init__(self, n_lanes=2, leading=False, **kw):
        super().__init__(n_devices=n_lanes, **kw)
        self.n_lanes = n_lanes
        self.leading_channels = bool(leading)
        self.cache_lane = {}                      # cache id -> lane
        self.lane_keys = {}                       # lane -> set of chunk keys its cache was asked for
        self.lane_calls = []                      # (lane, n_chunks) of decompress(..., lane=)

    def cache_create(self, capacity_bytes, lane=0):
        cid = super().cache_create(capacity_bytes)
        self.cache_lane[cid] = lane
        return cid

    def run_lanes(self, fn, n):
        import threading
        errors = []

        def guarded(k):
            try:
                fn(k)
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
        threads = [threading.Thread(target=guarded, args=(k,)) for k in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    def _note(self, cid, keys):
        self.lane_keys.setdefault(self.cache_lane[cid], set()).update(int(k) for k in keys)

    def cache_read_rows(self, cid, keys, cdata, offs, lens, n_rows, n_channels, dtype, flags, row_begin, row_end, out=None):
        self._note(cid, keys)
        status, got = super().cache_read_rows(cid, keys, cdata, offs, lens, n_rows, n_channels, dtype, flags, row_begin, row_end)
        if out is not None:
            out[...] = got
            got = out
        return status, got

    def cache_read_slices(self, cid, keys, cdata, offs, lens, n_rows, n_channels, dtype, flags, requests, n_leading=None):
        self._note(cid, keys)
        return super().cache_read_slices(cid, keys, cdata, offs, lens, n_rows, n_channels, dtype, flags, requests, n_leading=n_leading)

    def decompress(self, cbufs, n_rows, n_channels, dtype, flags, out=None, lane=None):
        if isinstance(cbufs, tuple):
            buf, offs, lens = cbufs
            cbufs = [bytes(memoryview(buf)[o:o + n]) for o, n in zip(offs, lens)]
        if lane is not None:
            self.lane_calls.append((lane, len(cbufs)))
        status, arrays = super().decompress(cbufs, n_rows, n_channels, dtype, flags)
        if out is not None:
            r0 = 0
            for nr, a in zip(n_rows, arrays):
                if a is not None:
                    out[r0:r0 + nr] = a
                r0 += nr
        return status, arrays
