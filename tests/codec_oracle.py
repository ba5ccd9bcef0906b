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
