/*
 * mtscomp_hip.h -- C ABI of libmtscomp_hip.so: mtscomp's per-chunk codec on MI355X (gfx950).
 *
 * The reference (int-brain-lab/mtscomp, pure Python) has no FFI for this path: the hot path sits
 * behind Python methods that call numpy and the stdlib zlib module.  The entry points below are
 * what a ctypes binding of that path binds (INTEGRATION.md shows the stub).  Each one cites the
 * reference interface it replaces (file:line in /root/reference/mtscomp.py).
 *
 * Conventions: plain C types; the caller owns every buffer; return value 0 (MTS_OK) or a negative
 * MTS_E_* code (no exceptions cross the boundary); every call is re-entrant and thread-safe (calls on
 * one device are serialised inside the library); ctypes releases the GIL for the duration.
 * There is NO CPU fallback: without a usable gfx950 device every compute entry point returns
 * MTS_E_NODEV.
 *
 * `flags`: bit0 do_time_diff, bit1 do_spatial_diff, bit2 chunk_order=='F'   (reference config keys,
 * mtscomp.py:52-55), bit3 items are IEEE floats.  `level`: zlib level 1..9 or -1 (= 6), every one
 * byte-identical to zlib.compress(stream, level) of libz 1.2.11; the reference always compresses at
 * zlib's default (6) whatever `comp_level` says (mtscomp.py:394), so 6/-1 is the drop-in value.
 * Supported item types: integer dtypes of 1, 2, 4 or 8 bytes (two's complement wrap, like numpy) and,
 * with MTS_FLAG_FLOAT, float32 / float64 (np.diff / np.cumsum in the item type, bit for bit).
 */
#ifndef MTSCOMP_HIP_H
#define MTSCOMP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MTS_OK 0
#define MTS_E_ARG (-1)         /* bad argument */
#define MTS_E_NODEV (-2)       /* no usable gfx950 device / HIP runtime */
#define MTS_E_HIP (-3)         /* HIP runtime error (see mts_last_error) */
#define MTS_E_NOMEM (-4)       /* device or host allocation failed */
#define MTS_E_UNSUPPORTED (-5) /* valid request this build does not implement (e.g. the match-table tap at levels 1..3) */
#define MTS_E_INTERNAL (-6)    /* internal consistency check failed */
#define MTS_E_MISS (-7)        /* mts_cache_read_rows: a chunk given without bytes is not resident (any more) */

/* per-chunk status written by mts_decompress_chunks */
#define MTS_CHUNK_OK 0
#define MTS_CHUNK_CORRUPT (-1)    /* zlib.decompress would raise  -> IOError, mtscomp.py:618-621 */
#define MTS_CHUNK_BADSIZE (-2)    /* valid stream (check value included) of the wrong length -> AssertionError, mtscomp.py:628;
                                     a damaged stream of the wrong length is MTS_CHUNK_CORRUPT, as zlib reports it first */

#define MTS_FLAG_TIME_DIFF 1
#define MTS_FLAG_SPATIAL_DIFF 2
#define MTS_FLAG_ORDER_F 4
#define MTS_FLAG_FLOAT 8          /* items are IEEE floats (itemsize 4 or 8): np.diff / np.cumsum in that type, bit for bit */

int mts_version(void);
int mts_device_count(void);                 /* number of gfx950 devices visible; 0 if none */
const char *mts_strerror(int code);
const char *mts_last_error(void);           /* thread-local detail of the last failure */

/* zlib's compressBound(): capacity to reserve per chunk slot */
long mts_compress_bound(long raw_len);

/*
 * diff_along_axis(axis=0/1) + ndarray.tobytes(order)            mtscomp.py:143-159, :381-382, :394
 * raw: C-order (n_samples, n_channels).  stream_out: n_samples*n_channels*itemsize bytes.
 */
int mts_delta_transpose(int device, const void *raw, long n_samples, int n_channels, int itemsize,
                        int flags, void *stream_out);

/*
 * reshape(order) + cumsum_along_axis(axis=1/0) + ascontiguousarray   mtscomp.py:622-635, :162-169
 * stream: the inflated bytes of one chunk.  out: C-order (n_samples, n_channels).
 */
int mts_cumsum_transpose(int device, const void *stream, long n_samples, int n_channels,
                         int itemsize, int flags, void *out);

/*
 * One batch of Writer._compress_chunk calls -- replaces `pool.map(self._compress_chunk, range(...))`
 * (mtscomp.py:375-397, :399-423).
 *   raw            C-order (rows, n_channels) array holding rows [chunk_bounds[0], chunk_bounds[n_chunks])
 *                  -- i.e. `raw` points at row chunk_bounds[0]
 *   chunk_bounds   n_chunks+1 row indices; chunk i = rows [b[i], b[i+1])        (mtscomp.py:324-339)
 *   out            caller buffer; chunk i's zlib stream is written at out + out_slot_offsets[i], which
 *                  must have room for mts_compress_bound(len_i) bytes
 *   out_sizes      n_chunks compressed lengths (what `len(chunkdc)` is in mtscomp.py:478)
 * Output bytes are identical to zlib.compress() of libz 1.2.11 at `level`.
 */
int mts_compress_chunks(int device, const void *raw, int n_channels, int itemsize,
                        const long *chunk_bounds, int n_chunks, int flags, int level,
                        unsigned char *out, const long *out_slot_offsets, long *out_sizes);

/*
 * One batch of Reader.read_chunk calls -- replaces `pool.map(self._decompress_chunk, ids)`
 * (mtscomp.py:602-643, :645-650).  Chunks need not be adjacent in the file.
 *   cdata          base of the compressed bytes the caller read (os.pread, mtscomp.py:609)
 *   c_offsets      n_chunks byte offsets into cdata;  c_lengths: n_chunks byte lengths
 *   n_rows         n_chunks row counts (chunk_bounds[i+1] - chunk_bounds[i])
 *   out            caller buffer;  chunk i's C-order (n_rows[i], n_channels) array goes to
 *                  out + out_offsets[i] (bytes)
 *   chunk_status   MTS_CHUNK_* per chunk; a corrupt chunk does not stop the others (mtscomp.py:621)
 * Accepts any valid RFC 1950/1951 stream (stored/fixed/dynamic blocks, any encoder); verifies adler32;
 * ignores trailing bytes after the stream like zlib.decompress does.
 */
int mts_decompress_chunks(int device, const unsigned char *cdata, const long *c_offsets,
                          const long *c_lengths, const long *n_rows, int n_chunks, int n_channels,
                          int itemsize, int flags, void *out, const long *out_offsets,
                          int *chunk_status);

/*
 * Reader random access (Reader.__getitem__, mtscomp.py:798-856, with read_chunk's lru_cache of decoded chunks,
 * mtscomp.py:582-588, :602) -- the cache lives in HBM: decoded chunks stay on the device and a slice costs one
 * device-to-host copy of exactly the requested rows.
 *   mts_cache_create    capacity in bytes of decoded chunks (least recently used chunks are dropped beyond it)
 *   mts_cache_query     present[i] = 0 if the decoded chunk with key chunk_keys[i] is not resident, else the number of channels
 *                       the entry holds (n_channels, or the leading channels of mts_cache_read_slices_leading)
 *   mts_cache_read_slices  (below) any number of row/column rectangles per call, gathered on the device
 *   mts_cache_read_rows the chunks of one slice, in file order: resident ones may come with c_lengths[i] = 0, the others
 *                       with their compressed bytes (cdata + c_offsets[i], c_lengths[i]) and are decoded in one batch and
 *                       kept.  Rows [row_begin, row_end) of the concatenation of the n_chunks chunks are written to `out`
 *                       (C order).  chunk_status as in mts_decompress_chunks (rows of a failed chunk are not written).
 *                       MTS_E_MISS: a chunk given without bytes is not resident -- call again with its bytes.
 * Keys are the caller's (the Reader uses the chunk index; one cache per open file).
 */
int mts_cache_create(int device, long capacity_bytes, long *cache_id);
int mts_cache_destroy(long cache_id);
int mts_cache_query(long cache_id, const long *chunk_keys, int n, int *present);
int mts_cache_read_rows(long cache_id, int n_chunks, const long *chunk_keys, const unsigned char *cdata,
                        const long *c_offsets, const long *c_lengths, const long *n_rows, int n_channels,
                        int itemsize, int flags, long row_begin, long row_end, void *out,
                        int *chunk_status);
/* Several rectangular pieces in one call -- Reader[rows, columns] (mtscomp.py:835-842, where the reference decodes whole
 * chunks and drops rows and columns on the host) and many slices per launch.  The chunks are the union of what the requests
 * touch (file order, every key once; residency and bytes as in mts_cache_read_rows).  Request k is six longs,
 *   row_begin, row_end, row_step (>= 1), col_begin, col_end, col_step (>= 1),
 * rows counted in the concatenation of the listed chunks; its ceil((row_end - row_begin) / row_step) x
 * ceil((col_end - col_begin) / col_step) items are gathered ON THE DEVICE and written C-contiguous at out + out_offsets[k]
 * (bytes); the out_bytes of `out` cross the bus in one copy and nothing else does.  Rows of a failed chunk are not written. */
int mts_cache_read_slices(long cache_id, int n_chunks, const long *chunk_keys, const unsigned char *cdata,
                          const long *c_offsets, const long *c_lengths, const long *n_rows, int n_channels,
                          int itemsize, int flags, int n_requests, const long *requests, void *out,
                          const long *out_offsets, long out_bytes, int *chunk_status);
/* The same for requests that only touch the first n_leading channels of channel-major integer chunks (chunk_order 'F', the
 * reference's default: the stream of a chunk is channel after channel, so the leading channels are a PREFIX of it): a chunk that
 * is not resident is inflated only until that prefix is complete -- whole deflate blocks, no adler32 check, which needs the
 * whole stream -- and kept as a (rows, n_leading) entry; c_lengths[i] may then be a prefix of the chunk's compressed bytes
 * (about n_leading / n_channels of them plus a margin).  MTS_E_MISS: the bytes given do not reach the prefix, or a resident entry
 * holds fewer channels than asked for -- call again with more bytes.  Column ranges of the requests must end at or before
 * n_leading.  n_leading == n_channels is mts_cache_read_slices.  (The reference decodes whole chunks and drops columns on the
 * host, mtscomp.py:835-842.) */
int mts_cache_read_slices_leading(long cache_id, int n_chunks, const long *chunk_keys, const unsigned char *cdata,
                                  const long *c_offsets, const long *c_lengths, const long *n_rows, int n_channels,
                                  int itemsize, int flags, int n_leading, int n_requests, const long *requests,
                                  void *out, const long *out_offsets, long out_bytes, int *chunk_status);

/* ---------------------------------------------------------------------------------------------
 * Device-resident variants (inputs and outputs already in HBM; used by bench.py and by callers that
 * keep recordings on the GPU).  Pointers are device pointers on `device`; `stream` is a hipStream_t
 * (0 = default stream).  The small index arrays stay on the host.  Not part of the drop-in.
 * ------------------------------------------------------------------------------------------- */
int mts_dev_compress_chunks(int device, void *stream, const void *d_raw, int n_channels, int itemsize,
                            const long *chunk_bounds, int n_chunks, int flags, int level,
                            unsigned char *d_out, const long *out_slot_offsets /* 16-B aligned */,
                            long *out_sizes /* host; valid when the call returns */);
int mts_dev_decompress_chunks(int device, void *stream, const unsigned char *d_cdata,
                              const long *c_offsets, const long *c_lengths, const long *n_rows,
                              int n_chunks, int n_channels, int itemsize, int flags, void *d_out,
                              const long *out_offsets, int *chunk_status /* host */);
/* integer-exact synthetic recording, rows [t0, t1) of n_channels int16, on device */
int mts_dev_synth_int16(int device, void *stream, void *d_out, long t0, long t1, int n_channels,
                        long seed);

/* Device memory for callers of the dev_* entry points (bench.py, the tests at BASELINE's sizes): allocate, copy and wait through
 * THIS library, so that a process holds its recordings with the HIP runtime the kernels are launched with and needs no other
 * (torch ships its own runtime libraries).  kind: 0 host -> device, 1 device -> host, 2 device -> device; mts_dev_copy returns
 * when the copy is done.  mts_dev_compare: bytes that differ between two device buffers (16-byte aligned) and the first such offset
 * (-1: none) -- the round-trip check of a recording that stays in HBM. */
/* Page-locked host memory (hipHostMalloc): the host entry points copy to / from such a buffer by DMA directly, without the pinned
 * pieces and host copies that pageable memory needs (Reader.tofile decodes into two of these and writes the file from them). */
int mts_host_alloc(long nbytes, void **h_ptr);
int mts_host_free(void *h_ptr);
int mts_dev_alloc(int device, long nbytes, void **d_ptr);
int mts_dev_free(int device, void *d_ptr);
int mts_dev_copy(int device, void *stream, void *dst, const void *src, long nbytes, int kind);
int mts_dev_sync(int device);               /* hipDeviceSynchronize on `device` */
int mts_dev_compare(int device, void *stream, const void *d_a, const void *d_b, long nbytes, long *n_diff, long *first_diff);

/* Kernel-stage timings (ms, HIP events on the launch stream) of the last dev_* call on `device`:
 * fills up to `cap` entries of (name, ms); returns the number of stages.  For bench.py / profiling. */
int mts_last_stage_times(int device, const char **names, float *ms, int cap);

/* Debug/parity taps for the GPU tests (stage-by-stage comparison with the oracle); host buffers. */
int mts_debug_match_tables(int device, const void *stream_bytes, long n, int level,
                           unsigned *t_full, unsigned *t_quarter);
int mts_debug_tokens(int device, const void *stream_bytes, long n, int level,
                     unsigned short *tokens /* (dist, lc) pairs, capacity n+1 */, long *n_tokens);
int mts_debug_deflate(int device, const void *stream_bytes, long n, int level, unsigned char *out,
                      long out_cap, long *out_len);
int mts_debug_inflate(int device, const unsigned char *zbytes, long zlen, unsigned char *out,
                      long out_cap, long *out_len, int *status);

/* release every device allocation held by the library (workspaces are otherwise cached) */
void mts_release(void);

#ifdef __cplusplus
}
#endif
#endif
