"""GPU parity, stage by stage, through the C ABI (libmtscomp_hip.so) against the CPU oracle.

Bit-exact everywhere: these are integer / byte / index computations."""
import time
from pathlib import Path
import zlib

import numpy as np
import pytest

from mtscomp_amd import hip
from mtscomp_amd.synth import synth_int16
from oracle import oracle as O
from tests import inputs
from tests.inputs import cases_small, ar1_stream, repeats

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu
CASES = cases_small()


def synth_rows(nt, nc, seed):
    return synth_int16(seed * 100000, seed * 100000 + nt, nc, 0)


def _first_diff(a, b):
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    n = min(a.size, b.size)
    d = np.nonzero(a[:n] != b[:n])[0]
    return (int(d[0]) if d.size else n), a.size, b.size


def test_device_visible():
    assert hip.require_device() >= 1


@pytest.mark.parametrize('dtype', ['int16', 'uint16', 'uint8', 'int8', 'int32', 'int64'])
@pytest.mark.parametrize('flags', range(8))
def test_k1_k2_transforms(dtype, flags):
    r = np.random.RandomState(flags + 10)
    info = np.iinfo(dtype)
    for shape in [(37, 11), (1, 385), (700, 70), (513, 1), (300, 129)]:
        x = r.randint(info.min, int(info.max) + 1, size=shape, dtype=np.int64).astype(dtype)
        want = O.delta_transpose(x, flags)
        got = hip.delta_transpose(x, flags)
        assert np.array_equal(got, want), (shape, _first_diff(got, want))
        back = hip.cumsum_transpose(want, shape[0], shape[1], dtype, flags)
        assert back.dtype == x.dtype and np.array_equal(back, x), (shape, _first_diff(back, x))


def test_k1_k2_neuropixels_shape():
    from mtscomp_amd.synth import synth_int16
    x = synth_int16(0, 3000, 385, 0)
    want = O.delta_transpose(x, 5)
    got = hip.delta_transpose(x, 5)
    assert np.array_equal(got, want)
    assert np.array_equal(hip.cumsum_transpose(want, 3000, 385, 'int16', 5), x)


TABLE_CASES = ['three', 'aaa', 'zeros_1k', 'rand_300', 'first50', 'ar1_8ch', 'text_100k', 'zeros_70k',
               'rand4_50k', 'repeats_200k', 'ar1_64ch_4k', 'rand_70k', 'ramp', 'wrap16']


@pytest.mark.parametrize('name', TABLE_CASES)
def test_match_tables(name):
    data = CASES[name]
    tf, tq = O.match_tables(data, 6)
    gf, gq = hip.debug_match_tables(data, 6)
    i, _, _ = _first_diff(gf, tf)
    assert np.array_equal(gf, tf), ('t_full', i, hex(int(gf[i])), hex(int(tf[i])))
    i, _, _ = _first_diff(gq, tq)
    assert np.array_equal(gq, tq), ('t_quarter', i, hex(int(gq[i])), hex(int(tq[i])))


@pytest.mark.parametrize('level', [4, 5, 7, 8, 9])
def test_match_tables_other_levels(level):
    data = CASES['repeats_200k'][:120000]
    tf, tq = O.match_tables(data, level)
    gf, gq = hip.debug_match_tables(data, level)
    assert np.array_equal(gf, tf) and np.array_equal(gq, tq)


@pytest.mark.parametrize('name', TABLE_CASES + ['empty', 'one', 'two'])
def test_tokens(name):
    data = CASES[name]
    _, toks, _, _ = O.deflate(data, 6, report=True)
    got = hip.debug_tokens(data, 6)
    assert got.shape == toks.shape, (got.shape, toks.shape, _first_diff(got, toks))
    assert np.array_equal(got, toks), _first_diff(got, toks)


@pytest.mark.parametrize('name', sorted(CASES))
def test_deflate_bytes(name):
    data = CASES[name]
    want = zlib.compress(data, 6)
    got = hip.debug_deflate(data, 6)
    assert len(got) == len(want), (len(got), len(want), _first_diff(np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)))
    assert got == want, _first_diff(np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8))


@pytest.mark.parametrize('level', [4, 5, 7, 8, 9])
def test_deflate_bytes_levels(level):
    for name in ('text_100k', 'ar1_8ch', 'repeats_200k'):
        data = CASES[name]
        assert hip.debug_deflate(data, level) == zlib.compress(data, level), name


def test_deflate_bytes_ballot_sort(monkeypatch):
    """The hash sort ranks with LDS atomics when the start-up probe finds them lane ordered; MTS_SORT_BALLOT=1
    forces the ballot ranking that is used otherwise.  Both must give zlib's bytes."""
    monkeypatch.setenv('MTS_SORT_BALLOT', '1')
    for name in ('text_100k', 'ar1_64ch_4k', 'repeats_200k', 'rand4_50k'):
        data = CASES[name]
        assert hip.debug_deflate(data, 6) == zlib.compress(data, 6), name


def test_deflate_bytes_one_pass_and_two_pass_sort_tiles(monkeypatch):
    """The hash sort takes a tile in one pass when it has few live hash buckets (recordings: small deltas) and leaves it to the
    two-pass kernel otherwise (uniform random bytes: all 32768 buckets) or when it is small -- decided per tile, on the device.
    A stream whose tiles (224 Ki positions) go either way, with runs and far copies in between; zlib's bytes at levels 6 and 9,
    with the default, with the two-pass kernel alone (MTS_SORT_TWO_PASS: read once per process, so that run is a process of its
    own) and with the ballot ranking."""
    import subprocess, sys, hashlib
    r = np.random.RandomState(123)
    few = inputs.ar1_stream(2400, 128, 7)                                 # ~600 KB of int16 deltas: few buckets
    rnd = r.randint(0, 256, size=700000).astype(np.uint8).tobytes()       # every bucket
    mid = inputs.repeats(300000, 31) + bytes(200000) + inputs.farcopies(250000, 32)
    data = few + rnd + mid + few[:300000] + rnd[:100000] + inputs.textlike(400000, 33)
    assert len(data) > 3 * 229376
    for level in (6, 9):
        want = zlib.compress(data, level)
        assert hip.debug_deflate(data, level) == want, level
    # tiles around the one-pass limit (1024 live buckets): symbols below 32 hash without collisions, so an alphabet of 10 gives
    # 10^3 = 1000 buckets (one pass, nearly all of its counters in use), alphabets of 11 and 12 give 1331 and 1728 (two passes)
    edge = b''.join(r.randint(0, a, size=n).astype(np.uint8).tobytes() for a, n in ((10, 500000), (11, 300000), (10, 250000), (12, 120000)))
    assert hip.debug_deflate(edge, 6) == zlib.compress(edge, 6)
    want6 = zlib.compress(data, 6)
    monkeypatch.setenv('MTS_SORT_BALLOT', '1')
    assert hip.debug_deflate(data, 6) == want6
    monkeypatch.delenv('MTS_SORT_BALLOT')
    code = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r); from mtscomp_amd import hip; "
            "d = open(sys.argv[1], 'rb').read(); print(hashlib.sha1(hip.debug_deflate(d, 6)).hexdigest())" % str(ROOT))
    import tempfile, os
    with tempfile.NamedTemporaryFile(delete=False) as f:
        f.write(data)
    try:
        env = dict(os.environ, MTS_SORT_TWO_PASS='1')
        out = subprocess.run([sys.executable, '-c', code, f.name], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-1000:]
        assert out.stdout.strip().split()[-1] == hashlib.sha1(want6).hexdigest()
    finally:
        os.unlink(f.name)


def test_deflate_bytes_from_the_marks_of_the_walks():
    """The tokens are written from the marks the parse walks leave (k_parse_emit_marks).  They must give zlib's bytes where a
    re-walk meets the first one (every segment), where it does not (zeros: the parse never re-synchronises), at levels whose lazy
    evaluation moves on more than twelve times in a step, and where a step looks past the two table windows a walker has in LDS
    (long lazy runs at the end of a segment's last window)."""
    big = inputs.repeats(300000, 21) + bytes(200000) + inputs.textlike(150000, 22) + inputs.skewlen(150000, 23)
    # lazy runs across segment ends: at every position a match one longer than the one before (a staircase), laid over the
    # last bytes of several 1024-position segments
    r = np.random.RandomState(77)
    stairs = bytearray(r.randint(0, 256, size=40000).astype(np.uint8).tobytes())
    unit = bytes(r.randint(0, 256, size=64).astype(np.uint8).tobytes())
    for segend in (4096, 9216, 20480, 30720):
        for j in range(12):                                           # position segend - 14 + j starts a copy of unit[j:] of length 3 + j ... + tail
            at = segend - 14 + j
            stairs[at:at + 1] = unit[j:j + 1]
        stairs[segend - 14:segend + 50] = unit
        stairs[segend - 300:segend - 300 + 64] = unit
    for level in (6, 9):
        for data in (CASES['text_100k'], CASES['ar1_64ch_4k'], CASES['zeros_999468'], big, bytes(stairs)):
            assert hip.debug_deflate(data, level) == zlib.compress(data, level), (level, len(data))


def test_deflate_block_boundaries():
    r = np.random.RandomState(7)
    base = r.randint(0, 256, size=16383 * 2 + 40).astype(np.uint8).tobytes()
    for n in (16382, 16383, 16384, 32766, 32767):
        assert hip.debug_deflate(base[:n], 6) == zlib.compress(base[:n], 6), n
    for extra in (3, 5, 9):
        d = base[:16382] + base[100:100 + extra]
        assert hip.debug_deflate(d, 6) == zlib.compress(d, 6), extra


def test_deflate_multi_tile():
    # > TILE (229376) positions: exercises the halo between match-stage tiles and many parse segments
    data = ar1_stream(20000, 16, seed=4)          # 640000 bytes: three tiles
    assert hip.debug_deflate(data, 6) == zlib.compress(data, 6)
    assert hip.debug_deflate(data, 9) == zlib.compress(data, 9)
    data = repeats(750000, 9)
    assert hip.debug_deflate(data, 6) == zlib.compress(data, 6)


def _fast_edge_cases():
    """Inputs picked for the in-order walk of levels 1..3 (deflate.hip, section F): windows of 64 positions, lists of a
    fixed length, a 32 KiB bitmap, the block buffer of 16383 tokens."""
    r = np.random.RandomState(11)
    out = {}
    for n in (0, 1, 2, 3, 4, 5, 63, 64, 65, 66, 67, 127, 128, 129, 191, 193, 257, 258, 259, 260, 261, 262):
        out['zeros_%d' % n] = bytes(n)                                    # matches of 258 across window ends, every member in the window
        out['rand_%d' % n] = r.randint(0, 4, size=n).astype(np.uint8).tobytes()
    out['zeros_100k'] = bytes(100000)
    out['ab_70k'] = b'ab' * 35000
    out['abc_70k'] = b'abc' * 23334
    per = r.randint(0, 256, size=32506).astype(np.uint8).tobytes()
    out['period_max_dist'] = per * 3 + per[:1000]                          # every candidate at exactly MAX_DIST
    per = r.randint(0, 256, size=32505).astype(np.uint8).tobytes()
    out['period_max_dist_m1'] = per * 3
    per = r.randint(0, 256, size=32507).astype(np.uint8).tobytes()
    out['period_max_dist_p1'] = per * 3
    out['period_32768'] = r.randint(0, 256, size=32768).astype(np.uint8).tobytes() * 3
    # many members of one hash run that the parse did not insert: long matches of one byte value, then short repeats of it
    blk = bytes([7]) * 300 + r.randint(0, 256, size=40).astype(np.uint8).tobytes()
    out['long_runs_then_probe'] = blk * 150 + bytes([7]) * 5 + b'x' + bytes([7]) * 4 + b'y' + bytes([7]) * 3
    # few distinct trigrams: long lists, chains that reach their budget
    out['rand2_200k'] = r.randint(0, 2, size=200000).astype(np.uint8).tobytes()
    out['rand3_64k'] = r.randint(0, 3, size=65536).astype(np.uint8).tobytes()
    # exactly a multiple of the block buffer in tokens (all literals): deflate_fast closes with an empty block
    out['literals_16383'] = r.permutation(np.arange(16383) % 251).astype(np.uint8).tobytes()
    return out


@pytest.mark.parametrize('level', [1, 2, 3])
def test_deflate_fast_levels(level):
    """Levels 1..3 are zlib's deflate_fast: greedy, and the inside of a long match is not entered into the hash chains, so
    the chains depend on the parse (deflate.hip, section F).  Byte identity on every small case, multi-tile streams and
    a dead channel between live ones."""
    for name, data in sorted(CASES.items()):
        got = hip.debug_deflate(data, level)
        want = zlib.compress(data, level)
        assert got == want, (name, len(got), len(want), _first_diff(np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)))
    for data in (ar1_stream(20000, 16, seed=4), repeats(750000, 9),
                 ar1_stream(3000, 8, seed=1) + bytes(90000) + ar1_stream(3000, 8, seed=2)):
        assert hip.debug_deflate(data, level) == zlib.compress(data, level), len(data)


@pytest.mark.parametrize('level', [1, 2, 3])
def test_fast_level_tokens(level):
    """The token sequence of the in-order walk against the oracle's deflate_fast (oracle.deflate(report=True)), token by token:
    a wrong decision shows here where it is made, not as a different byte somewhere in a Huffman block."""
    cases = dict(CASES)
    cases.update({k: v for k, v in _fast_edge_cases().items() if k in ('zeros_100k', 'long_runs_then_probe', 'rand3_64k', 'zeros_259', 'period_max_dist')})
    for name in TABLE_CASES + ['empty', 'one', 'two', 'zeros_100k', 'long_runs_then_probe', 'rand3_64k', 'zeros_259', 'period_max_dist']:
        data = cases[name]
        _, toks, _, _ = O.deflate(data, level, report=True)
        got = hip.debug_tokens(data, level)
        assert got.shape == toks.shape, (name, got.shape, toks.shape, _first_diff(got, toks))
        assert np.array_equal(got, toks), (name, _first_diff(got, toks))


@pytest.mark.parametrize('level', [1, 2, 3])
def test_deflate_fast_edge_cases(level):
    for name, data in sorted(_fast_edge_cases().items()):
        got = hip.debug_deflate(data, level)
        want = zlib.compress(data, level)
        assert got == want, (name, len(got), len(want), _first_diff(np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)))


def test_deflate_fast_many_phases(monkeypatch):
    # the candidate lists are made W positions at a time to fit a fixed workspace; the walk's state (position, token count,
    # the 32 KiB insertion bitmap, a long match in flight) is parked in memory in between: W = 256 here
    monkeypatch.setenv('MTS_FAST_LIST_BYTES', '1')
    cases = _fast_edge_cases()
    for level in (1, 2, 3):
        for data in (ar1_stream(6000, 16, seed=9), cases['zeros_100k'][:70000], cases['long_runs_then_probe'], cases['rand3_64k'],
                     cases['period_max_dist'][:70000]):
            assert hip.debug_deflate(data, level) == zlib.compress(data, level), (level, len(data))


def test_deflate_fast_sort_guard(monkeypatch):
    data = ar1_stream(3000, 16, seed=6)
    monkeypatch.setenv('MTS_SORT_INJECT_DISORDER', '1')
    assert hip.debug_deflate(data, 1) == zlib.compress(data, 1)


@pytest.mark.parametrize('name', sorted(CASES))
def test_inflate(name):
    data = CASES[name]
    for level in (0, 1, 6, 9):
        z = zlib.compress(data, level)
        st, out = hip.debug_inflate(z, len(data))
        assert st == 0 and out == data, (level, st, _first_diff(np.frombuffer(out, np.uint8), np.frombuffer(data, np.uint8)))
    st, out = hip.debug_inflate(zlib.compress(data) + b'trailing', len(data))
    assert st == 0 and out == data


@pytest.mark.parametrize('nseg', [2, 3, 4, 8, 16, 32])
def test_inflate_segmented_resolver(monkeypatch, nseg):
    """With few chunks in a batch the LZ resolver cuts a chunk into segments resolved by different workgroups on
    symbolic 16-bit cells (unknown 32 KiB window), then translates.  MTS_LZ_SEGS forces the cut."""
    monkeypatch.setenv('MTS_LZ_SEGS', str(nseg))
    big = inputs.repeats(700000, 11) + inputs.textlike(300000, 12) + inputs.farcopies(400000, 13)
    for data in (CASES['repeats_200k'], CASES['text_100k'], CASES['farcopies_900k'], big, inputs.ar1_stream(3000, 64)):
        for level in (1, 6):
            st, out = hip.debug_inflate(zlib.compress(data, level), len(data))
            assert st == 0 and out == data, (len(data), level, st, _first_diff(np.frombuffer(out, np.uint8), np.frombuffer(data, np.uint8)))


def test_inflate_other_encoders():
    data = CASES['text_100k']
    for strategy in (zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
        co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, strategy)
        z = co.compress(data) + co.flush()
        st, out = hip.debug_inflate(z, len(data))
        assert st == 0 and out == data, strategy
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8)
    z = b''.join(co.compress(data[i:i + 5000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(data), 5000)) + co.flush()
    st, out = hip.debug_inflate(z, len(data))
    assert st == 0 and out == data


def test_inflate_errors():
    data = CASES['ar1_8ch']
    z = zlib.compress(data)
    bad = [b'', z[:len(z) // 2], z[:-2], z[:-1] + bytes([z[-1] ^ 1]), b'\x79' + z[1:], b'\x78\x9c\x07']
    for b in bad:
        st, _ = hip.debug_inflate(b, len(data))
        assert st == hip.CHUNK_CORRUPT, (len(b), st)
    st, _ = hip.debug_inflate(z[:len(z) // 2] + bytes([z[len(z) // 2] ^ 0x55]) + z[len(z) // 2 + 1:], len(data))
    assert st != 0
    # a VALID stream of another length is a size mismatch (the assert of mtscomp.py:628) ...
    st, _ = hip.debug_inflate(z, len(data) - 1)
    assert st == hip.CHUNK_BADSIZE
    st, _ = hip.debug_inflate(z, len(data) + 1)
    assert st == hip.CHUNK_BADSIZE
    st, _ = hip.debug_inflate(zlib.compress(b''), len(data))
    assert st == hip.CHUNK_BADSIZE
    # ... a DAMAGED one that still parses to its end is zlib's data-check error (IOError, :618-621), whatever its length:
    # its check value is looked at before its size, as zlib.decompress does
    for other in (data + b'xyz', data[:-7], data[:5], data * 3):
        zo = zlib.compress(other)
        st, _ = hip.debug_inflate(zo, len(data))
        assert st == hip.CHUNK_BADSIZE
        for k in (1, 2, 3, 4):                                           # each byte of the check value
            st, _ = hip.debug_inflate(zo[:-k] + bytes([zo[-k] ^ 0x10]) + zo[len(zo) - k + 1:], len(data))
            assert st == hip.CHUNK_CORRUPT, (len(other), k)
    r = np.random.RandomState(5)
    for _ in range(60):
        b = bytearray(z)
        i = int(r.randint(2, len(b) - 4))
        b[i] ^= 1 << int(r.randint(0, 8))
        try:
            want = zlib.decompress(bytes(b))
        except zlib.error:
            want = None
        t0 = time.perf_counter()
        st, out = hip.debug_inflate(bytes(b), len(data))
        assert time.perf_counter() - t0 < 2.0          # (a copy from before the data once left the resolver's waves waiting for tens of seconds)
        if want is None:
            assert st == hip.CHUNK_CORRUPT
        elif len(want) != len(data):
            assert st == hip.CHUNK_BADSIZE
        else:
            assert st == 0 and out == want


def test_sub_batches(monkeypatch):
    """A call bigger than the workspace budget is split into sub-batches (MTS_BATCH_BYTES forces it: 1 MiB of stream
    per compress sub-batch, 4 MiB per decompress sub-batch, against 9 chunks of 0.3 .. 1.2 MB)."""
    monkeypatch.setenv('MTS_BATCH_BYTES', str(1 << 20))
    r = np.random.RandomState(9)
    chunks = [synth_rows(int(n), 64, k) for k, n in enumerate(r.randint(2400, 9600, size=9))]
    flags = 5
    bounds = np.concatenate(([0], np.cumsum([c.shape[0] for c in chunks])))
    cbufs = hip.compress_chunks(np.concatenate(chunks, axis=0), bounds, flags, 6)
    for c, z in zip(chunks, cbufs):
        assert bytes(z) == O.ref_compress_chunk(c)
    st, back = hip.decompress_chunks(cbufs, [c.shape[0] for c in chunks], 64, 'int16', flags)
    assert st == [0] * len(chunks)
    for c, b in zip(chunks, back):
        assert np.array_equal(c, b)


def test_inflate_errors_segmented_resolver(monkeypatch):
    """Bit flips in a stream big enough to be cut into resolver segments: same verdicts as zlib."""
    monkeypatch.setenv('MTS_LZ_SEGS', '4')
    data = inputs.repeats(500000, 21) + inputs.ar1_stream(2000, 64)
    z = zlib.compress(data)
    st, out = hip.debug_inflate(z, len(data))
    assert st == 0 and out == data
    r = np.random.RandomState(6)
    for _ in range(40):
        b = bytearray(z)
        i = int(r.randint(2, len(b) - 4))
        b[i] ^= 1 << int(r.randint(0, 8))
        try:
            want = zlib.decompress(bytes(b))
        except zlib.error:
            want = None
        st, out = hip.debug_inflate(bytes(b), len(data))
        if want is None:
            assert st == hip.CHUNK_CORRUPT
        elif len(want) != len(data):
            assert st == hip.CHUNK_BADSIZE
        else:
            assert st == 0 and out == want


@pytest.mark.parametrize('flags', [5, 7, 4, 1, 0, 3])
def test_compress_decompress_chunks(flags):
    from mtscomp_amd.synth import synth_int16
    x = synth_int16(0, 5300, 24, 3)
    bounds = [0, 1000, 2000, 3000, 4000, 5000, 5300]
    td, sd, order = bool(flags & 1), bool(flags & 2), 'F' if flags & 4 else 'C'
    want = [O.ref_compress_chunk(x[bounds[i]:bounds[i + 1]], td, sd, order) for i in range(6)]
    got = hip.compress_chunks(x, bounds, flags, 6)
    for i in range(6):
        assert got[i] == want[i], (i, len(got[i]), len(want[i]))
    rows = [bounds[i + 1] - bounds[i] for i in range(6)]
    st, arrs = hip.decompress_chunks(want, rows, 24, 'int16', flags)
    assert st == [0] * 6
    for i in range(6):
        assert np.array_equal(arrs[i], x[bounds[i]:bounds[i + 1]]), i
    # one corrupt chunk does not stop the others
    bad = list(want)
    bad[2] = bad[2][:50] + bytes([bad[2][50] ^ 0xff]) + bad[2][51:]
    st, arrs = hip.decompress_chunks(bad, rows, 24, 'int16', flags)
    assert st[2] != 0 and [s for i, s in enumerate(st) if i != 2] == [0] * 5
    assert np.array_equal(arrs[5], x[5000:5300])


@pytest.mark.parametrize('segs', ['1', '4'])
def test_lz_resolver_retry_pass(monkeypatch, segs):
    """A resolver wait that expires is not a verdict: the chunk is resolved again by a single worker wave.  The hook makes
    every chunk's first run give up at once, so the retry pass does all the work (byte and segmented resolver)."""
    monkeypatch.setenv('MTS_LZ_FORCE_RETRY', '1')
    monkeypatch.setenv('MTS_LZ_SEGS', segs)
    for data in (CASES['repeats_200k'], CASES['farcopies_900k'], CASES['zeros_999468'], inputs.ar1_stream(3000, 64)):
        st, out = hip.debug_inflate(zlib.compress(data, 6), len(data))
        assert st == 0 and out == data
    bad = bytearray(zlib.compress(CASES['text_100k'], 6))
    bad[len(bad) // 2] ^= 0x40
    st, _ = hip.debug_inflate(bytes(bad), len(CASES['text_100k']))
    assert st != 0


@pytest.mark.parametrize('env', [{}, {'MTS_INF_NO_ROWS': '1'}, {'MTS_INF_ROW_ROUNDS': '3'}, {'MTS_INF_ROW_ROUNDS': '0'}])
def test_inflate_token_rows(monkeypatch, env):
    """Pass A keeps the tokens of its counting decode in rows (192 per sub-sequence, from a pool); pass B only decodes
    the blocks whose tokens did not fit.  Same bytes with the rows (default), without them, with a pool that runs out
    after three rounds and with no pool at all; the inputs include rows that overflow (runs: 33 pieces per 258-byte
    copy), blocks of every size, and streams of other encoders' block structure."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    big = inputs.repeats(700000, 11) + inputs.textlike(300000, 12) + inputs.farcopies(400000, 13) + bytes(300000) + inputs.skewlen(200000, 5)
    for data in (CASES['repeats_200k'], CASES['text_100k'], CASES['zeros_999468'], big, inputs.ar1_stream(6000, 64)):
        for level in (1, 6, 9):
            st, out = hip.debug_inflate(zlib.compress(data, level), len(data))
            assert st == 0 and out == data, (len(data), level, st, _first_diff(np.frombuffer(out, np.uint8), np.frombuffer(data, np.uint8)))
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8)
    data = inputs.ar1_stream(4000, 64)
    z = b''.join(co.compress(data[i:i + 70000]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(data), 70000)) + co.flush()
    st, out = hip.debug_inflate(z, len(data))
    assert st == 0 and out == data
    bad = bytearray(zlib.compress(inputs.ar1_stream(3000, 64), 6))
    bad[len(bad) // 2] ^= 0x10
    st, _ = hip.debug_inflate(bytes(bad), 3000 * 64 * 2)
    assert st != 0


def _timed_inflate(z, n):
    hip.debug_inflate(z[:64] if len(z) > 64 else z, 1)          # (workspaces and code objects warm)
    t0 = time.perf_counter()
    st, out = hip.debug_inflate(z, n)
    return st, out, time.perf_counter() - t0


def test_inflate_writer_made_chunk_stays_on_the_fast_path():
    """A chunk as the Writer makes it (zlib level 6, dynamic blocks) is opened up by the block-start scan and its header
    validator: the block-after-block decoder finds nothing left to do.  (The scan's candidates are hints -- if the validator
    lost them the chunk would still decode, through that decoder, two orders of magnitude slower: this is the test that notices.)"""
    n = 385 * 30000 * 2
    ar = inputs.ar1_stream(30000, 385)
    for level in (6, 9, 1):
        z = zlib.compress(ar, level)
        st, out, dt = _timed_inflate(z, n)
        assert st == 0 and out == ar
        ms = dict(hip.last_stage_times())
        assert ms['inflate_wave_decoder'] < 1.0, (level, ms)
        assert dt < 0.25, (level, dt)


def test_inflate_final_block_outside_the_scan_zone(monkeypatch):
    """The block-start scan looks at candidates with BFINAL = 1 only near the end of a stream (the last 256 KiB: a final block
    starts within a block's length of the end).  With the zone shrunk below a block's length (MTS_SCAN_FINAL_ZONE = bytes) the
    final block of a Writer-made chunk is NOT announced: the chunk must still decode to the same bytes -- the chain walk hands the
    unannounced block to the block-after-block decoder -- and damage in it must still be refused."""
    n = 64 * 30000 * 2
    ar = inputs.ar1_stream(30000, 64)
    z = zlib.compress(ar, 6)
    st, out, _ = _timed_inflate(z, n)
    assert st == 0 and out == ar
    base = dict(hip.last_stage_times())
    monkeypatch.setenv('MTS_SCAN_FINAL_ZONE', '1024')
    st, out, _ = _timed_inflate(z, n)
    assert st == 0 and out == ar
    ms = dict(hip.last_stage_times())
    assert ms['inflate_wave_decoder'] > base['inflate_wave_decoder'], (base, ms)       # (the fallback did run: the test tests what it says)
    b = bytearray(z)
    b[len(b) - 3000] ^= 0x10                                         # inside the final block
    st, out, _ = _timed_inflate(bytes(b), n)
    assert st != 0
    monkeypatch.setenv('MTS_SCAN_FINAL_ZONE', '0')                   # no BFINAL = 1 candidate anywhere
    st, out, _ = _timed_inflate(z, n)
    assert st == 0 and out == ar


def test_host_entry_points_piece_by_piece(monkeypatch):
    """mts_compress_chunks / mts_decompress_chunks work piece by piece (the next piece's copy in and the last one's copy out on
    helper threads beside the kernels): MTS_PIPE_BYTES = 1 MiB cuts 11 chunks of 0.3 .. 1.2 MB into ~7 pieces; same bytes as the
    reference's statements, same verdicts -- a damaged chunk in a middle piece is refused, its neighbours are not --, and the
    same again as one piece (MTS_PIPE_BYTES = 0) and from page-locked memory."""
    r = np.random.RandomState(19)
    chunks = [synth_rows(int(n), 64, 40 + k) for k, n in enumerate(r.randint(2400, 9600, size=11))]
    rows = [c.shape[0] for c in chunks]
    bounds = np.concatenate(([0], np.cumsum(rows)))
    x = np.concatenate(chunks, axis=0)
    want = [O.ref_compress_chunk(c) for c in chunks]
    for pipe in ('1048576', '0', '300000'):
        monkeypatch.setenv('MTS_PIPE_BYTES', pipe)
        cbufs = hip.compress_chunks(x, bounds, 5, 6)
        assert [bytes(z) for z in cbufs] == want, pipe
        st, back = hip.decompress_chunks(cbufs, rows, 64, 'int16', 5)
        assert st == [0] * len(chunks) and all(np.array_equal(c, b) for c, b in zip(chunks, back)), pipe
        # chunks handed over as ONE range of a file (what Reader.tofile passes): (buffer, offsets, lengths)
        blob = b''.join(want)
        offs = np.concatenate(([0], np.cumsum([len(w) for w in want])))[:-1]
        bad = bytearray(blob)
        bad[int(offs[5]) + len(want[5]) // 2] ^= 0x40
        st, back = hip.decompress_chunks((bytes(bad), list(offs), [len(w) for w in want]), rows, 64, 'int16', 5)
        assert [s != 0 for s in st] == [k == 5 for k in range(len(chunks))], (pipe, st)
        assert all(np.array_equal(c, b) for k, (c, b) in enumerate(zip(chunks, back)) if k != 5)
    # page-locked memory in and out (the DMA reads and writes it directly)
    monkeypatch.setenv('MTS_PIPE_BYTES', '1048576')
    src, dst = hip.HostBuffer(x.nbytes), hip.HostBuffer(x.nbytes)
    try:
        src.array[:] = x.reshape(-1).view(np.uint8)
        cbufs = hip.compress_chunks(src.array.view(np.int16).reshape(x.shape), bounds, 5, 6)
        assert [bytes(z) for z in cbufs] == want
        st, back = hip.decompress_chunks(cbufs, rows, 64, 'int16', 5, out=dst.array.view(np.int16).reshape(x.shape))
        assert st == [0] * len(chunks) and np.array_equal(dst.array.view(np.int16).reshape(x.shape), x)
    finally:
        src.free(); dst.free()


def test_inflate_full_size_streams_without_dynamic_blocks():
    """Chunks of the headline size (385 x 30000 int16 = 23.1 MB) whose streams give the block-start scan nothing to find:
    stored blocks (incompressible data, level 0) and fixed-Huffman blocks (Z_FIXED) go through the wave decoder, not one lane."""
    n = 385 * 30000 * 2
    rnd = np.random.RandomState(7).randint(0, 256, n, dtype=np.uint8).tobytes()
    ar = inputs.ar1_stream(30000, 385)
    assert len(ar) == n
    for name, z, budget in (('uniform random, level 6 (stored blocks)', zlib.compress(rnd, 6), 0.25),
                            ('level 0', zlib.compress(ar, 0), 0.25),
                            ('Z_FIXED', (lambda co: co.compress(ar) + co.flush())(zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)), 2.6),
                            ('Z_HUFFMAN_ONLY', (lambda co: co.compress(ar) + co.flush())(zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_HUFFMAN_ONLY)), 2.6)):
        want = rnd if name.startswith('uniform') else ar
        st, out, dt = _timed_inflate(z, n)
        assert st == 0 and out == want, name
        print('inflate %s: %.1f ms' % (name, dt * 1e3))
        assert dt < budget, (name, dt)          # (host copies of 23 MB each way included; round 1's one-lane decoder took seconds to minutes)


def test_inflate_full_size_damage_is_refused_quickly():
    n = 385 * 30000 * 2
    ar = inputs.ar1_stream(30000, 385)
    z = zlib.compress(ar, 6)
    r = np.random.RandomState(11)
    for k in range(6):
        b = bytearray(z)
        i = int(r.randint(2, len(b) - 4)) if k else 1000          # (an early flip: nearly the whole chunk lies behind the damage)
        b[i] ^= 1 << int(r.randint(0, 8))
        try:
            want = zlib.decompress(bytes(b))
        except zlib.error:
            want = None
        st, out, dt = _timed_inflate(bytes(b), n)
        if want is None or len(want) != n:
            assert st != 0, i
        else:
            assert st == 0 and out == want, i
        print('inflate damaged at byte %d of %d: %.1f ms' % (i, len(z), dt * 1e3))
        assert dt < 0.4, (i, dt)


def test_level9_long_zero_runs_many_times():
    """A chain budget of 4096 makes the match stage's workgroups look back for the start of a run in SEVERAL rounds (512 slots per round)
    when the run reaches that far -- all-zero chunks.  Round 4's fuzzer found the rounds' exit racing with the next round's atomic
    (waves left the loop at different barriers; the run-order guard then fired at random and, one call in five, twice in a row =
    MTS_E_INTERNAL).  The same call many times: every one must succeed and give zlib's bytes."""
    x = np.zeros((2 * 7500, 1024), dtype=np.int32)
    flags = hip.make_flags(True, True, 'C')
    want = zlib.compress(bytes(O.delta_transpose(x[:7500], flags)), 9)
    for rep in range(16):
        got = hip.compress_chunks(x, [0, 7500, 15000], flags, 9)
        assert got[0] == want and got[1] == want, rep
