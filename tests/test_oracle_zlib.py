"""Pins the C oracle (oracle/mtsc_oracle.c) to the libz the reference calls (stdlib zlib, 1.2.11).

The reference holds no compressed-byte known answers; these differential tests and the
golden fixtures (test_golden.py) are what pins byte parity."""
import zlib

import numpy as np
import pytest

from oracle import oracle as O
from tests.inputs import cases_small, repeats, textlike

CASES = cases_small()


def test_zlib_version():
    # byte parity is defined against stock zlib 1.2.11 (what this image ships)
    assert zlib.ZLIB_RUNTIME_VERSION.startswith('1.2.11') or zlib.ZLIB_RUNTIME_VERSION.startswith('1.2.1') \
        or zlib.ZLIB_RUNTIME_VERSION.startswith('1.3')


@pytest.mark.parametrize('name', sorted(CASES))
@pytest.mark.parametrize('level', [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_deflate_bytes_equal_zlib(name, level):
    data = CASES[name]
    assert O.deflate(data, level) == zlib.compress(data, level)


def test_deflate_default_level_is_6():
    data = CASES['text_100k']
    assert O.deflate(data, -1) == zlib.compress(data) == zlib.compress(data, 6)


@pytest.mark.parametrize('seed', range(12))
def test_deflate_fuzz_lengths(seed):
    r = np.random.RandomState(100 + seed)
    n = int(r.randint(0, 150000))
    kind = seed % 3
    data = [repeats, textlike][kind % 2](n, seed) if kind < 2 else \
        (r.randint(0, 256, size=n) * (r.randint(0, 8, size=n) == 0)).astype(np.uint8).tobytes()
    for level in (1, 3, 6, 9):
        assert O.deflate(data, level) == zlib.compress(data, level), (seed, level, n)


def test_block_boundary_corners():
    # streams whose token count lands on / next to multiples of 16383 (lit_bufsize - 1): exercises
    # the "trailing literal does not flush" and "empty final block" corners of deflate_slow.
    r = np.random.RandomState(7)
    base = r.randint(0, 256, size=16383 * 2 + 40).astype(np.uint8).tobytes()   # all literals
    for n in (16382, 16383, 16384, 16385, 32765, 32766, 32767, 32768):
        d = base[:n]
        for level in (1, 6):
            assert O.deflate(d, level) == zlib.compress(d, level), (n, level)
    # last token is a match that fills the block exactly
    for extra in range(3, 12):
        d = base[:16382] + base[100:100 + extra]
        assert O.deflate(d, 6) == zlib.compress(d, 6), extra


@pytest.mark.parametrize('name', ['ar1_8ch', 'text_100k', 'repeats_200k', 'zeros_70k', 'rand_70k', 'first50'])
@pytest.mark.parametrize('level', [4, 6, 9])
def test_table_formulation_equals_sequential(name, level):
    """candidate tables + state machine reproduce deflate_slow token for token."""
    data = CASES[name]
    _, toks, tokpos, _ = O.deflate(data, level, report=True)
    tf, tq = O.match_tables(data, level)
    toks2, tokpos2 = O.parse_tables(data, tf, tq, level)
    assert np.array_equal(toks, toks2)
    assert np.array_equal(tokpos, tokpos2)


def test_block_report_consistent():
    data = CASES['ar1_64ch_4k']
    z, toks, tokpos, blocks = O.deflate(data, 6, report=True)
    assert z == zlib.compress(data, 6)
    assert sum(b['ntok'] for b in blocks) == len(toks)
    assert all(b['ntok'] == 16383 for b in blocks[:-1])
    assert blocks[-1]['last'] == 1 and blocks[0]['bit_start'] == 16
    assert sum(b['in_len'] for b in blocks) == len(data)
    for b in blocks:
        assert b['in_start'] == (tokpos[b['tok_start']] if b['ntok'] else len(data))


def test_slides_closed_form():
    # closed form used by the HIP path == number of thresholds crossed (see orc_slides_at)
    for n in (0, 10, 65273, 65274, 65275, 65536, 65537, 98041, 98042, 98043, 98304, 98305, 200000):
        k_prev = 0
        for q in range(0, n + 1, 1 if n < 70000 else 97):
            k = O.slides_at(q, n)
            assert k >= k_prev and k - k_prev <= 1
            k_prev = k
        if n >= 65536:
            assert O.slides_at(65274, n) == 0 and O.slides_at(65275, n) == 1
        elif n > 65274:
            assert O.slides_at(65273, n) == 0 and O.slides_at(65274, n) == 1


@pytest.mark.parametrize('name', sorted(CASES))
def test_inflate_roundtrip_all_levels(name):
    data = CASES[name]
    for level in (0, 1, 6, 9):
        z = zlib.compress(data, level)
        out, used = O.inflate(z, len(data))
        assert out == data and used == len(z)
        out, used = O.inflate(z + b'trailing garbage', len(data))     # trailing bytes ignored
        assert out == data and used == len(z)


def test_inflate_other_encoders():
    data = CASES['text_100k']
    for strategy in (zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
        co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, strategy)
        z = co.compress(data) + co.flush()
        assert O.inflate(z, len(data))[0] == data
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8)
    z = b''.join(co.compress(data[i:i + 5000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(data), 5000))
    z += co.flush()
    assert O.inflate(z, len(data))[0] == data
    co = zlib.compressobj(6, zlib.DEFLATED, 9, 1)                    # 512-byte window, memLevel 1
    z = co.compress(data) + co.flush()
    assert O.inflate(z, len(data))[0] == data


def test_inflate_errors_match_zlib():
    data = CASES['ar1_8ch']
    z = zlib.compress(data)
    bad = {
        'empty': b'',
        'trunc_half': z[:len(z) // 2],
        'trunc_adler': z[:-2],
        'bad_adler': z[:-1] + bytes([z[-1] ^ 1]),
        'bad_header': b'\x79' + z[1:],
        'bad_fcheck': z[:1] + bytes([z[1] ^ 1]) + z[2:],
        'flip_mid': z[:len(z) // 2] + bytes([z[len(z) // 2] ^ 0x55]) + z[len(z) // 2 + 1:],
        'reserved_btype': b'\x78\x9c\x07',
    }
    for name, b in bad.items():
        with pytest.raises(zlib.error):
            zlib.decompress(b)
        with pytest.raises(ValueError):
            O.inflate(b, len(data) + 10)
    r = np.random.RandomState(5)
    agree = 0
    for _ in range(300):
        b = bytearray(z)
        i = int(r.randint(2, len(b) - 4))
        b[i] ^= 1 << int(r.randint(0, 8))
        try:
            want = zlib.decompress(bytes(b))
        except zlib.error:
            want = None
        try:
            got = O.inflate(bytes(b), len(data) + 1000)[0]
        except ValueError:
            got = None
        assert got == want
        agree += 1
    assert agree == 300


def test_adler32():
    for name, data in CASES.items():
        assert O.adler32(data) == zlib.adler32(data), name


@pytest.mark.parametrize('dtype', ['int16', 'uint16', 'uint8', 'int8', 'int32', 'int64'])
@pytest.mark.parametrize('flags', range(8))
def test_transforms_equal_numpy(dtype, flags):
    r = np.random.RandomState(flags)
    info = np.iinfo(dtype)
    x = r.randint(info.min, int(info.max) + 1, size=(37, 11), dtype=np.int64).astype(dtype)
    td, sd, order = bool(flags & 1), bool(flags & 2), 'F' if flags & 4 else 'C'
    d = O.ref_diff_along_axis(x, 0 if td else None)
    d = O.ref_diff_along_axis(d, 1 if sd else None)
    want = d.tobytes(order=order)
    got = O.delta_transpose(x, flags)
    assert got.tobytes() == want
    back = O.cumsum_transpose(got, 37, 11, dtype, flags)
    assert back.dtype == x.dtype and np.array_equal(back, x)


def test_chunk_restatements_agree():
    from mtscomp_amd.synth import synth_int16
    x = synth_int16(0, 2500, 16, 3)
    for flags in range(8):
        td, sd, order = bool(flags & 1), bool(flags & 2), 'F' if flags & 4 else 'C'
        ref = O.ref_compress_chunk(x, td, sd, order)
        assert O.compress_chunk(x, flags, 6) == ref
        rc, back = O.decompress_chunk(ref, 2500, 16, 'int16', flags)
        assert rc == 0 and np.array_equal(back, x)
        assert np.array_equal(O.ref_decompress_chunk(ref, 2500, 16, 'int16', td, sd, order), x)
    rc, _ = O.decompress_chunk(ref[:100], 2500, 16, 'int16', 7)
    assert rc < 0
    rc, _ = O.decompress_chunk(ref, 2499, 16, 'int16', 7)
    assert rc == 1
