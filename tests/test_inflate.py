"""CPU: the host's own DEFLATE decoder and CRC-32 (genrich_amd/host/gx_inflate.h, gx_crc32.h -- what bgzf_reader.h
inflates and checks BGZF blocks with) against zlib on the same bytes: every block type, compression level and strategy,
sizes around the decoder's fast / checked boundary, and damaged streams, which it must reject or decode exactly as zlib
does -- never accept with other bytes (the reader falls back to zlib for whatever it rejects)."""
import ctypes as C
import random
import zlib

import pytest


@pytest.fixture(scope="module")
def lib():
    from genrich_amd import build
    L = C.CDLL(build.build_inflate_test())
    L.gx_fast_inflate.argtypes = [C.c_char_p, C.c_ulong, C.c_char_p, C.c_ulong]
    L.gx_fast_crc32.argtypes = [C.c_char_p, C.c_ulong]
    L.gx_fast_crc32.restype = C.c_uint
    return L


def _inflate(L, comp, n):
    out = C.create_string_buffer(max(n, 1))
    ok = L.gx_fast_inflate(comp, len(comp), out, n)
    return bool(ok), out.raw[:n]


def _deflate(data, level, strategy, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
    return c.compress(data) + c.flush()


def _sample(rng, kind, n):
    if kind == 0:
        return bytes(rng.getrandbits(8) for _ in range(n))
    if kind == 1:
        return bytes(rng.choice(b"ACGT") for _ in range(n))
    if kind == 2:
        return ((b"chr1\t12345\tread_name_%d\t99\t60\t100M\t=\t" % rng.randrange(10 ** 6)) * (n // 30 + 1))[:n]
    if kind == 3:
        return bytes([rng.randrange(4)]) * n
    b = bytearray()  # literals and copies of every distance / length
    while len(b) < n:
        if b and rng.random() < 0.5:
            o, l = rng.randrange(1, min(len(b), 32768) + 1), rng.randrange(3, 300)
            for _ in range(l):
                b.append(b[-o])
        else:
            b += bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 40)))
    return bytes(b[:n])


@pytest.mark.parametrize("seed", range(4))
def test_decoder_equals_zlib_on_valid_streams(lib, seed):
    rng = random.Random(seed)
    for it in range(400):
        n = rng.choice([0, 1, 2, 3, 7, 8, 9, 100, 257, 258, 259, 273, 274, 275, 1000, 4096, 30000, 65280, 65535, 65536])
        if rng.random() < 0.5:
            n = rng.randrange(0, 65537)
        data = _sample(rng, rng.randrange(5), n)
        comp = _deflate(data, rng.choice([0, 1, 2, 4, 6, 9]),
                        rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]),
                        rng.choice([1, 4, 8, 9]))
        ok, out = _inflate(lib, comp, len(data))
        assert ok and out == data, (seed, it, n)
        if data:  # the block's stated size is part of the contract
            assert not _inflate(lib, comp, len(data) - 1)[0]
            assert not _inflate(lib, comp, len(data) + 1)[0]


def test_several_blocks_in_one_member(lib):
    rng = random.Random(9)
    data = b"".join(_sample(rng, k % 5, 3000 + 700 * k) for k in range(8))
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = b""
    for k in range(0, len(data), 5000):  # full flushes: stored empty blocks between dynamic ones
        comp += c.compress(data[k:k + 5000]) + c.flush(zlib.Z_FULL_FLUSH if k % 10000 else zlib.Z_SYNC_FLUSH)
    comp += c.flush()
    ok, out = _inflate(lib, comp, len(data))
    assert ok and out == data


@pytest.mark.parametrize("seed", range(3))
def test_damaged_streams_are_rejected_or_decoded_as_zlib_does(lib, seed):
    rng = random.Random(100 + seed)
    both = 0
    for it in range(1500):
        data = _sample(rng, rng.randrange(5), rng.randrange(1, 20000))
        comp = bytearray(_deflate(data, rng.choice([1, 6, 9]), zlib.Z_DEFAULT_STRATEGY))
        mode = rng.randrange(3)
        if mode == 0:
            for _ in range(rng.randrange(1, 4)):
                comp[rng.randrange(len(comp))] ^= 1 << rng.randrange(8)
        elif mode == 1:
            comp = comp[:rng.randrange(len(comp) + 1)]
        else:
            comp = bytearray(rng.getrandbits(8) for _ in range(rng.randrange(0, 300)))
        comp = bytes(comp)
        try:
            d = zlib.decompressobj(-15)
            ref = d.decompress(comp)
            zok = d.eof
        except zlib.error:
            ref, zok = None, False
        ok, out = _inflate(lib, comp, len(ref) if zok else len(data))
        if ok:
            assert zok and out == ref, (seed, it, mode)
            both += 1
    assert both > 50  # (bit flips in literals leave a valid stream: those must decode, to zlib's bytes)


def test_crc32_equals_zlib(lib):
    rng = random.Random(5)
    for n in list(range(0, 300)) + [rng.randrange(0, 70000) for _ in range(300)] + [65280, 65535, 65536]:
        d = bytes(rng.getrandbits(8) for _ in range(n))
        assert lib.gx_fast_crc32(d, n) == zlib.crc32(d) & 0xFFFFFFFF, n
