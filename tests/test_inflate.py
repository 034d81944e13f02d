"""The library's own DEFLATE decoder (inflate_fast.c, host C) against zlib: every block type, compression level and
strategy, edge sizes, and corrupted streams (must fail cleanly, never crash or write out of bounds)."""
import ctypes as C
import zlib

import numpy as np
import pytest

from fastf_amd import _lib

L = _lib.lib()
L.fastf_inflate_raw.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
L.fastf_inflate_raw.restype = C.c_int


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


def inflate(comp, n, guard=64):
    buf = np.full(n + 2 * guard, 0xAB, dtype=np.uint8)
    rc = L.fastf_inflate_raw(comp, len(comp), buf.ctypes.data + guard, n)
    assert (buf[:guard] == 0xAB).all() and (buf[guard + n:] == 0xAB).all(), "wrote outside the output buffer"
    return rc, bytes(buf[guard:guard + n])


def corpus():
    rng = np.random.default_rng(1)
    out = [b"", b"a", b"ab" * 3, b"\0" * 70000, bytes(range(256)) * 200]
    out.append(rng.integers(0, 256, 65280, dtype=np.uint8).tobytes())                 # incompressible → stored blocks
    out.append(rng.integers(0, 4, 65280, dtype=np.uint8).tobytes())                   # low entropy
    out.append(b"".join(b"r%d\tCB:Z:%s-1\txf:i:25\tGX:Z:ENSG%011d\tUB:Z:%s\n" % (
        i, bytes(rng.choice(list(b"ACGT"), 16).tolist()), int(rng.integers(1, 30000)), bytes(rng.choice(list(b"ACGT"), 10).tolist()))
        for i in range(3000)))                                                         # BAM-like text
    z = rng.zipf(1.3, 60000).astype(np.uint32) % 251
    out.append(z.astype(np.uint8).tobytes())                                           # skewed: long Huffman codes
    out.append(b"abcdefgh" * 8000 + rng.integers(0, 256, 1000, dtype=np.uint8).tobytes() + b"xyz" * 5000)
    for n in (1, 2, 3, 7, 8, 9, 257, 258, 259, 32767, 32768, 32769, 65535):
        out.append((b"ACGT" * (n // 4 + 1))[:n])
    return out


@pytest.mark.parametrize("level", [0, 1, 2, 4, 6, 9])
def test_matches_zlib_levels(level):
    for data in corpus():
        comp = raw_deflate(data, level)
        rc, got = inflate(comp, len(data))
        assert rc == 0 and got == data, (level, len(data))


@pytest.mark.parametrize("strategy", [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
def test_matches_zlib_strategies(strategy):
    for data in corpus():
        for memlevel in (1, 9):
            comp = raw_deflate(data, 6, strategy, memlevel=memlevel)
            rc, got = inflate(comp, len(data))
            assert rc == 0 and got == data, (strategy, memlevel, len(data))


def test_multi_block_streams_and_small_windows():
    rng = np.random.default_rng(2)
    data = rng.integers(0, 8, 200000, dtype=np.uint8).tobytes()
    c = zlib.compressobj(6, zlib.DEFLATED, -9)                    # 512-byte window
    comp = b"".join(c.compress(data[i:i + 5000]) + c.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(data), 5000)) + c.flush()
    rc, got = inflate(comp, len(data))
    assert rc == 0 and got == data


def test_wrong_output_size_is_rejected():
    data = b"hello hello hello hello" * 100
    comp = raw_deflate(data)
    assert inflate(comp, len(data) - 1)[0] != 0
    assert inflate(comp, len(data) + 1)[0] != 0
    assert inflate(comp[:-3], len(data))[0] != 0                     # truncated input
    assert inflate(b"", 0)[0] != 0


def test_corrupted_streams_fail_cleanly():
    rng = np.random.default_rng(3)
    datas = [c for c in corpus() if 100 < len(c) < 70000][:6]
    n_fail = n_total = 0
    for data in datas:
        comp = bytearray(raw_deflate(data, 6))
        for _ in range(300):
            bad = bytearray(comp)
            for _ in range(int(rng.integers(1, 4))):
                bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            rc, got = inflate(bytes(bad), len(data))                 # guard bytes are checked inside
            n_total += 1
            n_fail += (rc != 0) or (got != data)
    assert n_fail > 0.5 * n_total                                    # most corruptions are noticed even before the CRC
    for _ in range(300):                                             # pure noise
        junk = rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8).tobytes()
        inflate(junk, int(rng.integers(0, 5000)))


def test_crc32_matches_zlib():
    """the carry-less-multiplication CRC-32 of the BGZF reader (crc32_fast.c) against zlib on every length class"""
    import ctypes as C
    import zlib as Z
    L = _lib.lib()
    L.fastf_crc32.argtypes = [C.c_char_p, C.c_size_t]
    L.fastf_crc32.restype = C.c_uint32
    L.fastf_crc32_selftest.restype = C.c_int
    assert L.fastf_crc32_selftest() in (0, 1)
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes()
    for n in list(range(0, 200)) + [255, 256, 1000, 4095, 4096, 65280, 65535, 65536, 299_999]:
        for off in (0, 1, 7):
            b = data[off:off + n]
            assert L.fastf_crc32(b, len(b)) == Z.crc32(b)
