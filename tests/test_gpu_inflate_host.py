"""The device inflate decoder (fastf_amd/csrc/gpu_inflate.hpp) compiled for the host with one lane (tools/gi_host.cpp)
and fuzzed against zlib: every block type (stored, fixed, dynamic), every zlib strategy and level, many small deflate
blocks, full flushes, codes longer than the direct tables, malformed input.  The same source runs on the GPU, one
wavefront per BGZF block (tests/test_gpu_inflate.py checks that build on the device)."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gi():
    so = os.path.join(ROOT, "build", "libgi_host.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fastf_amd", "csrc"), so])
    L = C.CDLL(so)
    L.gi_host_inflate.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32]

    def inflate(comp, n):
        out = C.create_string_buffer(max(n, 1))
        rc = L.gi_host_inflate(comp + bytes(16), len(comp), out, n)
        return rc, out.raw[:n]
    return inflate


def raw_streams(data):
    for lvl in (0, 1, 6, 9):
        for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, strat)
            yield co.compress(data) + co.flush()
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 1)                 # memLevel 1: many small deflate blocks
    yield co.compress(data) + co.flush()
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = []
    for i in range(0, len(data), 5000):
        parts += [co.compress(data[i:i + 5000]), co.flush(zlib.Z_FULL_FLUSH)]
    yield b"".join(parts) + co.flush()


def sample_payloads(rng, n_random):
    bam_like = b"".join(b"r%d\tCB:Z:ACGT%dTTGA-1\tUB:Z:GGG%d\n" % (i, i * 7919 % 10007, i * 31 % 977) for i in range(2500))[:65280]
    skew = bytes(rng.permutation(np.repeat(np.arange(40, dtype=np.uint8), (1.35 ** np.arange(40)).astype(int) + 1)))[:65280]   # codes > 10 bits
    out = [b"", b"a", b"abc" * 5, bytes(65280), rng.integers(0, 256, 65280, dtype=np.uint8).tobytes(),
           rng.integers(0, 4, 65280, dtype=np.uint8).tobytes(), (b"ACGTACGTTTGA" * 7000)[:65280], bytes(range(256)) * 255, bam_like, skew]
    for _ in range(n_random):
        n = int(rng.integers(0, 65281)); a = int(rng.integers(1, 257))
        d = bytearray(rng.integers(0, a, n, dtype=np.uint8).tobytes())
        if rng.random() < 0.5 and n > 400:
            for _ in range(50):
                i = int(rng.integers(0, n - 300)); l = int(rng.integers(3, 300)); j = int(rng.integers(0, n - 300))
                d[j:j + l] = d[i:i + l]
        out.append(bytes(d[:n]))
    return out


def test_host_build_matches_zlib(gi):
    rng = np.random.default_rng(3)
    n = 0
    for data in sample_payloads(rng, 40):
        for comp in raw_streams(data):
            rc, out = gi(comp, len(data))
            assert rc == 0 and out == data
            n += 1
    assert n > 1000


def test_host_build_rejects_malformed_input_without_crashing(gi):
    rng = np.random.default_rng(4)
    data = sample_payloads(rng, 0)[8]
    comp = zlib.compress(data, 6)[2:-4]
    assert gi(comp[:len(comp) // 2], len(data))[0] != 0             # truncated
    assert gi(comp, len(data) - 1)[0] != 0 and gi(comp, len(data) + 1)[0] != 0   # wrong ISIZE
    for _ in range(300):                                              # bit flips: an error, or output the CRC check will reject
        c = bytearray(comp); i = int(rng.integers(0, len(c))); c[i] ^= 1 << int(rng.integers(0, 8))
        rc, out = gi(bytes(c), len(data))
        assert rc != 0 or len(out) == len(data)


# ---- the lane-per-block decoder of the two-kernel device inflate (gpu_inflate2.hpp): plain single-threaded C++, the same source
#      the device runs per lane; tools/gi2_host.cpp decodes to literals + match tokens and applies the tokens one by one ----
@pytest.fixture(scope="module")
def gi2():
    so = os.path.join(ROOT, "build", "libgi2_host.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fastf_amd", "csrc"), so])
    L = C.CDLL(so)
    L.gi2_host_inflate.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]

    def inflate(comp, n):
        out = C.create_string_buffer(max(n, 1)); nt = C.c_uint32()
        rc = L.gi2_host_inflate(comp, len(comp), out, n, C.byref(nt))
        return rc, out.raw[:n], nt.value
    return inflate


def test_token_decoder_matches_zlib(gi2):
    rng = np.random.default_rng(3)
    n = 0
    for data in sample_payloads(rng, 40):
        for comp in raw_streams(data):
            rc, out, nt = gi2(comp, len(data))
            assert rc == 0 and out == data
            assert nt <= len(data) // 3 + len(data) // 256 + 4
            n += 1
    assert n > 1000


def test_token_decoder_edge_shapes(gi2):
    """a block of 65 536 literals (a literal run that needs two step-over tokens if a stored block ends it), long literal runs in
    front of a match, an overlapping match of every short distance, stored blocks between compressed ones"""
    rng = np.random.default_rng(9)
    lit = rng.integers(0, 256, 65536, dtype=np.uint8).tobytes()
    co = zlib.compressobj(0, zlib.DEFLATED, -15)                     # stored blocks only
    rc, out, nt = gi2(co.compress(lit) + co.flush(), len(lit)); assert rc == 0 and out == lit
    co = zlib.compressobj(9, zlib.DEFLATED, -15, 9, zlib.Z_HUFFMAN_ONLY)   # literals only, one Huffman block
    rc, out, nt = gi2(co.compress(lit) + co.flush(), len(lit)); assert rc == 0 and out == lit and nt <= 2       # (zlib stores what does not compress: step-over tokens at most)
    for d in (1, 2, 3, 5, 63, 64, 65, 257, 258, 259):
        data = (rng.integers(0, 256, d, dtype=np.uint8).tobytes() * (70000 // d + 1))[:65000]
        for lvl in (1, 6, 9):
            co = zlib.compressobj(lvl, zlib.DEFLATED, -15)
            rc, out, nt = gi2(co.compress(data) + co.flush(), len(data)); assert rc == 0 and out == data
    # 300..60000 incompressible bytes, then a repeat of them (a long literal run in front of matches), then a stored block
    for n in (300, 5000, 30000):
        head = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        data = head + head[:n // 2] + b"xyz" * 100
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(data) + co.flush(zlib.Z_FULL_FLUSH)
        co0 = zlib.compressobj(0, zlib.DEFLATED, -15)
        tail = rng.integers(0, 256, 1000, dtype=np.uint8).tobytes()
        comp += co0.compress(tail) + co0.flush()
        rc, out, nt = gi2(comp, len(data) + len(tail)); assert rc == 0 and out == data + tail


def test_token_decoder_rejects_malformed_input_without_crashing(gi2):
    rng = np.random.default_rng(4)
    data = sample_payloads(rng, 0)[8]
    comp = zlib.compress(data, 6)[2:-4]
    assert gi2(comp[:len(comp) // 2], len(data))[0] != 0
    assert gi2(comp, len(data) - 1)[0] != 0 and gi2(comp, len(data) + 1)[0] != 0
    for _ in range(600):
        c = bytearray(comp); i = int(rng.integers(0, len(c))); c[i] ^= 1 << int(rng.integers(0, 8))
        rc, out, nt = gi2(bytes(c), len(data))
        assert rc != 0 or len(out) == len(data)
    for _ in range(200):                                              # garbage
        g = rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8).tobytes()
        gi2(g, int(rng.integers(0, 65537)))


def _bits(*fields):
    """(value, width) fields, LSB first, into bytes (DEFLATE's bit order)"""
    acc = n = 0
    for v, w in fields:
        acc |= v << n; n += w
    return acc.to_bytes((n + 7) // 8 + 8, "little")


def test_token_decoder_rejects_the_incomplete_code_sets_zlib_rejects(gi2):
    """zlib (inftrees.c) takes an incomplete Huffman set in ONE form only: a literal/length or distance alphabet with a single code
    of length 1.  An incomplete code-length alphabet, or a single code of another length, is "invalid code lengths set" there
    and must not decode here either (this decoder used to accept a single code of any length, in any alphabet)."""
    # BFINAL=1, BTYPE=2, HLIT=0 (257), HDIST=0 (1), HCLEN=0 (4 code-length codes: symbols 16, 17, 18, 0)
    # the code-length code: only symbol 0, with length 2 -> incomplete
    bad_cl = _bits((1, 1), (2, 2), (0, 5), (0, 5), (0, 4), (0, 3), (0, 3), (0, 3), (2, 3))
    assert zlib_rejects(bad_cl)
    assert gi2(bad_cl, 16)[0] != 0
    # ... with length 1: still incomplete, still rejected for THIS alphabet
    bad_cl1 = _bits((1, 1), (2, 2), (0, 5), (0, 5), (0, 4), (0, 3), (0, 3), (0, 3), (1, 3))
    assert zlib_rejects(bad_cl1)
    assert gi2(bad_cl1, 16)[0] != 0


def zlib_rejects(raw):
    try:
        zlib.decompressobj(-15).decompress(raw)
    except zlib.error:
        return True
    return False
