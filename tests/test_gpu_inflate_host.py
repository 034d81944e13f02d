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
