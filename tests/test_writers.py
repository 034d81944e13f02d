"""Chunk-parallel gzip writers (host C, no GPU): the decompressed bytes of all four outputs equal the
oracle's text for COO/umi rows spanning several compression chunks."""
import ctypes as C
import gzip

import numpy as np
import pytest

import fastf_amd as F
from fastf_amd import _lib
from fastf_amd._lib import Coo, UmiRows, ListsStruct
from helpers import Case
from oracle import oracle as O


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint32))


@pytest.mark.parametrize("threads,level", [("1", None), ("7", "1"), ("3", "9")])
def test_write_outputs_bytes(tmp_path, monkeypatch, threads, level):
    monkeypatch.setenv("FASTF_HOST_THREADS", threads)
    if level:
        monkeypatch.setenv("FASTF_GZIP_LEVEL", level)
    case = Case(n=700_000, n_bar=3000, n_gene=2000, umi_pool=4096, p_n_umi=0.01)     # ~600 k rows: 3 chunks
    ora = case.oracle()
    assert ora["nnz"] > 2 * (1 << 18) and ora["n_umi_rows"] > 2 * (1 << 18)
    lists = case.lists()
    L = _lib.lib()
    L.fastf_write_outputs.argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_float, C.POINTER(C.c_uint64 * 3),
                                      C.POINTER(ListsStruct), C.POINTER(Coo), C.POINTER(UmiRows)]
    f, fp = _u32(ora["feature"]); c, cp = _u32(ora["cell"]); k, kp = _u32(ora["count"])
    coo = Coo(fp, cp, kp, ora["nnz"])
    # umi rows: rebuild the engine-side representation (left-aligned 2-bit bases) from the oracle's text
    n = ora["n_umi_rows"]
    nonnull = np.array([t != b"NULL" for t in ora["umi_text"]], dtype=np.uint8)
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    umi = np.zeros(n, dtype=np.uint32)
    for i, t in enumerate(ora["umi_text"]):
        if nonnull[i]:
            v = 0
            for ch in t:
                v = (v << 2) | code[ch]
            umi[i] = v << 12
    uf, ufp = _u32(ora["umi_feature"]); uc, ucp = _u32(ora["umi_cell"]); un, unp = _u32(ora["umi_ncopy"])
    uu, uup = _u32(umi)
    rows = UmiRows(ufp, ucp, unp, uup, nonnull.ctypes.data_as(C.POINTER(C.c_uint8)), n)
    cnt = (C.c_uint64 * 3)(ora["total"], ora["sampled"], ora["valid"])
    rc = L.fastf_write_outputs(str(tmp_path).encode(), case.label, C.c_float(case.rate_cell), C.c_float(case.rate_depth),
                               C.byref(cnt), C.byref(lists._s), C.byref(coo), C.byref(rows))
    assert rc == 0, L.fastf_last_error()
    rd = lambda name: gzip.decompress((tmp_path / name).read_bytes())
    assert rd("matrix.mtx.gz") == ora["matrix"]
    assert rd("barcodes.tsv.gz") == ora["barcodes"]
    assert rd("features.tsv.gz") == ora["features"]
    assert rd("umi.tsv.gz") == ora["umi"]
    # zlib's gzread (what downstream C tools use) also sees one stream
    import zlib
    d = zlib.decompressobj(31); raw = (tmp_path / "matrix.mtx.gz").read_bytes(); out = b""
    while raw:
        out += d.decompress(raw); raw = d.unused_data
        if raw:
            d = zlib.decompressobj(31)
    assert out == ora["matrix"]


def test_write_outputs_empty(tmp_path):
    case = Case(n=10, n_bar=3, n_gene=2, rate_cell=0.0)
    ora = case.oracle()
    lists = case.lists()
    L = _lib.lib()
    L.fastf_write_outputs.argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_float, C.POINTER(C.c_uint64 * 3),
                                      C.POINTER(ListsStruct), C.POINTER(Coo), C.POINTER(UmiRows)]
    coo = Coo(None, None, None, 0)
    cnt = (C.c_uint64 * 3)(ora["total"], ora["sampled"], ora["valid"])
    assert L.fastf_write_outputs(str(tmp_path).encode(), case.label, C.c_float(0.0), C.c_float(1.0), C.byref(cnt),
                                 C.byref(lists._s), C.byref(coo), None) == 0
    assert gzip.decompress((tmp_path / "matrix.mtx.gz").read_bytes()) == ora["matrix"]
    assert gzip.decompress((tmp_path / "barcodes.tsv.gz").read_bytes()) == b""
    assert not (tmp_path / "umi.tsv.gz").exists()
    # unwritable directory: error code + message, no crash
    assert L.fastf_write_outputs(b"/nonexistent/dir", case.label, C.c_float(0.0), C.c_float(1.0), C.byref(cnt),
                                 C.byref(lists._s), C.byref(coo), None) == 1
    assert b"can not open file" in L.fastf_last_error()


# ---- the built-in deflate encoder (deflate_fast.c): any valid stream will do, so the check is "zlib inflates it back" ----
def _gz_fast(b):
    L = _lib.lib()
    L.fastf_gz_bound.restype = C.c_size_t; L.fastf_gz_bound.argtypes = [C.c_size_t]
    L.fastf_gz_member_fast.restype = C.c_size_t
    L.fastf_gz_member_fast.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    cap = L.fastf_gz_bound(len(b))
    out = C.create_string_buffer(cap)
    n = L.fastf_gz_member_fast(b, len(b), out, cap)
    assert n > 0
    return out.raw[:n]


def test_fast_deflate_round_trips_edge_cases():
    rng = np.random.default_rng(11)
    cells = np.sort(rng.integers(1, 25_001, 300_000)); feats = rng.integers(1, 36_602, 300_000)
    rows = b"".join(b"%d %d %d\n" % (f, c, 1 + (f % 7)) for f, c in zip(feats.tolist(), cells.tolist()))
    cases = {
        "empty": b"", "one byte": b"x", "three bytes": b"abc", "short repeat": b"hello hello hello hello",
        "zeros": bytes(700_000),                                   # matches of the maximum length 258, distance 1
        "random": rng.integers(0, 256, 400_000, dtype=np.uint8).tobytes(),      # incompressible: literal-only blocks
        "matrix rows": rows,                                       # several 128 K-token blocks
        "period 321": (b"abcdefgh" * 40 + b"X") * 3000,
        "far matches": (rng.integers(0, 256, 32_768, dtype=np.uint8).tobytes()) * 3,     # distance exactly 32 768
        "too far": (rng.integers(0, 256, 32_769, dtype=np.uint8).tobytes()) * 3,         # one beyond the window: no match allowed
        "two symbols": bytes(rng.integers(0, 2, 100_000, dtype=np.uint8)),
        # Fibonacci-like symbol frequencies: the unrestricted Huffman tree is deeper than 15 levels (and the code-length
        # code deeper than 7), so the length-limiting step has to rebalance it
        "skewed": bytes(rng.permutation(np.repeat(np.arange(24, dtype=np.uint8), [1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377,
                                                  610, 987, 1597, 2584, 4181, 6765, 10946, 17711, 28657, 46368]))),
        "skewed wide": bytes(rng.permutation(np.repeat(np.arange(120, dtype=np.uint8), (1.1 ** np.arange(120)).astype(int) + 1))),
        "all bytes": bytes(range(256)) * 50,
    }
    import zlib
    for name, b in cases.items():
        z = _gz_fast(b)
        d = zlib.decompressobj(31)
        assert d.decompress(z) == b and d.eof and d.unused_data == b"", name
        assert gzip.decompress(z) == b, name
    assert len(_gz_fast(rows)) < len(rows) / 2.5                   # digit rows: in the class of zlib's fast levels


def test_fast_deflate_fuzz():
    rng = np.random.default_rng(12)
    for it in range(400):
        n = int(rng.integers(0, 6000)); alpha = int(rng.integers(1, 257))
        b = rng.integers(0, alpha, n, dtype=np.uint8).tobytes()
        if it % 3 == 0:
            b = b * int(rng.integers(1, 30))
        if it % 7 == 0:
            b = b + b[: len(b) // 2][::-1] + b
        assert gzip.decompress(_gz_fast(b)) == b
