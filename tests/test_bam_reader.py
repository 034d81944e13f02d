"""BGZF/BAM front end of the product (host C, no GPU): records decoded from a BAM file give
the same packed SoA as packing the generator's strings directly."""
import ctypes as C
import struct

import numpy as np
import pytest

import fastf_amd as F
from fastf_amd import synth, _lib
from helpers import Case


def read_all(path, lists, cap=1000, threads=0):
    L = _lib.lib()
    h = L.fastf_bam_open(str(path).encode(), threads)
    assert h, L.fastf_last_error()
    outs = [[], [], [], []]
    try:
        while True:
            cb = np.empty(cap, np.uint64); gx = np.empty(cap, np.uint64)
            um = np.empty(cap, np.uint32); me = np.empty(cap, np.uint32)
            n = L.fastf_bam_read_batch(h, lists.cell_dict, lists.feat_dict, cb.ctypes.data, gx.ctypes.data,
                                       um.ctypes.data, me.ctypes.data, cap)
            assert n >= 0, L.fastf_last_error()
            if n == 0:
                break
            for o, a in zip(outs, (cb, gx, um, me)):
                o.append(a[:n].copy())
    finally:
        L.fastf_bam_close(h)
    return [np.concatenate(o) if o else np.zeros(0) for o in outs]


@pytest.mark.parametrize("xf_type", [b"C", b"c", b"i", b"S"])
def test_bam_roundtrip_equals_direct_packing(tmp_path, xf_type):
    case = Case(n=30000, n_bar=300, n_gene=120, umi_pool=128, p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.2,
                p_n_umi=0.02, p_multi_gene=0.05, p_no_ub=0.03)
    lists = case.lists()
    bam = tmp_path / "t.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub, xf_type=xf_type)
    got = read_all(bam, lists, cap=7001)
    want = case.packed(lists)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)


def test_bam_mapped_read_layout(tmp_path):
    """records shaped like Cell Ranger's: CIGAR words, packed bases and qualities of varying length before the tags,
    mapped and unmapped reads mixed, names of varying length"""
    case = Case(n=20000, n_bar=100, n_gene=50, umi_pool=64, p_no_cb=0.05, p_bad_xf=0.1)
    lists = case.lists()
    bam = tmp_path / "t.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub,
                    shape=lambda i: ((28 + 7 * (i % 19)) if i % 11 else 0, i % 4, (i % 3) - 1, 1000 + i))
    got = read_all(bam, lists, cap=4099)
    for g, w in zip(got, case.packed(lists)):
        np.testing.assert_array_equal(g, w)


def test_mapped_and_buffered_reads_agree(tmp_path, monkeypatch):
    """the reader maps regular files and falls back to read() (FASTF_BAM_MMAP=0): same records either way"""
    case = Case(n=40000, n_bar=200, n_gene=80, umi_pool=64, p_no_cb=0.05)
    lists = case.lists()
    bam = tmp_path / "t.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub)
    monkeypatch.setenv("FASTF_BAM_WINDOW", str(1 << 17))
    a = read_all(bam, lists, cap=9001)
    monkeypatch.setenv("FASTF_BAM_MMAP", "0")
    b = read_all(bam, lists, cap=9001)
    for x, y, w in zip(a, b, case.packed(lists)):
        np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(x, w)


def test_bam_other_aux_types_and_tag_order(tmp_path):
    """tags of every aux type before/between the ones we need; first occurrence wins (bam_aux_get)"""
    case = Case(n=500, n_bar=20, n_gene=10, umi_pool=8)
    lists = case.lists()

    def extra(i):
        a = b"NHC" + bytes([i % 250]) + b"ASi" + struct.pack("<i", -i) + b"XYf" + struct.pack("<f", 1.5)
        a += b"ZZZhello\0" + b"HHH1AE3\0" + b"BBBs" + struct.pack("<i", 3) + struct.pack("<hhh", 1, 2, 3)
        a += b"BCBC" + struct.pack("<i", 2) + b"\1\2" + b"ddd" + struct.pack("<d", 2.5) + b"AAAx"
        return a
    bam = tmp_path / "t.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub, extra_aux=extra)
    got = read_all(bam, lists)
    for g, w in zip(got, case.packed(lists)):
        np.testing.assert_array_equal(g, w)


def test_bam_non_integer_xf_and_non_string_cb(tmp_path):
    lists = F.Lists(b"AAAA-1\n", b"G1\tN\tT\n")
    recs = [
        synth.bam_record(b"a", synth.aux_Z(b"CB", b"AAAA-1") + b"xfA" + b"\x19" + synth.aux_Z(b"GX", b"G1") + synth.aux_Z(b"UB", b"ACGT")),
        synth.bam_record(b"b", synth.aux_int(b"CB", 5) + synth.aux_int(b"xf", 25) + synth.aux_Z(b"GX", b"G1") + synth.aux_Z(b"UB", b"ACGT")),
        synth.bam_record(b"c", synth.aux_Z(b"CB", b"AAAA-1") + synth.aux_int(b"xf", 17, b"s") + synth.aux_Z(b"GX", b"G1") + synth.aux_Z(b"UB", b"ACGT")),
    ]
    payload = b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", 0) + b"".join(recs)
    p = tmp_path / "t.bam"
    p.write_bytes(synth._bgzf_block(payload) + synth._BGZF_EOF)
    cb, gx, um, me = read_all(p, lists)
    assert cb[0] != 0 and not (me[0] & 1)      # xf of type 'A' → bam_aux2i gives 0 → not 25/17
    assert cb[1] == 0                          # CB not a string → lookup misses
    assert cb[2] != 0 and (me[2] & 1)


def test_hex_typed_strings_read_like_Z(tmp_path):
    """htslib's bam_aux2Z hands back the bytes of an 'H' value exactly as of a 'Z' value (sam.c: `type == 'Z' || type ==
    'H'`), so the reference looks such a CB / GX up and encodes such a UB (bam2db_ds.c:366,403,412): same packed records"""
    case = Case(n=20000, n_bar=300, n_gene=120, umi_pool=128, p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.2,
                p_n_umi=0.02, p_multi_gene=0.05, p_no_ub=0.03)
    lists = case.lists()
    bam = tmp_path / "t.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub, str_type=lambda i: b"H" if i % 3 else b"Z")
    assert b"CBH" in __import__("gzip").decompress(bam.read_bytes())
    got = read_all(bam, lists, cap=7001)
    for g, w in zip(got, case.packed(lists)):
        np.testing.assert_array_equal(g, w)


def test_bam_truncated_and_garbage(tmp_path):
    L = _lib.lib()
    p = tmp_path / "g.bam"
    p.write_bytes(b"this is not a bam file at all")
    assert not L.fastf_bam_open(str(p).encode(), 0)
    assert not L.fastf_bam_open(str(tmp_path / "missing.bam").encode(), 0)
    case = Case(n=2000, n_bar=20, n_gene=10)
    lists = case.lists()
    full = tmp_path / "full.bam"
    synth.write_bam(str(full), case.flags, case.xf, case.cb, case.gx, case.ub)
    data = full.read_bytes()
    cut = tmp_path / "cut.bam"
    cut.write_bytes(data[: len(data) // 2])
    h = L.fastf_bam_open(str(cut).encode(), 0)
    assert h
    cb = np.empty(5000, np.uint64); gx = np.empty(5000, np.uint64); um = np.empty(5000, np.uint32); me = np.empty(5000, np.uint32)
    n = L.fastf_bam_read_batch(h, lists.cell_dict, lists.feat_dict, cb.ctypes.data, gx.ctypes.data, um.ctypes.data, me.ctypes.data, 5000)
    L.fastf_bam_close(h)
    assert n < 2000      # stops at the damage instead of inventing records (error or short read)


def test_empty_bam(tmp_path):
    lists = F.Lists(b"AAAA-1\n", b"G1\tN\tT\n")
    p = tmp_path / "e.bam"
    synth.write_bam(str(p), np.zeros(0, np.uint8), np.zeros(0, np.int32), np.zeros(0, "S4"), np.zeros(0, "S4"), np.zeros(0, "S4"))
    got = read_all(p, lists)
    assert len(got[0]) == 0


@pytest.mark.parametrize("window,threads", [(1 << 17, 1), (1 << 17, 3), (200_000, 8), (1 << 25, 5)])
def test_reader_windows_and_threads_are_invisible(tmp_path, monkeypatch, window, threads):
    """tiny read windows force records and BGZF blocks to straddle refills; any thread count gives
    the same SoA in the same order"""
    case = Case(n=60000, n_bar=300, n_gene=120, umi_pool=128, p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.2,
                p_n_umi=0.02, p_multi_gene=0.05, p_no_ub=0.03)
    lists = case.lists()
    bam = tmp_path / "t.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub, header_text=b"@HD\tVN:1.6\n" + b"@CO\tx\n" * 40000)
    monkeypatch.setenv("FASTF_BAM_WINDOW", str(window))
    monkeypatch.setenv("FASTF_HOST_THREADS", str(threads))
    got = read_all(bam, lists, cap=9973)
    for g, w in zip(got, case.packed(lists)):
        np.testing.assert_array_equal(g, w)


def test_reader_rejects_corrupt_block(tmp_path, monkeypatch):
    """a byte flipped late in a multi-window file: the open succeeds, a later refill fails on the prefetch thread,
    and the calling thread still reads the cause from fastf_last_error()"""
    case = Case(n=60000, n_bar=20, n_gene=10)
    lists = case.lists()
    p = tmp_path / "c.bam"
    synth.write_bam(str(p), case.flags, case.xf, case.cb, case.gx, case.ub)
    data = bytearray(p.read_bytes())
    assert len(data) > 4 * (1 << 17)
    data[len(data) * 3 // 4] ^= 0xFF                  # inside a deflate stream of a late block
    p.write_bytes(bytes(data))
    monkeypatch.setenv("FASTF_BAM_WINDOW", str(1 << 17))
    L = _lib.lib()
    L.fastf_set_error_.argtypes = [C.c_char_p]
    L.fastf_set_error_(b"stale text from an earlier call")
    h = L.fastf_bam_open(str(p).encode(), 2)
    assert h, L.fastf_last_error()
    cb = np.empty(9000, np.uint64); gx = np.empty(9000, np.uint64); um = np.empty(9000, np.uint32); me = np.empty(9000, np.uint32)
    n, total = 1, 0
    while n > 0:
        n = L.fastf_bam_read_batch(h, lists.cell_dict, lists.feat_dict, cb.ctypes.data, gx.ctypes.data, um.ctypes.data, me.ctypes.data, 9000)
        total += max(n, 0)
    L.fastf_bam_close(h)
    assert n == -1 and 0 < total < case.n
    assert b"BGZF" in L.fastf_last_error() or b"corrupt BAM record" in L.fastf_last_error()


def test_reader_rejects_oversized_isize(tmp_path):
    """BGZF caps a block's inflated size at 64 KiB: a crafted ISIZE near 4 GiB must be refused, not allocated"""
    case = Case(n=2000, n_bar=20, n_gene=10)
    p = tmp_path / "i.bam"
    synth.write_bam(str(p), case.flags, case.xf, case.cb, case.gx, case.ub)
    data = bytearray(p.read_bytes())
    bsize = struct.unpack_from("<H", data, 16)[0] + 1          # first block: ISIZE is its last 4 bytes
    struct.pack_into("<I", data, bsize - 4, 0xFFFFFF00)
    p.write_bytes(bytes(data))
    L = _lib.lib()
    h = L.fastf_bam_open(str(p).encode(), 2)
    assert not h and b"BGZF" in L.fastf_last_error()


@pytest.mark.parametrize("window,threads", [(1 << 17, 8), (1 << 20, 4), (32 << 20, 8)])
def test_parallel_record_hop_survives_decoy_records(tmp_path, monkeypatch, window, threads):
    """The parallel hop guesses a record start per segment and verifies it against the true chain.  Here most of the file
    is decoy: every record carries a byte array whose content is itself a run of perfectly plausible BAM records, so
    the guesses usually land in decoys; the result must still be the serial hop's."""
    case = Case(n=6000, n_bar=50, n_gene=20, umi_pool=32, p_no_cb=0.1, p_bad_xf=0.1)
    lists = case.lists()
    decoy_rec = synth.bam_record(b"decoy", synth.aux_Z(b"CB", b"AAAAAAAAAAAAAAAA-1") + synth.aux_int(b"xf", 25) +
                                 synth.aux_Z(b"GX", b"ENSG00000000001") + synth.aux_Z(b"UB", b"ACGTACGTAC"), seq_len=20, n_cigar=1,
                                 ref_id=0, pos=5)

    def extra(i):
        payload = decoy_rec * (3 + i % 9) + bytes([i % 251] * (i % 7))           # misaligned runs of whole decoy records
        return b"ZBBC" + struct.pack("<i", len(payload)) + payload
    bam = tmp_path / "t.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub, extra_aux=extra)
    monkeypatch.setenv("FASTF_BAM_WINDOW", str(window))
    want = case.packed(lists)
    got = read_all(bam, lists, cap=2500, threads=threads)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    monkeypatch.setenv("FASTF_BAM_SERIAL_HOP", "1")
    got = read_all(bam, lists, cap=2500, threads=threads)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)


@pytest.mark.parametrize("window,threads", [(1 << 17, 4), (1 << 22, 8)])
def test_odd_block_structure_and_giant_records(tmp_path, monkeypatch, window, threads):
    """The same BAM stream re-blocked the hard way: blocks of 1 byte to 65 280 bytes, stored (level 0) blocks, empty
    blocks (the EOF marker) in the middle of the file, and two records far larger than a BGZF block — one of them larger
    than the space the reader reserves in front of a window for a record that straddles two windows."""
    import gzip, zlib
    case = Case(n=4000, n_bar=40, n_gene=25, umi_pool=32, p_no_cb=0.05, p_bad_xf=0.1)
    lists = case.lists()

    def extra(i):
        if i == 1500:
            return b"ZBBC" + struct.pack("<i", 300_000) + bytes(300_000)          # a record of 300 KB: five blocks
        if i == 3000:
            return b"ZCBs" + struct.pack("<i", 1_200_000) + bytes(2_400_000)      # 2.4 MB: longer than the carry-over reserve
        return b""
    plain = tmp_path / "plain.bam"
    synth.write_bam(str(plain), case.flags, case.xf, case.cb, case.gx, case.ub, extra_aux=extra)
    stream = gzip.decompress(plain.read_bytes())
    rng = np.random.default_rng(8)
    out, pos, k = bytearray(), 0, 0
    while pos < len(stream):
        n = int(rng.choice([1, 7, 300, 5000, 65280])) if k % 3 else int(rng.integers(1, 65281))
        chunk = stream[pos:pos + n]
        if k % 5 == 0:                                                          # stored block
            co = zlib.compressobj(0, zlib.DEFLATED, -15)
            data = co.compress(chunk) + co.flush()
            if len(data) + 25 < 65536:
                hdr = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord("B"), ord("C"), 2, len(data) + 25)
                out += hdr + data + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
            else:
                out += synth._bgzf_block(chunk)
        else:
            out += synth._bgzf_block(chunk)
        if k % 11 == 0:
            out += synth._BGZF_EOF                                              # an empty block in the middle
        pos += len(chunk); k += 1
    out += synth._BGZF_EOF
    odd = tmp_path / "odd.bam"
    odd.write_bytes(bytes(out))
    monkeypatch.setenv("FASTF_BAM_WINDOW", str(window))
    want = case.packed(lists)
    for path in (plain, odd):
        got = read_all(path, lists, cap=1234, threads=threads)
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)
