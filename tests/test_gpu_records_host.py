"""The device-side record functions (fastf_amd/csrc/gpu_records.hpp: record chain, aux scan, key packing, UMI codec, CRC-32
slicing) compiled for the host with one lane (tools/gr_host.cpp) and compared, on the CPU, with the host reader they
restate (host_io.c pack_record / rec_end / rec_plausible, host_prims.c fastf_keydict_pack / fastf_pack_umi) and with zlib."""
import ctypes as C
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import fastf_amd as F
from fastf_amd import synth, _lib
from helpers import Case
from test_bam_reader import read_all

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class View(C.Structure):
    _fields_ = [("n_prefix", C.c_uint32), ("prefix_len", C.c_uint32 * 8), ("prefix_id", C.c_uint64 * 8), ("prefix", (C.c_ubyte * 32) * 8)]


@pytest.fixture(scope="module")
def gr():
    so = os.path.join(ROOT, "build", "libgr_host.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fastf_amd", "csrc"), so])
    L = C.CDLL(so)
    L.gr_host_parse.restype = C.c_long
    L.gr_host_parse.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.gr_host_guess.restype = C.c_uint64
    L.gr_host_guess.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
    L.gr_host_crc.restype = C.c_uint32
    L.gr_host_crc.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32]
    L.gr_host_pack_key.restype = C.c_uint64
    L.gr_host_pack_key.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
    return L


def views(lists):
    L = _lib.lib()
    L.fastf_keydict_export.argtypes = [C.c_void_p, C.c_void_p]
    vc, vf = View(), View()
    assert L.fastf_keydict_export(lists.cell_dict, C.byref(vc)) == 0
    assert L.fastf_keydict_export(lists.feat_dict, C.byref(vf)) == 0
    return vc, vf


def inflated_records(path):
    """the BAM's inflated byte stream and the offset of its first record"""
    raw = gzip.decompress(open(path, "rb").read())
    l_text = struct.unpack_from("<i", raw, 4)[0]
    n_ref = struct.unpack_from("<i", raw, 8 + l_text)[0]
    o = 12 + l_text
    for _ in range(n_ref):
        o += 8 + struct.unpack_from("<i", raw, o)[0]
    return raw, o, n_ref


@pytest.mark.parametrize("str_type", [None, lambda i: b"H" if i % 3 else b"Z"], ids=["Z", "H_and_Z"])
@pytest.mark.parametrize("xf_type,shape", [(b"C", None), (b"i", lambda i: ((28 + 7 * (i % 19)) if i % 11 else 0, i % 4, (i % 3) - 1, 1000 + i)),
                                           (b"s", None)])
def test_chain_and_pack_equal_the_host_reader(gr, tmp_path, xf_type, shape, str_type):
    case = Case(n=20000, n_bar=300, n_gene=120, umi_pool=128, p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.2,
                p_n_umi=0.02, p_multi_gene=0.05, p_no_ub=0.03)
    lists = case.lists()
    bam = tmp_path / "t.bam"
    extra = lambda i: (b"NHC\x01" if i % 3 == 0 else b"") + (b"ZBBS\x02\x00\x00\x00\x01\x00\x02\x00" if i % 7 == 0 else b"") + (b"RGZgrp\x00" if i % 5 == 0 else b"")
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub, xf_type=xf_type, extra_aux=extra, refs=(("chr1", 100000), ("chr2", 100000)),
                    str_type=str_type, **({"shape": shape} if shape else {}))
    want = read_all(bam, lists, cap=7001)
    for w_, d_ in zip(want, case.packed(lists)):                    # (and the host reader itself against direct packing)
        np.testing.assert_array_equal(w_, d_)
    raw, start, n_ref = inflated_records(bam)
    vc, vf = views(lists)
    n = case.n
    cb = np.zeros(n + 5, np.uint64); gx = np.zeros(n + 5, np.uint64); um = np.zeros(n + 5, np.uint32); me = np.zeros(n + 5, np.uint32)
    nxf, ngx, hand = C.c_uint32(), C.c_uint32(), C.c_uint64()
    got = gr.gr_host_parse(raw, start, len(raw), C.byref(vc), C.byref(vf), cb.ctypes.data, gx.ctypes.data, um.ctypes.data, me.ctypes.data,
                           n + 5, C.byref(nxf), C.byref(ngx), C.byref(hand))
    assert got == n and hand.value == len(raw)
    for g, w in zip((cb, gx, um, me), want):
        np.testing.assert_array_equal(g[:n], w)
    # the guess of a chain start from the middle of a record lands on a true record start
    mid = start + (len(raw) - start) // 2
    o = gr.gr_host_guess(raw, mid, min(mid + 4096, len(raw)), len(raw), n_ref)
    assert o != 2**64 - 1
    k = gr.gr_host_parse(raw, o, len(raw), C.byref(vc), C.byref(vf), cb.ctypes.data, gx.ctypes.data, um.ctypes.data, me.ctypes.data,
                         n + 5, C.byref(nxf), C.byref(ngx), C.byref(hand))
    assert hand.value == len(raw) and 0 < k < n
    np.testing.assert_array_equal(cb[:k], want[0][n - k:])          # the same records as the tail of the true chain


def test_an_incomplete_last_record_is_handed_over(gr, tmp_path):
    case = Case(n=500, n_bar=20, n_gene=10)
    lists = case.lists()
    bam = tmp_path / "t.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub)
    raw, start, _ = inflated_records(bam)
    vc, vf = views(lists)
    a = [np.zeros(600, np.uint64), np.zeros(600, np.uint64), np.zeros(600, np.uint32), np.zeros(600, np.uint32)]
    nxf, ngx, hand = C.c_uint32(), C.c_uint32(), C.c_uint64()
    for cut in (len(raw) - 1, len(raw) - 40, start + 3, start + 200):
        n = gr.gr_host_parse(raw, start, cut, C.byref(vc), C.byref(vf), *[x.ctypes.data for x in a], 600, C.byref(nxf), C.byref(ngx), C.byref(hand))
        assert start <= hand.value <= cut and n < 500
        # the host takes over exactly there: the rest of the stream from the hand-over gives the remaining records
        m = gr.gr_host_parse(raw, hand.value, len(raw), C.byref(vc), C.byref(vf), *[x.ctypes.data for x in a], 600, C.byref(nxf), C.byref(ngx), C.byref(hand))
        assert n + m == 500


def test_key_packing_equals_the_host_dictionary(gr):
    bt = b"".join(b + b"\n" for b in [b"ACGTACGTACGTACGT-1", b"ACGTACGTACGTACGA-1", b"TTTTACGTACGTACGA"])
    ft = b"".join(b"%s\tn\tGene Expression\n" % g for g in [b"ENSG00000000012", b"ENSMUSG00000000003", b"ENSG00000001", b"Gm123"])
    lists = F.Lists(bt, ft)
    vc, vf = views(lists)
    L = _lib.lib()
    L.fastf_keydict_pack.restype = C.c_uint64
    L.fastf_keydict_pack.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    probes = [b"ACGTACGTACGTACGT-1", b"ACGTACGTACGTACGT-01", b"ACGTACGTACGTACGT-255", b"ACGTACGTACGTACGT-254", b"ACGTNCGTACGTACGT-1", b"acgt", b"",
              b"ACGTACGTACGTACGTACGTACGTA", b"ACGTACGTACGTACGTACGTACGT", b"NNNNNNNNNNNNNNNNN", b"ENSG00000000012", b"ENSG12", b"ENSG00000000000000012",
              b"ENSMUSG00000000003", b"ENSX00000000003", b"Gm123", b"Gm", b"123", b"ENSG00000000012;ENSG00000001", b"A-", b"A-1x", b"N-7", b"T" * 24 + b"-9"]
    for s in probes:
        for d, v in ((lists.cell_dict, vc), (lists.feat_dict, vf)):
            assert gr.gr_host_pack_key(C.byref(v), s, len(s)) == L.fastf_keydict_pack(d, s, len(s)), s


def test_escape_strings_keep_the_host_packer():
    bt = b"cell one\nACGT-1\n"                                   # a barcode that is neither DNA nor ID form: escape dictionary
    ft = b"G1\tn\tt\n"
    lists = F.Lists(bt, ft)
    L = _lib.lib()
    L.fastf_keydict_export.argtypes = [C.c_void_p, C.c_void_p]
    v = View()
    assert L.fastf_keydict_export(lists.cell_dict, C.byref(v)) == 1
    assert L.fastf_keydict_export(lists.feat_dict, C.byref(v)) == 0


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 65535, 65536])
def test_sliced_crc_equals_zlib(gr, n):
    rng = np.random.default_rng(n)
    data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    for slices in (1, 2, 64):
        assert gr.gr_host_crc(data, n, slices) == (zlib.crc32(data) & 0xFFFFFFFF)
