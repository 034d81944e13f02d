"""Device-side BAM front end, second stage (fastf_amd/csrc/gpu_records.hpp): the windows the device inflates are hopped and
packed there — CRC-32 per block, segment chains, one lane per record — and only the 24-byte SoA reaches the engine.  The
records must be exactly the host reader's, in file order, whatever the window size, the block structure, the record sizes
(records longer than a hop segment or than the carry-over reserve send a window back to the host parser) and the decoys."""
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

pytestmark = pytest.mark.gpu


def read_all_dev(path, lists, cap=1000, threads=0, gpu=2):
    """(cb, gx, umi, meta, records that came packed from the device)"""
    L = _lib.lib()
    hip = C.CDLL(_lib.hip_runtimes_loaded()[0])           # the ONE runtime of this process (by path: no second copy)
    L.fastf_bam_open2.restype = C.c_void_p
    L.fastf_bam_open2.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.fastf_bam_enable_device_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.fastf_bam_read_batch_dev.restype = C.c_long
    L.fastf_bam_read_batch_dev.argtypes = [C.c_void_p] * 7 + [C.c_size_t, C.POINTER(C.c_int), C.c_void_p]
    h = L.fastf_bam_open2(str(path).encode(), threads, gpu)
    assert h, L.fastf_last_error()
    assert L.fastf_bam_enable_device_parse(h, lists.cell_dict, lists.feat_dict) == 0
    outs = [[], [], [], []]
    n_dev = 0
    try:
        while True:
            a = [np.empty(cap, np.uint64), np.empty(cap, np.uint64), np.empty(cap, np.uint32), np.empty(cap, np.uint32)]
            on_dev = C.c_int(0)
            dev = _lib.Batch()
            n = L.fastf_bam_read_batch_dev(h, lists.cell_dict, lists.feat_dict, *[x.ctypes.data for x in a], cap, C.byref(on_dev), C.byref(dev))
            assert n >= 0, L.fastf_last_error()
            if n == 0:
                break
            if on_dev.value:
                assert dev.n == n
                for x, src in zip(a, (dev.cb_key, dev.gx_key, dev.umi, dev.meta)):
                    assert hip.hipMemcpy(C.c_void_p(x.ctypes.data), C.c_void_p(src), C.c_size_t(n * x.itemsize), 2) == 0
                n_dev += n
            for o, x in zip(outs, a):
                o.append(x[:n].copy())
    finally:
        L.fastf_bam_close(h)
    return [np.concatenate(o) if o else np.zeros(0) for o in outs] + [n_dev]


def check(path, lists, want, monkeypatch, window, cap=50_000, expect_dev=True, threads=0):
    monkeypatch.setenv("FASTF_BAM_WINDOW", str(window))
    if os.environ.get("FASTF_TEST_TRACE"):                           # per-window reader trace: thousands of lines at small windows
        monkeypatch.setenv("FASTF_BAM_PROFILE", "2")
    else:
        monkeypatch.delenv("FASTF_BAM_PROFILE", raising=False)
    *got, n_dev = read_all_dev(path, lists, cap=cap, threads=threads)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    if expect_dev:
        assert n_dev > 0.5 * len(want[0]), n_dev                       # the records really came from the device
    return n_dev


@pytest.mark.parametrize("window", [1 << 17, 1 << 20, 1 << 22, 128 << 20])
@pytest.mark.parametrize("xf_type", [b"C", b"i", b"H"])          # b"H": xf 'C', every third record's strings of type 'Z', the others 'H'
def test_device_parse_gives_the_host_readers_records(tmp_path, monkeypatch, window, xf_type):
    case = Case(n=150_000, n_bar=300, n_gene=120, umi_pool=128, p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.2, p_n_umi=0.02,
                p_multi_gene=0.05, p_no_ub=0.03)
    lists = case.lists()
    bam = tmp_path / "t.bam"
    shape = lambda i: ((28 + 7 * (i % 19)) if i % 11 else 0, i % 4, (i % 2) - 1, 1000 + i)
    extra = lambda i: (b"NHC\x01" if i % 3 == 0 else b"") + (b"ZBBS\x02\x00\x00\x00\x01\x00\x02\x00" if i % 7 == 0 else b"") + (b"RGZgrp\x00" if i % 5 == 0 else b"")
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub, xf_type=b"C" if xf_type == b"H" else xf_type, shape=shape,
                    extra_aux=extra, str_type=(lambda i: b"H" if i % 3 else b"Z") if xf_type == b"H" else None)
    # (the device parse is switched on after the reader is open: with windows as large as the file everything has been
    # prefetched for the host parser by then — still the same records)
    check(bam, lists, case.packed(lists), monkeypatch, window, cap=50_000 if window > (1 << 17) else 7001, expect_dev=window <= (1 << 20))


def test_decoy_records_do_not_fool_the_segment_chains(tmp_path, monkeypatch):
    """most of the file is decoy (byte arrays holding runs of perfectly plausible records): a segment whose guess lands in
    a decoy does not line up with its neighbour; gr_repair_kernel walks such segments again from where the true chain enters
    them and the window stays on the device (round 4: it went back to the host parser) — never a wrong record"""
    case = Case(n=6000, n_bar=50, n_gene=20, umi_pool=32, p_no_cb=0.1, p_bad_xf=0.1)
    lists = case.lists()
    decoy_rec = synth.bam_record(b"decoy", synth.aux_Z(b"CB", b"AAAAAAAAAAAAAAAA-1") + synth.aux_int(b"xf", 25) +
                                 synth.aux_Z(b"GX", b"ENSG00000000001") + synth.aux_Z(b"UB", b"ACGTACGTAC"), seq_len=20, n_cigar=1,
                                 ref_id=0, pos=5)

    def extra(i):
        payload = decoy_rec * (3 + i % 9) + bytes([i % 251] * (i % 7))
        return b"ZBBC" + struct.pack("<i", len(payload)) + payload
    bam = tmp_path / "t.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub, extra_aux=extra)
    for window in (1 << 17, 32 << 20):
        n_dev = check(bam, lists, case.packed(lists), monkeypatch, window, cap=2500, expect_dev=False)
        if window == (1 << 17):
            assert n_dev > 1000, n_dev            # windows full of decoys were packed on the device (round 4: none of them)


@pytest.mark.parametrize("window", [1 << 17, 1 << 22])
def test_odd_blocks_and_giant_records_with_device_parse(tmp_path, monkeypatch, window):
    """blocks of 1 byte to 65 280 bytes, stored blocks, empty blocks mid-file, a 300 KB record (longer than a hop segment)
    and a 2.4 MB record (longer than the carry-over reserve): inside a record longer than a hop segment no chain start can be
    guessed — gr_repair_kernel walks through it from the chain before; what the window cannot hold whole is the host's"""
    case = Case(n=4000, n_bar=40, n_gene=25, umi_pool=32, p_no_cb=0.05, p_bad_xf=0.1)
    lists = case.lists()

    def extra(i):
        if i == 1500:
            return b"ZBBC" + struct.pack("<i", 300_000) + bytes(300_000)
        if i == 3000:
            return b"ZCBs" + struct.pack("<i", 1_200_000) + bytes(2_400_000)
        return b""
    plain = tmp_path / "plain.bam"
    synth.write_bam(str(plain), case.flags, case.xf, case.cb, case.gx, case.ub, extra_aux=extra)
    stream = gzip.decompress(plain.read_bytes())
    rng = np.random.default_rng(8)
    out, pos, k = bytearray(), 0, 0
    while pos < len(stream):
        n = int(rng.choice([1, 7, 300, 5000, 65280])) if k % 3 else int(rng.integers(1, 65281))
        chunk = stream[pos:pos + n]
        if k % 5 == 0:
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
            out += synth._BGZF_EOF
        pos += len(chunk); k += 1
    out += synth._BGZF_EOF
    odd = tmp_path / "odd.bam"
    odd.write_bytes(bytes(out))
    for path in (plain, odd):
        check(path, lists, case.packed(lists), monkeypatch, window, cap=1234, expect_dev=False)


def test_a_block_with_a_wrong_crc_is_an_error_with_device_parse_too(tmp_path, monkeypatch):
    case = Case(n=30_000, n_bar=100, n_gene=50)
    lists = case.lists()
    bam = tmp_path / "t.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub)
    raw = bytearray(bam.read_bytes())
    # flip a bit in the CRC field of the third block: the payload inflates, the check must fail (device CRC, then zlib agrees)
    pos = 0
    for _ in range(2):
        pos += struct.unpack_from("<H", raw, pos + 16)[0] + 1
    bsize = struct.unpack_from("<H", raw, pos + 16)[0] + 1
    raw[pos + bsize - 8] ^= 0x01
    bad = tmp_path / "bad.bam"
    bad.write_bytes(bytes(raw))
    monkeypatch.setenv("FASTF_BAM_WINDOW", str(1 << 20))
    L = _lib.lib()
    with pytest.raises(AssertionError):
        read_all_dev(bad, lists, cap=10_000)
    assert b"CRC" in L.fastf_last_error() or b"inflate" in L.fastf_last_error()


@pytest.mark.parametrize("umicopies", [False, True])
def test_cli_with_device_parse_equals_oracle_bytes(tmp_path, umicopies):
    case = Case(n=200_000, n_bar=800, n_gene=300, rate_cell=0.5, rate_depth=0.5, umi_len=12, dup_factor=3.0,
                p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.15, p_n_umi=0.005)
    bam = tmp_path / "in.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub, shape=lambda i: (91, 1, 0, i * 37 % 900))
    b, f = tmp_path / "b.tsv", tmp_path / "f.tsv"
    b.write_bytes(case.bt); f.write_bytes(case.ft)
    out = tmp_path / "out"; out.mkdir()
    case.label = str(bam).encode()
    ora = case.oracle()
    env = dict(os.environ, FASTF_GPU_INFLATE="2", FASTF_BAM_WINDOW=str(1 << 21), FASTF_BATCH_RECORDS="30000", FASTF_BAM_PROFILE="1")
    r = subprocess.run([_lib.cli_path(), "bam2db", "-b", str(bam), "-a", str(b), "-f", str(f), "-o", str(out),
                        "-c", "0.5", "-r", "0.5"] + (["-u"] if umicopies else []), capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "records hopped and packed there" in r.stderr, r.stderr        # the device parse really ran
    rd = lambda n: gzip.decompress((out / n).read_bytes())
    assert rd("matrix.mtx.gz") == ora["matrix"]
    if umicopies:
        assert rd("umi.tsv.gz") == ora["umi"]


def test_cli_with_two_windows_in_flight_equals_oracle_bytes(tmp_path):
    """FASTF_BAM_EARLY=1 (host_io.c: early_queue_next): the next window's device share is queued on a second context while this
    window's is still running.  Same bytes as the oracle, and the trace says windows really were queued a window ahead."""
    case = Case(n=1_200_000, n_bar=800, n_gene=300, rate_cell=0.5, rate_depth=0.5, umi_len=12, dup_factor=3.0,
                p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.15, p_n_umi=0.005)
    bam = tmp_path / "in.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub, shape=lambda i: (91, 1, 0, i * 37 % 900))
    b, f = tmp_path / "b.tsv", tmp_path / "f.tsv"
    b.write_bytes(case.bt); f.write_bytes(case.ft)
    out = tmp_path / "out"; out.mkdir()
    case.label = str(bam).encode()
    ora = case.oracle()
    env = dict(os.environ, FASTF_GPU_INFLATE="2", FASTF_BAM_EARLY="1", FASTF_BAM_WINDOW=str(2 << 20), FASTF_BATCH_RECORDS="50000",
               FASTF_BAM_PROFILE="2")
    r = subprocess.run([_lib.cli_path(), "bam2db", "-b", str(bam), "-a", str(b), "-f", str(f), "-o", str(out),
                        "-c", "0.5", "-r", "0.5", "-u"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "queued a window ago" in r.stderr, r.stderr
    rd = lambda n: gzip.decompress((out / n).read_bytes())
    assert rd("matrix.mtx.gz") == ora["matrix"]
    assert rd("umi.tsv.gz") == ora["umi"]
