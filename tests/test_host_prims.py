"""Host-side C of the product (libfastf_amd.so, no GPU needed) against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import fastf_amd as F
from fastf_amd import engine, synth, _lib
from oracle import oracle as O
from helpers import Case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,skip", [(926, 0), (926, 500), (1, 623), (1, 624), (77, 100000)])
def test_mt_stream(seed, skip):
    want = O.mt_stream(seed, skip + 5000)[skip:]
    np.testing.assert_array_equal(F.mt_draws(seed, skip, 5000), want)


@pytest.mark.parametrize("rate", [0.0, 1e-9, 0.01, 0.1, 0.25, 0.3, 0.5, 0.7, 0.999, 1.0, 1.5, -0.5, float("nan")])
def test_draw_threshold_is_the_reference_compare(rate):
    T = F.draw_threshold(rate)
    L = O.lib()
    probes = {0, 1, 0xFFFFFFFF, 0xFFFFFFFE, 0x7FFFFFFF, 0x80000000}
    for d in (T - 2, T - 1, T, T + 1):
        if 0 <= d <= 0xFFFFFFFF:
            probes.add(d)
    rng = np.random.default_rng(1)
    probes.update(int(x) for x in rng.integers(0, 1 << 32, size=2000))
    for d in probes:
        assert (d < T) == bool(L.oracle_keep_draw(d, C.c_float(rate))), (rate, d, T)


@pytest.mark.parametrize("n,rate,seed", [(6, 0.5, 926), (1000, 0.5, 926), (1000, 1.0, 3), (1000, 0.0, 3), (999, 0.333, 5), (50000, 0.25, 11), (3, 0.34, 1)])
def test_sample_cells(n, rate, seed):
    lines, used = engine.sample_cells(n, rate, seed)
    k = O.lib().oracle_n_cells_sampled(n, C.c_float(rate))
    np.testing.assert_array_equal(lines, O.sample_cells(n, k, seed))
    assert used == (0 if k == n else k)


def test_sample_cells_rejects_what_the_reference_exits_on():
    with pytest.raises(F.FastfError):
        engine.sample_cells(10, 1.5, 1)
    with pytest.raises(F.FastfError):
        engine.sample_cells(10, -0.1, 1)


@pytest.mark.parametrize("umi", [b"", b"A", b"ACGT", b"ACGTA", b"ACGTACGTAC", b"TTTTTTTTTTTT", b"ACGTACGTACGTACGT", b"ACGTNACGTA", b"acgt", b"ACGTACGTACGTACGTA"])
def test_pack_umi(umi):
    out = C.c_uint32(0)
    meta = _lib.lib().fastf_pack_umi(umi, len(umi), C.byref(out))
    assert meta & 2
    if len(umi) > 16:
        assert meta & 8
        return
    blob = O.encode_dna(umi)
    if blob is None:
        assert not (meta & 4) and out.value == 0
    else:
        assert meta & 4
        assert (meta >> 4) & 7 == len(blob)
        assert out.value.to_bytes(4, "big")[:len(blob)] == blob
        assert out.value.to_bytes(4, "big")[len(blob):] == b"\0" * (4 - len(blob))


def _rand_strings(rng, n):
    alph = [b"ACGT", b"ACGTN", b"0123456789", b"ENSGMUS_.-;abcxyz0123456789", bytes(range(33, 127))]
    out = []
    for _ in range(n):
        a = alph[rng.integers(0, len(alph))]
        ln = int(rng.integers(0, 30))
        s = bytes(a[i] for i in rng.integers(0, len(a), size=ln))
        if rng.random() < 0.3:
            s += b"-" + str(int(rng.integers(0, 400))).encode()
        if rng.random() < 0.3:
            s = b"ENSG" + s + str(int(rng.integers(0, 10 ** int(rng.integers(1, 16))))).zfill(int(rng.integers(1, 16))).encode()
        out.append(s)
    return out


def test_keydict_is_exact():
    """pack(a) == pack(b)  <=>  a == b, for registered a and arbitrary b"""
    L = _lib.lib()
    rng = np.random.default_rng(7)
    reg = list(dict.fromkeys(_rand_strings(rng, 3000) + [b"", b"A", b"ACGT-1", b"ACGT-01", b"ACGT-255", b"ACGT-254",
                                                             b"G1", b"G01", b"G1;G2", b"ENSG00000000003", b"ENSG00000000003.14",
                                                             b"A" * 24, b"A" * 25, b"A" * 24 + b"-1", b"123", b"0123", b"9" * 13, b"9" * 14]))
    d = L.fastf_keydict_create()
    keys = {}
    for s in reg:
        k = L.fastf_keydict_add(d, s, len(s))
        assert k != 0
        keys[s] = k
    assert len(set(keys.values())) == len(reg)                       # injective on the registered set
    for s in reg:                                                    # stable
        assert L.fastf_keydict_pack(d, s, len(s)) == keys[s]
        assert L.fastf_keydict_add(d, s, len(s)) == keys[s]
    regset = set(reg)
    inv = {v: k for k, v in keys.items()}
    probes = _rand_strings(rng, 6000) + [s + b"x" for s in reg[:500]] + [s[:-1] for s in reg[:500] if s]
    for s in probes:
        k = L.fastf_keydict_pack(d, s, len(s))
        if s in regset:
            assert k == keys[s]
        else:
            assert k == 0 or k not in inv, (s, inv.get(k))
    L.fastf_keydict_destroy(d)


def test_keydict_many_prefixes_fall_back_to_escape():
    L = _lib.lib()
    d = L.fastf_keydict_create()
    reg = [b"P%dX_7" % i for i in range(20000)]                     # 20 000 distinct prefixes > 2^14
    keys = [L.fastf_keydict_add(d, s, len(s)) for s in reg]
    assert len(set(keys)) == len(reg) and 0 not in keys
    for s, k in zip(reg, keys):
        assert L.fastf_keydict_pack(d, s, len(s)) == k
    assert L.fastf_keydict_pack(d, b"P99999X_7", 9) == 0
    L.fastf_keydict_destroy(d)


@pytest.mark.parametrize("kw", [
    dict(n=10, n_bar=1000, n_gene=500),
    dict(n=10, n_bar=1000, n_gene=500, rate_cell=0.5),
    dict(n=10, n_bar=7, n_gene=3, rate_cell=0.3, seed=5),
])
def test_lists_match_oracle(kw):
    case = Case(**kw)
    ora = case.oracle()
    lists = case.lists()
    assert lists.barcodes_text() == ora["barcodes"]
    assert lists.features_text() == ora["features"]
    assert (lists.n_features, lists.n_cells) == (ora["n_feature"], ora["n_barcode"])
    assert lists.mt_skip == (0 if lists.n_sampled_target == lists.n_lines_barcodes else lists.n_sampled_target)


def test_lists_quirks_match_oracle():
    """duplicate barcode stalls insertion; duplicate/odd feature lines; CRLF; gz-style long lines"""
    bar = b"AAAA-1\nCCCC-1\r\nAAAA-1\nGGGG-1\nTTTT-1\tx\n"
    feat = b"G1\tN1\tGene Expression\nG2\tN2\tGene Expression\r\nG1\tdup\tGene Expression\n\tG9\tN9\tT9\nG3\t\tN3\tAntibody Capture\textra\n"
    n = 4
    flags = np.full(n, 15, np.uint8); xf = np.full(n, 25, np.int32)
    cb = np.array([b"AAAA-1", b"CCCC-1", b"GGGG-1", b"TTTT-1"], dtype="S8")
    gx = np.array([b"G1", b"G2", b"\tG9", b"G3"], dtype="S8")
    ub = np.array([b"ACGTACGTAC"] * n, dtype="S16")
    ora = O.run_bam2db(bar, feat, flags, xf, cb, gx, ub, 1.0, 1.0, 926)
    lists = F.Lists(bar, feat, 1.0, 926)
    assert lists.barcodes_text() == ora["barcodes"] == b"AAAA-1\nCCCC-1\n"
    assert lists.features_text() == ora["features"]
    assert lists.dup_barcodes == 1 and lists.dup_features == 1
    # packed keys reproduce the oracle's hit pattern: rows for AAAA-1/G1 and CCCC-1/G2 only
    k = F.pack_records(lists, flags, xf, cb, gx, ub)
    cell_hit = np.isin(k[0], lists.cell_keys) & (k[0] != 0)
    feat_hit = np.isin(k[1], lists.feature_keys) & (k[1] != 0)
    assert cell_hit.tolist() == [True, True, False, False]
    assert feat_hit.tolist() == [True, True, True, True]
    assert ora["valid"] == 2


def test_lists_bad_feature_line_is_an_error():
    with pytest.raises(F.FastfError):
        F.Lists(b"AAAA-1\n", b"G1\tonly two columns\n", 1.0, 926)


def test_format_matrix_matches_oracle_text():
    case = Case(n=5000, n_bar=40, n_gene=20, umi_pool=16, rate_depth=0.5)
    ora = case.oracle()
    lists = case.lists()
    res = dict(feature=ora["feature"].astype(np.uint32), cell=ora["cell"].astype(np.uint32), count=ora["count"].astype(np.uint32),
               total=ora["total"], sampled=ora["sampled"], valid=ora["valid"], nnz=ora["nnz"])
    txt = engine.Engine.format_matrix(_FakeEngine(), res, case.label, case.rate_cell, case.rate_depth, lists.n_features, lists.n_cells)
    assert txt == ora["matrix"]


class _FakeEngine:
    """format_matrix only needs the library handle (pure host formatting, no device)"""
    _L = _lib.lib()


def test_pack_umi_long_matches_the_codec_up_to_32_bases():
    """fastf_pack_umi_long: bases 1..16 in the first word, 17..32 left-aligned in the second — together the zero-padded
    2-bit blob of encode_DNA (bam2db_ds.c:22-51) for lengths up to 32; N anywhere makes it NULL; 33 bases: too long"""
    import ctypes as C
    from fastf_amd import _lib
    L = _lib.lib()
    L.fastf_pack_umi_long.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.fastf_pack_umi_long.restype = C.c_uint32
    rng = np.random.default_rng(8)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    for ln in list(range(0, 33)) * 4:
        sq = "".join("ACGT"[i] for i in rng.integers(0, 4, size=ln))
        u, e = C.c_uint32(), C.c_uint32()
        m = L.fastf_pack_umi_long(sq.encode(), ln, C.byref(u), C.byref(e))
        want = 0
        for ch in sq:
            want = (want << 2) | code[ch]
        want <<= 64 - 2 * ln if ln else 0
        assert (u.value << 32 | e.value) == want and (m & 2) and (m & 4) and not (m & 8)
        assert (m >> 4) & 15 == (ln + 3) // 4
        if ln <= 16:                                          # the short packer agrees on what it can hold
            u2 = C.c_uint32()
            assert L.fastf_pack_umi(sq.encode(), ln, C.byref(u2)) == m and u2.value == u.value and e.value == 0
    u, e = C.c_uint32(), C.c_uint32()
    assert L.fastf_pack_umi_long(b"ACGTNACGTACGTACGTACG", 20, C.byref(u), C.byref(e)) & 4 == 0 and (u.value, e.value) == (0, 0)
    assert L.fastf_pack_umi_long(b"A" * 33, 33, C.byref(u), C.byref(e)) & 8


def test_mt_jump_polynomials_are_the_generator_advanced():
    """mt_jump.c: x^J mod the characteristic polynomial of MT19937 (Berlekamp-Massey on its own output, squarings, products mod
    phi), applied to a block-boundary array as a convolution with the generated sequence, must give the array the generator reaches
    by generating J draws — for the table the library carries (the two-level seating: x^(i J) and x^(i R J), J = 624 x 256 draws,
    R = 32) and for polynomials computed here.  The reference is the ORACLE's generator (oracle.mt_stream: mt19937ar.c restated),
    continued draw by draw."""
    import ctypes as C
    L = _lib.lib()
    NW, J, R = 312, 624 * 256, 32
    L.fastf_mt_jump_table.restype = C.POINTER(C.c_uint64)
    L.fastf_mt_jump_polys.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
    L.fastf_mt_jump_table_compute.argtypes = [C.c_uint64, C.c_void_p]
    L.fastf_mt_jump_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    table = np.ctypeslib.as_array(L.fastf_mt_jump_table(), shape=(2 * (R - 1) * NW,)).copy()
    fresh = np.zeros(2 * (R - 1) * NW, np.uint64)
    assert L.fastf_mt_jump_table_compute(J, fresh.ctypes.data) == 0
    np.testing.assert_array_equal(fresh, table)                          # the build tool's table is what the code computes
    direct = np.zeros(NW, np.uint64)
    for i in (1, 2, 5, 31):                                              # products mod phi agree with square-and-multiply on the exponent
        assert L.fastf_mt_jump_polys(i * J, 1, direct.ctypes.data) == 0
        np.testing.assert_array_equal(direct, table[(i - 1) * NW:i * NW])
    assert L.fastf_mt_jump_polys(3 * R * J, 1, direct.ctypes.data) == 0
    np.testing.assert_array_equal(direct, table[(R - 1 + 2) * NW:(R - 1 + 3) * NW])
    odd = np.zeros(2 * NW, np.uint64)
    assert L.fastf_mt_jump_polys(624 * 33, 2, odd.ctypes.data) == 0      # another stride (odd part 39 * 33 > 624: square-and-multiply)

    def array_after(seed, blocks):
        """the ORACLE generator's 624-word array after `blocks` whole blocks of draws"""
        return O.mt_stream(seed, 0, skip=624 * blocks, state=True)[1]

    for seed in (926, 1):
        a = array_after(seed, 5)
        cases = ((table[:NW], J), (table[NW:2 * NW], 2 * J), ((table[(R - 1) * NW:R * NW]), R * J), (odd[:NW], 624 * 33), (odd[NW:], 624 * 66))
        for poly, stride in cases:
            out = np.zeros(624, np.uint32)
            L.fastf_mt_jump_apply(a.ctypes.data, np.ascontiguousarray(poly).ctypes.data, out.ctypes.data)
            want = array_after(seed, 5 + stride // 624)
            np.testing.assert_array_equal(out[1:], want[1:])
            assert (int(out[0]) ^ int(want[0])) >> 31 == 0                # word 0 lends only its top bit


@pytest.mark.parametrize("devices,device,want", [
    (None, None, (0, -1)), (None, "3", (3, -1)), ("", "2", (2, -1)),
    ("1", "5", (5, -1)),                   # one device: FASTF_DEVICE says which
    ("1", None, (0, -1)),
    ("4", "5", (0, 1)),                    # a count: devices 0..3; FASTF_DEVICE is not consulted
    ("2,3", "7", (2, 3)),                  # an explicit list wins over FASTF_DEVICE
    ("2,2", None, (2, -1)),                # the second entry aliases the first: no second device
    ("3", None, (0, 1)), ("0,1,2,3", None, (0, 1)), ("5,300", None, (5, -1)), ("-1,2", None, (0, 2)),
])
def test_device_selection_from_the_environment_values(devices, device, want):
    """bam2db()'s device choice (FASTF_DEVICES / FASTF_DEVICE): round 5's dangling else made FASTF_DEVICE alone a no-op"""
    L = _lib.lib()
    L.fastf_pick_devices.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.fastf_pick_devices.restype = None
    d0, d2 = C.c_int(-9), C.c_int(-9)
    L.fastf_pick_devices(None if devices is None else devices.encode(), None if device is None else device.encode(), C.byref(d0), C.byref(d2))
    assert (d0.value, d2.value) == want


def test_big_buffers_are_private_mappings_and_the_pin_ledger_guards_their_release():
    """fastf_big_alloc: page-aligned (2 MiB from 2 MiB on) mappings of their own, never malloc heap; the ledger of registered
    ranges knows what is live; releasing unknown (heap) memory through fastf_big_free still frees it"""
    L = _lib.lib()
    L.fastf_big_alloc.restype = C.c_void_p; L.fastf_big_alloc.argtypes = [C.c_size_t]
    L.fastf_big_free.argtypes = [C.c_void_p, C.c_size_t]; L.fastf_big_free.restype = None
    L.fastf_pin_ledger_add.argtypes = [C.c_void_p, C.c_size_t]; L.fastf_pin_ledger_add.restype = None
    L.fastf_pin_ledger_remove.argtypes = [C.c_void_p]; L.fastf_pin_ledger_live.restype = C.c_int

    def mapping_of(addr):
        for ln in open("/proc/self/maps"):
            lo, hi = (int(x, 16) for x in ln.split()[0].split("-"))
            if lo <= addr < hi:
                return ln
        return ""
    small, big = L.fastf_big_alloc(100), L.fastf_big_alloc(5 << 20)
    assert small and big and small % 4096 == 0 and big % (2 << 20) == 0
    for a in (small, big):
        assert "[heap]" not in mapping_of(a)
    C.memset(big, 7, 5 << 20); C.memset(small, 7, 100)
    base = L.fastf_pin_ledger_live()
    L.fastf_pin_ledger_add(big + 4096, 8192)
    assert L.fastf_pin_ledger_live() == base + 1
    assert L.fastf_pin_ledger_remove(big + 4096) == 1 and L.fastf_pin_ledger_live() == base
    L.fastf_big_free(big, 5 << 20); L.fastf_big_free(small, 100)
    libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]
    L.fastf_big_free(libc.malloc(64), 64)                               # heap memory: passed on to free()


def test_releasing_a_registered_buffer_is_caught(tmp_path):
    """FASTF_DEBUG_PINS=1: fastf_big_free of a range the ledger holds aborts the process (a child process here)"""
    code = ("import ctypes as C, sys; sys.path.insert(0, %r)\n"
            "from fastf_amd import _lib\nL = _lib.lib()\n"
            "L.fastf_big_alloc.restype = C.c_void_p; L.fastf_big_alloc.argtypes = [C.c_size_t]\n"
            "L.fastf_big_free.argtypes = [C.c_void_p, C.c_size_t]; L.fastf_pin_ledger_add.argtypes = [C.c_void_p, C.c_size_t]\n"
            "p = L.fastf_big_alloc(1 << 20); L.fastf_pin_ledger_add(p + 4096, 4096); L.fastf_big_free(p, 1 << 20); print('survived')\n") % ROOT
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, FASTF_DEBUG_PINS="1"))
    assert r.returncode == -6 and "still registered with the HIP runtime" in r.stderr and "survived" not in r.stdout
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, FASTF_DEBUG_PINS="0"))
    assert r.returncode == 0 and "INTERNAL ERROR" in r.stderr and "survived" in r.stdout


def test_one_hip_runtime_whatever_the_import_order():
    """fastf_amd loaded BEFORE torch used to bring in /opt/rocm's libamdhip64 and torch then its bundled copy beside it: two HIP
    runtimes in one process.  _lib.lib() now loads torch's copy first when torch is installed."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\nfrom fastf_amd import _lib\n_lib.lib()\nimport torch\n"
            "print(len(_lib.hip_runtimes_loaded()), _lib.hip_runtimes_loaded())\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split()[0] == "1", r.stdout
