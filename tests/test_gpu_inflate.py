"""BGZF inflate on the device (gpu_frontend.hpp: gpu_inflate2.hpp's lane-per-block decoder + the match resolver, and
gpu_inflate.hpp's one wavefront per block) against zlib, and the BAM reader with FASTF_GPU_INFLATE on against the host-only reader."""
import ctypes as C
import zlib

import numpy as np
import pytest

import fastf_amd as F
from fastf_amd import synth, _lib
from helpers import Case
from test_gpu_inflate_host import raw_streams, sample_payloads
from test_bam_reader import read_all

pytestmark = pytest.mark.gpu


class Blk(C.Structure):
    _fields_ = [("coff", C.c_uint64), ("clen", C.c_uint32), ("isize", C.c_uint32), ("uoff", C.c_uint64)]


@pytest.mark.parametrize("kernel", ["lanes", "wave"])
def test_device_inflate_matches_zlib(kernel, monkeypatch):
    """both device decoders: the two-kernel form (one lane per block decodes to literals + match tokens, one wave per block
    resolves the matches: gpu_inflate2.hpp) and round 4's one wavefront per block (FASTF_GI_KERNEL=wave)"""
    import torch  # noqa: F401
    if kernel == "wave":
        monkeypatch.setenv("FASTF_GI_KERNEL", "wave")
    else:
        monkeypatch.delenv("FASTF_GI_KERNEL", raising=False)
    L = _lib.lib()
    L.fastf_gpuinf_create.restype = C.c_void_p; L.fastf_gpuinf_create.argtypes = [C.c_int]
    L.fastf_gpuinf_destroy.argtypes = [C.c_void_p]
    L.fastf_gpuinf_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(7)
    blocks = []
    for data in sample_payloads(rng, 30):
        for comp in raw_streams(data):
            blocks.append((comp, data))
    assert len(blocks) > 800                                          # several slices
    comp_all = bytearray(); desc = (Blk * len(blocks))(); uoff = 0
    for i, (c, d) in enumerate(blocks):
        comp_all += b"\x1f\x8b" * 9                                    # something between the payloads, like BGZF headers
        desc[i] = Blk(len(comp_all), len(c), len(d), uoff)
        comp_all += c; uoff += len(d)
    comp_all += bytes(64)
    out_p = L.fastf_pinned_alloc(max(uoff, 1)); assert out_p
    status = (C.c_uint8 * len(blocks))()
    cbuf = L.fastf_pinned_alloc(len(comp_all) + 64); assert cbuf                    # the driver takes pinned compressed bytes
    C.memmove(cbuf, bytes(comp_all), len(comp_all))
    g = L.fastf_gpuinf_create(0); assert g, L.fastf_last_error()
    try:
        for _ in range(2):                                             # second run reuses the slots
            assert L.fastf_gpuinf_run(g, cbuf, desc, len(blocks), out_p, status) == 0, L.fastf_last_error()
            out = C.string_at(out_p, uoff)
            assert bytes(status) == bytes(len(blocks))
            o = 0
            for c, d in blocks:
                assert out[o:o + len(d)] == d
                o += len(d)
        # ordinary (heap) memory on both sides: fastf_gpuinf_run registers the ranges for the call — the compressed bytes with the
        # 64 bytes of slack the submit copies along — and unregisters them; nothing stays in the ledger
        L.fastf_debug_live_registrations.restype = int
        live0 = L.fastf_debug_live_registrations()
        h_comp = np.frombuffer(bytes(comp_all), dtype=np.uint8).copy()
        h_out = np.zeros(max(uoff, 1), dtype=np.uint8)
        st2 = (C.c_uint8 * len(blocks))()
        assert L.fastf_gpuinf_run(g, h_comp.ctypes.data, desc, len(blocks), h_out.ctypes.data, st2) == 0, L.fastf_last_error()
        assert bytes(st2) == bytes(len(blocks)) and h_out.tobytes() == b"".join(d for _, d in blocks)
        assert L.fastf_debug_live_registrations() == live0
        # a corrupted block is declined (or its CRC will differ), the others are untouched
        bad = bytearray(comp_all); bad[desc[5].coff + 3] ^= 0x5A
        C.memmove(cbuf, bytes(bad), len(bad))
        assert L.fastf_gpuinf_run(g, cbuf, desc, len(blocks), out_p, status) == 0
        out = C.string_at(out_p, uoff)
        o = 0
        for i, (c, d) in enumerate(blocks):
            if i != 5:
                assert status[i] == 0 and out[o:o + len(d)] == d
            o += len(d)
    finally:
        L.fastf_gpuinf_destroy(g)
        L.fastf_pinned_free(out_p)
        L.fastf_pinned_free(cbuf)


@pytest.mark.parametrize("window", [1 << 17, 1 << 22, 128 << 20])
def test_reader_with_device_inflate_gives_the_same_records(tmp_path, monkeypatch, window):
    case = Case(n=120_000, n_bar=300, n_gene=120, umi_pool=128, p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.2, p_n_umi=0.02, p_no_ub=0.03)
    lists = case.lists()
    bam = tmp_path / "t.bam"
    shape = lambda i: (91, 1, 0, i * 37 % 100000)
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub, shape=shape)
    monkeypatch.setenv("FASTF_BAM_WINDOW", str(window))
    monkeypatch.setenv("FASTF_GPU_INFLATE", "2")                       # wait for the device: every window goes through it
    monkeypatch.delenv("FASTF_BAM_PROFILE", raising=False)          # (no reader trace in the suite's output)
    got = read_all(bam, lists, cap=50_000)
    for g, w in zip(got, case.packed(lists)):
        np.testing.assert_array_equal(g, w)
