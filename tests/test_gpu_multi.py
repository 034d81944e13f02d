"""One host process driving several devices behind fastf_engine_push / _finish (fastf_engine_config_t.n_devices,
bam2db(): FASTF_DEVICES) — SURVEY 8e through the C entry points.  The box has one GPU, so the devices alias: every
shard lives on device 0 and the key exchange runs as device-to-device copies; the dealing of chunks, the hit-rank
bases across devices, the per-destination buffers, the local sorts and the merge by cell are the real thing, and the
result must be the oracle's, bit for bit."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import fastf_amd as F
from fastf_amd import synth, _lib
from helpers import Case, assert_matches_oracle

pytestmark = pytest.mark.gpu

CASES = {
    "mix": dict(n=300_000, n_bar=900, n_gene=400, rate_cell=0.8, rate_depth=0.6, umi_pool=512, cell_dist="lognormal",
                p_no_cb=0.03, p_unlisted_cb=0.1, p_bad_xf=0.1, p_n_umi=0.01),
    "keepall": dict(n=120_000, n_bar=300, n_gene=100, umi_pool=64),
    "c3 shape": dict(n=400_000, n_bar=30_000, n_gene=4000, rate_cell=0.5, rate_depth=0.5, umi_len=12, dup_factor=4.0,
                     cell_dist="lognormal", gene_dist="zipf", p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.15, p_n_umi=0.001),
}


@pytest.mark.parametrize("G", [2, 3, 4, 8])
@pytest.mark.parametrize("name", list(CASES))
def test_multi_device_engine_matches_oracle(name, G):
    case = Case(**CASES[name])
    ora = case.oracle()
    lists = case.lists()
    cbk, gxk, umi, meta = case.packed(lists)
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, umi_max_bases=12,
                              batch_records=17_000, devices=[0] * G)          # several rounds, the last one short
    try:
        cuts = [0, 1, 40_000, 40_001, case.n // 2, case.n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            eng.push(cbk[a:b], gxk[a:b], umi[a:b], meta[a:b])
        res = eng.finish()
        assert_matches_oracle(res, ora, eng, case, lists, eng.umi_rows())
        again = eng.finish()                                                   # idempotent
        assert again["nnz"] == res["nnz"] and np.array_equal(again["count"], res["count"])
        eng.reset(); eng.reseed(case.seed, lists.mt_skip)
        eng.push(cbk, gxk, umi, meta)
        assert_matches_oracle(eng.finish(), ora)
    finally:
        eng.close()


WIDE = {
    # 210 k barcodes x 70 k genes x 16-base UMIs: 71 key bits
    "71 bits": (dict(n=150_000, n_bar=210_000, n_gene=70_000, rate_depth=0.9, umi_len=16, umi_pool=4096, p_n_umi=0.01), 16),
    "20 bases": (dict(n=200_000, n_bar=2000, n_gene=900, rate_cell=0.7, rate_depth=0.8, umi_len=20, dup_factor=3.0, p_no_cb=0.03,
                      p_unlisted_cb=0.05, p_bad_xf=0.1, p_n_umi=0.02), 24),
    "30 bases": (dict(n=120_000, n_bar=300, n_gene=100, umi_len=30, umi_pool=256, rate_depth=0.7, p_n_umi=0.05), 32),
}


@pytest.mark.parametrize("G", [2, 4])
@pytest.mark.parametrize("name", list(WIDE))
def test_multi_device_wide_keys_match_oracle(name, G):
    """keys wider than 64 bits and UMIs beyond 16 bases through the multi-device engine: K1b writes the group word and the rest of
    the key per destination, both cross in the one exchange into the receiver's own wide store, every device sorts and reduces
    what it owns (hashtable.c:70-115, bam2db_ds.c:417-419: any list size, any UMI length)"""
    kw, umi_max = WIDE[name]
    case = Case(**kw)
    ora = case.oracle()
    lists = case.lists()
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, umi_max_bases=umi_max, batch_records=23_000, devices=[0] * G)
    try:
        assert eng.key_bits > 64 or umi_max > 16
        if umi_max > 16:
            cb, gx, umi, meta, ext = case.packed_long(lists)
            assert ext.any()
        else:
            cb, gx, umi, meta = case.packed(lists); ext = None
        cuts = [0, 1, 50_000, case.n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            eng.push(cb[a:b], gx[a:b], umi[a:b], meta[a:b], umi_ext=None if ext is None else ext[a:b])
        res = eng.finish()
        assert_matches_oracle(res, ora, eng, case, lists, eng.umi_rows())
        counts = eng.device_records(G)
        assert len(counts) == G and all(c > 0 for c in counts)
        eng.reset(); eng.reseed(case.seed, lists.mt_skip)                      # the same job again on the same handle
        eng.push(cb, gx, umi, meta, umi_ext=ext)
        assert_matches_oracle(eng.finish(), ora)
    finally:
        eng.close()


def test_multi_device_chunks_with_millions_of_hits_use_the_parallel_generator():
    """a chunk with more than 4 x 624 x 256 CB hits: every device continues the one MT19937 stream by jump-ahead on many
    workgroups (launch_mt_decisions_par) — the decisions must still be the serial stream's, draw for draw (bam2db_ds.c:385)"""
    case = Case(n=3_300_000, n_bar=500, n_gene=200, rate_cell=1.0, rate_depth=0.5, umi_pool=4096, p_bad_xf=0.05)
    ora = case.oracle()
    lists = case.lists()
    cbk, gxk, umi, meta = case.packed(lists)
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, umi_max_bases=12, batch_records=1_600_000, devices=[0, 0])
    try:
        eng.push(cbk, gxk, umi, meta)                       # two chunks of 1.6 M hits, a third of 0.1 M (the serial kernel)
        assert_matches_oracle(eng.finish(), ora)
    finally:
        eng.close()


def test_multi_device_empty_and_tiny():
    case = Case(n=10, n_bar=4, n_gene=3)
    lists = case.lists()
    eng = F.Engine.from_lists(lists, devices=[0, 0, 0])
    try:
        res = eng.finish()
        assert res["nnz"] == 0 and res["total"] == 0
        eng.push(*case.packed(lists))
        assert_matches_oracle(eng.finish(), case.oracle(), eng, case, lists, eng.umi_rows())
    finally:
        eng.close()


def test_multi_device_deep_groups_finish_the_sort():
    """one cell x one gene with thousands of UMIs: the group-only sort raises its run flag on a shard and the engine
    sorts fully everywhere"""
    case = Case(n=60_000, n_bar=2, n_gene=1, umi_len=12)
    lists = case.lists()
    eng = F.Engine.from_lists(lists, umi_max_bases=12, devices=[0, 0])
    try:
        eng.push(*case.packed(lists))
        assert_matches_oracle(eng.finish(), case.oracle())
    finally:
        eng.close()


def test_rccl_exchange_call_sequence_on_one_device(monkeypatch):
    """the RCCL backend of the exchange (dlopen, ncclCommInitAll, one group of send/recv on the compute stream) with a
    single rank: the only form of it a one-GPU box can run"""
    monkeypatch.setenv("FASTF_FORCE_MULTI", "1")
    monkeypatch.setenv("FASTF_EXCHANGE", "rccl")
    case = Case(n=50_000, n_bar=200, n_gene=100, rate_depth=0.5, umi_pool=64)
    lists = case.lists()
    eng = F.Engine.from_lists(lists, rate_depth=0.5, seed=case.seed, devices=[0])
    try:
        eng.push(*case.packed(lists))
        assert_matches_oracle(eng.finish(), case.oracle())
    finally:
        eng.close()


@pytest.mark.parametrize("umi_len", [12, 22])
def test_cli_with_fastf_devices(tmp_path, umi_len):
    """(22 bases: the first run's 64-bit keys overflow, bam2db() runs again with room for 32 — wide keys on every device)"""
    case = Case(n=90_000, n_bar=600, n_gene=250, rate_cell=0.5, rate_depth=0.5, umi_len=umi_len, dup_factor=3.0,
                p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.15, p_n_umi=0.005)
    bam = tmp_path / "in.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub)
    b, f = tmp_path / "b.tsv", tmp_path / "f.tsv"
    b.write_bytes(case.bt); f.write_bytes(case.ft)
    out = tmp_path / "out"; out.mkdir()
    case.label = str(bam).encode()
    ora = case.oracle()
    env = dict(os.environ, FASTF_DEVICES="0,0,0,0", FASTF_BATCH_RECORDS="9000")
    r = subprocess.run([_lib.cli_path(), "bam2db", "-b", str(bam), "-a", str(b), "-f", str(f), "-o", str(out),
                        "-c", "0.5", "-r", "0.5", "-u"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    rd = lambda n: gzip.decompress((out / n).read_bytes())
    assert rd("matrix.mtx.gz") == ora["matrix"]
    assert rd("umi.tsv.gz") == ora["umi"]


@pytest.mark.parametrize("G,pinned,tile_form", [(4, False, False), (3, True, False), (2, False, True)])
def test_every_device_takes_part_when_pushes_are_small(G, pinned, tile_form, monkeypatch):
    """bam2db() pushes one batch of at most batch_records records per call: chunk i of the STREAM goes to device i mod G
    across calls (a cursor that restarted with every push left every chunk on device 0), ragged pushes included, and the
    asynchronous pipeline (a chunk is retired only when its device is needed again) gives the oracle's matrix"""
    if tile_form:
        monkeypatch.setenv("FASTF_NO_STREAM_K1B", "1")      # K1b in its tile form (what sharded passes ran before the partition kernel)
    case = Case(n=210_000, n_bar=700, n_gene=300, rate_cell=0.7, rate_depth=0.5, umi_pool=256, cell_dist="lognormal",
                p_no_cb=0.03, p_unlisted_cb=0.1, p_bad_xf=0.1, p_n_umi=0.01)
    lists = case.lists()
    cbk, gxk, umi, meta = case.packed(lists)
    cap = 9_000
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, umi_max_bases=12, batch_records=cap, devices=[0] * G)
    try:
        rng = np.random.default_rng(5)
        off, n_push = 0, 0
        if pinned:
            pb = F.PinnedBatch(case.n)
            pb.fill(0, cbk, gxk, umi, meta)
        while off < case.n:
            n = int(min(case.n - off, rng.integers(1, cap + 1)))                 # never more than one chunk per call
            if pinned:
                eng.push_pinned(pb, off, off + n)
            else:
                eng.push(cbk[off:off + n], gxk[off:off + n], umi[off:off + n], meta[off:off + n])
            off += n; n_push += 1
        per_dev = eng.device_records(G)
        assert sum(per_dev) == case.n and all(r > 0 for r in per_dev)
        assert max(per_dev) - min(per_dev) <= case.n // 3, per_dev                # dealt, not piled up
        assert_matches_oracle(eng.finish(), case.oracle(), eng, case, lists, eng.umi_rows())
    finally:
        eng.close()
        if pinned:
            pb.close()
