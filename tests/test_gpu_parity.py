"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, bit-exact."""
import os
import sys

import numpy as np
import pytest

import fastf_amd as F
from fastf_amd import hostmem
from helpers import Case, assert_matches_oracle

pytestmark = pytest.mark.gpu

CASES = {
    # BASELINE config 1: 100 k records, 1 k barcodes x 500 genes, keep-all, 64-UMI pool
    "c1_100k": dict(n=100_000, n_bar=1000, n_gene=500, umi_pool=64),
    "c1_half": dict(n=100_000, n_bar=1000, n_gene=500, umi_pool=64, rate_cell=0.5, rate_depth=0.5),
    "mixed": dict(n=300_000, n_bar=3000, n_gene=2000, rate_cell=0.5, rate_depth=0.5, data_seed=7,
                  cell_dist="lognormal", gene_dist="zipf", umi_len=12, dup_factor=4.0,
                  p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.15, p_n_umi=0.01, p_multi_gene=0.02, p_no_ub=0.01),
    "tiny": dict(n=37, n_bar=5, n_gene=3, umi_pool=4, p_n_umi=0.2),
    "one_tile_edge": dict(n=8192, n_bar=50, n_gene=20, umi_pool=16),
    "tile_plus_one": dict(n=8193, n_bar=50, n_gene=20, umi_pool=16),
    "skewed": dict(n=200_000, n_bar=2000, n_gene=300, gene_dist="zipf", umi_pool=4096, zipf_umi=1.5, data_seed=5),
    "depth_tiny": dict(n=50_000, n_bar=100, n_gene=50, rate_depth=0.01, umi_pool=8),
    "cells_none": dict(n=1000, n_bar=10, n_gene=5, rate_cell=0.05),
    # groups far larger than a reduce tile (4096 keys): the carry fix-up path, counts in the tens of thousands
    "huge_groups": dict(n=400_000, n_bar=3, n_gene=2, umi_len=12, data_seed=11),
    "one_group_all_dups": dict(n=100_000, n_bar=1, n_gene=1, umi_pool=1),
    "one_group_null_umis": dict(n=30_000, n_bar=1, n_gene=1, umi_pool=3, p_n_umi=0.9),
    # BASELINE config 5 shape, scaled: 100 k barcodes, Zipf-skewed UMI reuse from a 4096-pool per gene
    "c5_like": dict(n=1_500_000, n_bar=100_000, n_gene=400, umi_len=12, umi_pool=4096, zipf_umi=1.5, data_seed=9),
    # sparse gene id ranges (real Ensembl lists): bitmap + rank + permutation image; the second needs > 78 KB of LDS
    # (one workgroup per CU, the 128-VGPR build of K1b)
    "sparse_gene_ids": dict(n=200_000, n_bar=800, n_gene=5000, gene_stride=7, gene_dist="zipf", umi_pool=2048, p_bad_xf=0.1),
    "sparse_gene_ids_big_image": dict(n=300_000, n_bar=2000, n_gene=30_000, gene_stride=8, umi_pool=4096, rate_depth=0.8,
                                      p_unlisted_cb=0.05),
    # gene ids that start far from zero (direct table and bitmap are indexed relative to the smallest listed id)
    "gene_ids_offset_dense": dict(n=100_000, n_bar=500, n_gene=4000, gene_start=123_456_789, umi_pool=512, p_bad_xf=0.1),
    "gene_ids_offset_sparse": dict(n=100_000, n_bar=500, n_gene=4000, gene_start=99_000_000, gene_stride=13, umi_pool=512),
    # a barcode list too large for the LDS perfect hash: L2 table behind the LDS miss filter, most CB tags not sampled
    # or not listed (the filter's false positives reach the table and must miss there)
    "big_barcode_list_mostly_misses": dict(n=400_000, n_bar=30_000, n_gene=300, rate_cell=0.3, rate_depth=0.9, umi_pool=2048,
                                           p_unlisted_cb=0.3, p_no_cb=0.05, data_seed=13),
    "every_record_misses": dict(n=20_000, n_bar=50, n_gene=20, p_unlisted_cb=1.0),
    "no_cb_at_all": dict(n=20_000, n_bar=50, n_gene=20, p_no_cb=1.0),
}


@pytest.mark.parametrize("name", list(CASES))
def test_engine_matches_oracle(name):
    case = Case(**CASES[name])
    ora = case.oracle()
    lists = case.lists()
    umi_max = 12
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, umi_max_bases=umi_max)
    try:
        eng.push(*case.packed(lists))
        res = eng.finish()
        rows = eng.umi_rows()
        assert_matches_oracle(res, ora, eng, case, lists, rows)
    finally:
        eng.close()


def test_mixed_umi_lengths_0_to_16():
    """UMI lengths 0..16 in one run (blob byte lengths 0..4): the length field and zero padding decide equality"""
    from fastf_amd import synth
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    n = 60_000
    bt, ft, bar, genes = synth.make_lists(40, 30, seed=5)
    flags, xf, cb, gx, _ = synth.make_records(n, bar, genes, seed=6)
    lens = rng.integers(0, 17, size=n)
    pool = [b"", b"A", b"AC", b"ACG", b"ACGT", b"ACGTA", b"ACGTAAAA", b"ACGTAAAAA", b"ACGTAAAAAAAA", b"ACGTAAAAAAAAA",
            b"ACGTACGTACGTACGT", b"ACGTACGTACGTACGA", b"TTTTTTTTTTTTTTTT", b"ACGTN", b"NNNN", b"TTTT", b"TTTTA", b"TTTTAAAA"]
    ub = np.array([pool[i % len(pool)] if rng.random() < 0.7 else bytes(rng.choice(list(b"ACGT"), size=lens[i]).tolist())
                   for i in range(n)], dtype="S17")
    cb, gx = synth.as_cstr(cb), synth.as_cstr(gx)
    ora = O.run_bam2db(bt, ft, flags, xf, cb, gx, ub, 1.0, 0.8, 926, b"synthetic.bam", True)
    lists = F.Lists(bt, ft, 1.0, 926)
    eng = F.Engine.from_lists(lists, rate_depth=0.8, seed=926, umi_max_bases=16)
    try:
        eng.push(*F.pack_records(lists, flags, xf, cb, gx, ub))
        res = eng.finish()
        case = Case(n=1, n_bar=2, n_gene=2); case.rate_cell, case.rate_depth, case.label = 1.0, 0.8, b"synthetic.bam"
        assert_matches_oracle(res, ora, eng, case, lists, eng.umi_rows())
    finally:
        eng.close()


def test_long_unsorted_runs_fall_back_to_the_full_sort():
    """two (cell, feature) groups whose 12-bp UMIs all start with the same 5 bases: the run of keys that agree on the
    sorted bits holds ~16 k distinct UMIs, far beyond the group-only path's cap — the engine must finish the sort itself"""
    from fastf_amd import synth
    from oracle import oracle as O
    rng = np.random.default_rng(17)
    n = 120_000
    bt, ft, bar, genes = synth.make_lists(2, 1, seed=5)
    flags, xf, cb, gx, _ = synth.make_records(n, bar, genes, seed=6)
    tail = synth._kmers(rng.integers(0, 1 << 14, size=n, dtype=np.uint64), 7)
    ub = synth._as_S(np.concatenate([np.tile(np.frombuffer(b"ACGTA", dtype=np.uint8), (n, 1)), tail], axis=1), 13)
    cb, gx = synth.as_cstr(cb), synth.as_cstr(gx)
    ora = O.run_bam2db(bt, ft, flags, xf, cb, gx, ub, 1.0, 1.0, 926, b"synthetic.bam", True)
    assert ora["count"].max() > 10_000
    lists = F.Lists(bt, ft, 1.0, 926)
    eng = F.Engine.from_lists(lists, umi_max_bases=12)
    try:
        assert 0 < eng.skip_bits < 27 and eng.sort_passes(True) < eng.sort_passes(False)
        eng.push(*F.pack_records(lists, flags, xf, cb, gx, ub))
        res = eng.finish()
        case = Case(n=1, n_bar=2, n_gene=2); case.rate_cell, case.rate_depth, case.label = 1.0, 1.0, b"synthetic.bam"
        assert_matches_oracle(res, ora, eng, case, lists, eng.umi_rows())
    finally:
        eng.close()


def test_umi_longer_than_engine_limit_is_an_error_not_a_wrong_answer():
    from fastf_amd import synth
    bt, ft, bar, genes = synth.make_lists(10, 5, seed=5)
    flags, xf, cb, gx, ub = synth.make_records(5000, bar, genes, seed=6, umi_len=14)
    lists = F.Lists(bt, ft, 1.0, 926)
    eng = F.Engine.from_lists(lists, umi_max_bases=12)
    try:
        eng.push(*F.pack_records(lists, flags, xf, synth.as_cstr(cb), synth.as_cstr(gx), synth.as_cstr(ub)))
        with pytest.raises(F.FastfError) as ei:
            eng.finish()
        assert "umi_max_bases" in str(ei.value)
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["c1_half", "mixed", "skewed", "tile_plus_one", "huge_groups"])
@pytest.mark.parametrize("env", [{"FASTF_LDS_TABLES": "0"}, {"FASTF_LDS_CELLS": "0"}, {"FASTF_SORT_SKIP_BITS": "0"},
                                 {"FASTF_GENES_NO_DIRECT": "1"}, {"FASTF_FORCE_WIDE_KEYS": "1"}, {"FASTF_HOST_DRAWS": "1"}])
def test_general_paths_match_oracle(name, env, monkeypatch):
    """the same cases with the LDS tables off (L2 open-addressed probes) and with the full 7-pass sort"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    case = Case(**CASES[name])
    ora = case.oracle()
    lists = case.lists()
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, umi_max_bases=12)
    try:
        eng.push(*case.packed(lists))
        assert_matches_oracle(eng.finish(), ora, eng, case, lists, eng.umi_rows())
    finally:
        eng.close()


def test_mixed_name_forms_use_both_lookup_paths():
    """barcodes and gene ids of several forms: 16-mers with and without suffix, free text, two Ensembl families, plain
    names, versioned ids — the barcode list is not LDS-eligible (L2 probes), the gene list has one LDS family and the
    rest resolved through the L2 table inside the same kernel"""
    from fastf_amd import synth
    from oracle import oracle as O
    rng = np.random.default_rng(23)
    def mer(k): return bytes(rng.choice(list(b"ACGT"), k).tolist())
    bars = [mer(16) + b"-1" for _ in range(300)] + [mer(16) for _ in range(50)] + [mer(14) + b"-2" for _ in range(50)] + \
           [b"cell_%d" % i for i in range(40)] + [mer(26) + b"-1" for _ in range(10)]
    genes = [b"ENSG%011d" % (1000 + 7 * i) for i in range(400)] + [b"ENSMUSG%011d" % (5 + 3 * i) for i in range(150)] + \
            [b"GFP", b"mCherry", b"ENSG00000000003.14", b"LINC01409", b"7SK", b"AC114498.1"]
    bt = b"".join(b + b"\n" for b in bars)
    ft = b"".join(g + b"\tname\tGene Expression\n" for g in genes)
    n = 120_000
    unl_b = [mer(16) + b"-1" for _ in range(50)] + [b"cell_x", b"", b"ACGT"]
    unl_g = [b"ENSG%011d" % 999, b"ENSG%011d" % (1000 + 7 * 400), b"ENSG%011d" % 1001, b"ENSMUSG%011d" % 6, b"GFP2", b"gfp",
             b"ENSG00000000003", b"ENSG00000000003.15", b"ENSG0000001007"]
    cb = np.array([(bars + unl_b)[i] for i in rng.integers(0, len(bars) + len(unl_b), n)], dtype="S32")
    gx = np.array([(genes + unl_g)[i] for i in rng.integers(0, len(genes) + len(unl_g), n)], dtype="S32")
    ub = synth._as_S(synth._kmers(rng.integers(0, 1 << 20, size=n, dtype=np.uint64) % np.uint64(3000), 10), 11)
    flags = np.full(n, 15, np.uint8); xf = np.where(rng.random(n) < 0.9, 25, 3).astype(np.int32)
    ora = O.run_bam2db(bt, ft, flags, xf, cb, gx, ub, 0.8, 0.7, 926, b"synthetic.bam", True)
    assert ora["nnz"] > 10_000
    lists = F.Lists(bt, ft, 0.8, 926)
    eng = F.Engine.from_lists(lists, rate_depth=0.7, seed=926, umi_max_bases=12)
    try:
        eng.push(*F.pack_records(lists, flags, xf, cb, gx, ub))
        res = eng.finish()
        case = Case(n=1, n_bar=2, n_gene=2); case.rate_cell, case.rate_depth, case.label = 0.8, 0.7, b"synthetic.bam"
        assert_matches_oracle(res, ora, eng, case, lists, eng.umi_rows())
    finally:
        eng.close()


@pytest.mark.parametrize("host_draws", ["0", "1"])
def test_multi_batch_push_equals_single(host_draws, monkeypatch):
    """pushes cut at odd places, chunks of 20 000 records: the decision stream is continued chunk by chunk — by mt_fill_kernel
    (a word of 32 decisions straddles two launches) and, with FASTF_HOST_DRAWS=1, by the host packer (the same word goes up
    twice) — then the same records with caller-supplied draws"""
    monkeypatch.setenv("FASTF_HOST_DRAWS", host_draws)
    case = Case(n=150_000, n_bar=800, n_gene=400, rate_depth=0.7, umi_pool=256, p_unlisted_cb=0.1, p_bad_xf=0.1)
    ora = case.oracle()
    lists = case.lists()
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, batch_records=20_000)
    try:
        cbk, gxk, umi, meta = case.packed(lists)
        cuts = [0, 1, 4097, 50_000, 50_001, 120_000, 150_000]
        for a, b in zip(cuts[:-1], cuts[1:]):
            eng.push(cbk[a:b], gxk[a:b], umi[a:b], meta[a:b])
        res = eng.finish()
        assert_matches_oracle(res, ora, eng, case, lists, eng.umi_rows())
        # reset + caller-supplied draws give the same answer
        eng.reset()
        draws = F.mt_draws(case.seed, lists.mt_skip, case.n)
        eng.push(cbk, gxk, umi, meta, draws=draws)
        assert_matches_oracle(eng.finish(), ora)
    finally:
        eng.close()


def test_engine_reuse_cycles_grow_the_row_buffer():
    """finish is idempotent; reset + push + finish cycles reuse the engine's buffers: the row buffer is ordinary memory on
    its first use, pinned from the second, and reallocated (while pinned) when a later run has more rows"""
    small = Case(n=20_000, n_bar=60, n_gene=40, umi_pool=64, data_seed=3)
    big = Case(n=200_000, n_bar=60, n_gene=40, umi_pool=4096, data_seed=3)      # the same lists (they follow data_seed), ten times the records
    lists = small.lists()
    assert big.bt == small.bt and big.ft == small.ft
    live0 = _live_registrations()
    eng = F.Engine.from_lists(lists, rate_depth=small.rate_depth, seed=small.seed, batch_records=16_384)
    try:
        for case in (small, small, big, small, big):
            eng.reset()
            eng.reseed(case.seed, lists.mt_skip)           # (reset alone lets the draw stream go on, like the reference's global generator)
            eng.push(*case.packed(lists))
            res = eng.finish()
            again = eng.finish()
            for k in ("feature", "cell", "count"):
                np.testing.assert_array_equal(res[k], again[k])
            assert_matches_oracle(res, case.oracle(), eng, case, lists, eng.umi_rows())
        assert _live_registrations() == live0 + 1            # the row buffer, pinned in place since its second use
    finally:
        eng.close()
    assert _live_registrations() == live0                    # ... and unregistered before it was unmapped


def _live_registrations():
    from fastf_amd import _lib
    L = _lib.lib()
    L.fastf_debug_live_registrations.restype = int
    return L.fastf_debug_live_registrations()


def test_tables_from_heap_arrays_that_are_recycled_before_the_rows_come_back():
    """Round 5's fault, as a sequence: a multi-megabyte barcode table uploaded from heap memory at create (100 k barcodes: the L2
    table), that memory freed and the same sizes allocated again, then the rows of a FIRST finish copied into a row buffer that is
    not pinned yet, and the -u rows into the caller's arrays.  None of these copies hands heap memory to the runtime (bounce
    buffer), the row buffer is a mapping of its own, and nothing stays registered."""
    import gc
    live0 = _live_registrations()
    case = Case(n=600_000, n_bar=100_000, n_gene=3000, umi_pool=4096, data_seed=5, rate_depth=0.9)
    ora = case.oracle()
    lists = case.lists()
    ck, fk = lists.cell_keys.copy(), lists.feature_keys.copy()          # heap arrays of 0.8 MB and 24 KB
    eng = F.Engine(ck, fk, rate_depth=case.rate_depth, seed=case.seed, mt_skip=lists.mt_skip)
    try:
        eng._cell_keys = eng._feature_keys = None
        del ck, fk
        gc.collect()
        junk = [np.full(100_000, i, dtype=np.uint64) for i in range(8)]          # the allocator hands the freed chunks out again
        eng.push(*case.packed(lists))
        res = eng.finish()
        rows = eng.umi_rows()
        assert_matches_oracle(res, ora, eng, case, lists, rows)
        assert all(int(j[0]) == i for i, j in enumerate(junk))
        assert _live_registrations() == live0                            # a first finish pins nothing
    finally:
        eng.close()
    assert _live_registrations() == live0


def test_empty_input():
    case = Case(n=10, n_bar=4, n_gene=3)
    lists = case.lists()
    eng = F.Engine.from_lists(lists)
    try:
        res = eng.finish()
        assert res["nnz"] == 0 and res["total"] == 0
        assert eng.umi_rows()["n"] == 0
    finally:
        eng.close()


def test_survey_edge_case_fixture():
    """the 9-record edge-case BAM of SURVEY.md §8c (reference output recorded there)"""
    import json, os
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_8c.json")))["edge_case"]
    bt, ft = fx["barcodes"].encode(), fx["features"].encode()
    recs = fx["records"]
    flags = np.full(len(recs), 15, np.uint8)
    xf = np.full(len(recs), 25, np.int32)
    cb = np.array([r[0].encode() for r in recs], dtype="S8")
    gx = np.array([r[1].encode() for r in recs], dtype="S8")
    ub = np.array([r[2].encode() for r in recs], dtype="S16")
    lists = F.Lists(bt, ft, 1.0, 926)
    eng = F.Engine.from_lists(lists, umi_max_bases=16)
    try:
        eng.push(*F.pack_records(lists, flags, xf, cb, gx, ub))
        res = eng.finish()
        assert [res["total"], res["sampled"], res["valid"]] == fx["counters"]
        rows = ["%d %d %d" % t for t in zip(res["feature"], res["cell"], res["count"])]
        assert rows == fx["matrix_rows"]
        assert eng.format_umi_rows(eng.umi_rows()).decode().splitlines() == fx["umi_rows"]
    finally:
        eng.close()


# key widths around the byte boundaries: the matrix sort's digit grid is anchored at the top key bit, so its shifts are
# run-time values (non-byte-aligned) for most of these; -u rows take the full, byte-aligned sort of the same keys
WIDTHS = [
    # (n_bar, n_gene, umi_max_bases, expected key bits)
    (3, 1, 16, 2 + 1 + 36),
    (200, 100, 16, 8 + 7 + 36),
    (1000, 500, 10, 10 + 9 + 23),
    (5000, 3000, 12, 13 + 12 + 27),
    (20_000, 40_000, 12, 15 + 16 + 27),
    (40_000, 40_000, 12, 16 + 16 + 27),
    (70_000, 40_000, 12, 17 + 16 + 27),
    (70_000, 70_000, 12, 17 + 17 + 27),
    (2000, 1500, 16, 11 + 11 + 36),
    (260_000, 600, 16, 18 + 10 + 36),                       # 64 bits exactly
]


@pytest.mark.parametrize("n_bar,n_gene,umi_max,bits", WIDTHS)
def test_key_widths(n_bar, n_gene, umi_max, bits):
    case = Case(n=150_000, n_bar=n_bar, n_gene=n_gene, rate_cell=0.9, rate_depth=0.7, umi_len=min(umi_max, 12), dup_factor=3.0,
                cell_dist="lognormal", gene_dist="zipf", p_unlisted_cb=0.05, p_bad_xf=0.1, p_n_umi=0.01, data_seed=bits)
    ora = case.oracle()
    lists = case.lists()
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, umi_max_bases=umi_max)
    try:
        if case.rate_cell == 0.9 and lists.n_cells == int(np.float32(n_bar) * np.float32(0.9)):
            cb = max(1, int(lists.n_cells).bit_length())
            assert eng.key_bits == bits - max(1, n_bar.bit_length()) + cb      # sampled cells set the cell field
        assert eng.sort_passes(True) <= eng.sort_passes(False)
        eng.push(*case.packed(lists))
        res = eng.finish()
        assert_matches_oracle(res, ora, eng, case, lists, eng.umi_rows())
    finally:
        eng.close()


# keys wider than 64 bits (hashtable.c:70-115 takes any number of barcodes and features, bam2db_ds.c:417-419 any UMI): the
# engine sorts the (cell, feature) word and carries the rest of the key beside it
WIDE_CASES = {
    # 18 + 17 + 36 = 71 bits: L2 tables for both lists, 32-bit cell scratch, the tile form of K1b
    "200k barcodes x 70k genes, 16-base UMIs": (dict(n=300_000, n_bar=210_000, n_gene=70_000, umi_len=16, rate_depth=0.8, dup_factor=3.0,
                                                     cell_dist="lognormal", gene_dist="zipf", p_unlisted_cb=0.05, p_bad_xf=0.1, p_n_umi=0.01, data_seed=71), 16, 71),
    "70k x 70k, 12 bases (61 bits) forced wide": (dict(n=200_000, n_bar=70_000, n_gene=70_000, umi_len=12, rate_cell=0.8, rate_depth=0.7, umi_pool=4096,
                                                       p_no_cb=0.03, p_unlisted_cb=0.1, p_n_umi=0.02, data_seed=72), 12, None),
    # LDS tables, deep groups: windows full of one group (giant_groups_kernel on the values), all duplicates, NULL UMIs only
    "huge groups": (dict(n=400_000, n_bar=3, n_gene=2, umi_len=12, data_seed=11), 12, None),
    "one group, all duplicates": (dict(n=100_000, n_bar=1, n_gene=1, umi_pool=1), 12, None),
    # one (cell, feature) group of a million reads: 652 hash partitions, more work items than reduce_hashed_kernel has threads
    # (items 512.. were never written)
    "one group of a million reads": (dict(n=1_000_000, n_bar=1, n_gene=1, umi_len=12, data_seed=13), 12, None),
    "one group, NULL UMIs": (dict(n=30_000, n_bar=1, n_gene=1, umi_pool=3, p_n_umi=0.9), 12, None),
    "mixed": (dict(n=250_000, n_bar=3000, n_gene=1500, rate_cell=0.5, rate_depth=0.5, umi_pool=64, p_no_cb=0.05, p_unlisted_cb=0.05,
                   p_bad_xf=0.15, p_n_umi=0.02, p_multi_gene=0.05, p_no_ub=0.02), 16, None),
    "tiny": (dict(n=37, n_bar=5, n_gene=3, umi_pool=4, p_n_umi=0.2), 16, None),
}


@pytest.mark.parametrize("name", list(WIDE_CASES))
def test_keys_wider_than_64_bits_match_oracle(name, monkeypatch):
    kw, umi_max, bits = WIDE_CASES[name]
    if bits is None:
        monkeypatch.setenv("FASTF_FORCE_WIDE_KEYS", "1")              # the wide path on lists whose keys would fit
    case = Case(**kw)
    ora = case.oracle()
    lists = case.lists()
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, umi_max_bases=umi_max)
    try:
        if bits is not None:
            assert eng.key_bits == bits > 64
        packed = case.packed(lists)
        h = len(packed[0]) // 3
        eng.push(*(a[:h] for a in packed))                            # (two pushes: the key store grows with its values)
        eng.push(*(a[h:] for a in packed))
        res = eng.finish()
        assert_matches_oracle(res, ora, eng, case, lists, eng.umi_rows())
        with pytest.raises(F.FastfError):                             # the device-level calls stay with 64-bit keys
            eng.probe_capacity(1000) or eng.dev_sort(0, 0, 0, 1)
    finally:
        eng.close()


@pytest.mark.parametrize("umi_len,kw", [
    (20, dict(n=200_000, n_bar=2000, n_gene=900, rate_cell=0.7, rate_depth=0.8, dup_factor=3.0, p_no_cb=0.03, p_unlisted_cb=0.05, p_bad_xf=0.1, p_n_umi=0.02)),
    (24, dict(n=120_000, n_bar=300, n_gene=100, umi_pool=256, p_n_umi=0.05)),
    (17, dict(n=60_000, n_bar=70_000, n_gene=70_000, rate_depth=0.9, umi_pool=4096)),
])
def test_umis_of_17_to_24_bases_match_oracle(umi_len, kw):
    """bam2db_ds.c:417-419 binds a blob of (len + 3) / 4 bytes for a UMI of any length: with umi_max_bases 24 the engine takes
    bases 17.. beside the packed record (fastf_batch_t.umi_ext) and carries 52 bits of UMI field beside the group word"""
    case = Case(umi_len=umi_len, **kw)
    ora = case.oracle()
    lists = case.lists()
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, umi_max_bases=24)
    try:
        cb, gx, umi, meta, ext = case.packed_long(lists)
        assert ext.any()
        eng.push(cb, gx, umi, meta, umi_ext=ext)
        res = eng.finish()
        assert_matches_oracle(res, ora, eng, case, lists, eng.umi_rows())
    finally:
        eng.close()


def test_mixed_umi_lengths_0_to_28_with_room_for_24():
    """lengths 0..28 in one run against an engine with room for 24 bases: the blob byte length and the zero padding decide
    equality (ACGT x 5 and ACGT x 5 + A: 5 and 6 bytes, different blobs; 21 and 24 bases of the same prefix padded with A: the
    same 6-byte blob); 25..28 bases (7 bytes) do not fit THIS engine and must raise the error bit, not a wrong count (an engine
    with room for 32 takes them: test_mixed_umi_lengths_0_to_32)"""
    from fastf_amd import synth
    from oracle import oracle as O
    rng = np.random.default_rng(4)
    n = 40_000
    bt, ft, bar, genes = synth.make_lists(40, 30, seed=5)
    flags, xf, cb, gx, _ = synth.make_records(n, bar, genes, seed=6)
    pool = [b"", b"ACGT", b"ACGTACGTACGTACGT", b"ACGTACGTACGTACGTA", b"ACGTACGTACGTACGTAAAA", b"ACGTACGTACGTACGTACGT", b"ACGTACGTACGTACGTACGTA",
            b"ACGTACGTACGTACGTACGTAAAA", b"ACGTACGTACGTACGTACGTACGT", b"TTTTTTTTTTTTTTTTTTTTTTTT", b"TTTTTTTTTTTTTTTTTTTTTTTN", b"ACGTACGTACGTACGTN",
            b"GGGGGGGGGGGGGGGGGGGGG", b"GGGGGGGGGGGGGGGGGGGGGA", b"GGGGGGGGGGGGGGGGC"]
    ub = np.array([pool[i] for i in rng.integers(0, len(pool), size=n)], dtype="S30")
    lists = F.Lists(bt, ft, 1.0, 926)
    ora = O.run_bam2db(bt, ft, flags, xf, synth.as_cstr(cb), synth.as_cstr(gx), synth.as_cstr(ub), 1.0, 1.0, 926, b"x.bam", True)
    eng = F.Engine.from_lists(lists, umi_max_bases=24)
    try:
        c, g, u, m, e = F.pack_records(lists, flags, xf, synth.as_cstr(cb), synth.as_cstr(gx), synth.as_cstr(ub), long_umis=True)
        eng.push(c, g, u, m, umi_ext=e)
        res = eng.finish()
        assert_matches_oracle(res, ora)
        assert eng.format_umi_rows(eng.umi_rows()) == ora["umi"]
        # a 25-base UMI (7 blob bytes) is beyond an engine created for 24: refused loudly
        eng.reset()
        ub2 = ub.copy(); ub2[0] = b"A" * 25
        flags2 = flags.copy(); flags2[0] |= 1 | 2 | 4 | 8; xf2 = xf.copy(); xf2[0] = 25
        cb2 = synth.as_cstr(cb).copy(); cb2[0] = bar[0]; gx2 = synth.as_cstr(gx).copy(); gx2[0] = genes[0]
        c, g, u, m, e = F.pack_records(lists, flags2, xf2, cb2, gx2, synth.as_cstr(ub2), long_umis=True)
        eng.push(c, g, u, m, umi_ext=e)
        with pytest.raises(F.FastfError) as ei:
            eng.finish()
        assert "umi_max_bases" in str(ei.value)
    finally:
        eng.close()


def test_mixed_umi_lengths_0_to_32():
    """the reference binds a UMI of any length (bam2db_ds.c:417-419); an engine with room for 32 bases takes every blob of up to 8
    bytes exactly: 28 bases and the same 28 + "A" are 7- and 8-byte blobs (different), 29 and 32 bases of one prefix padded with A
    the same 8-byte blob, 32 x A the all-zero 8-byte blob (counted: it is not NULL), UMIs that share their first 8 bases (one
    sub-group of the sorted word) and UMIs that differ there (several matrix rows on the device, summed on the host); N anywhere:
    NULL; 33 bases: refused loudly"""
    from fastf_amd import synth
    from oracle import oracle as O
    rng = np.random.default_rng(14)
    n = 60_000
    bt, ft, bar, genes = synth.make_lists(40, 30, seed=5)
    flags, xf, cb, gx, _ = synth.make_records(n, bar, genes, seed=6)
    p28 = b"ACGTTGCAACGTTGCAGGATCCTTAGGA"
    pool = [b"", b"ACGT", b"ACGTACGTACGTACGT", b"ACGTACGTACGTACGTA", b"ACGTACGTACGTACGTACGTAAAA", p28, p28 + b"A", p28 + b"AAAA", p28 + b"C", p28 + b"CAAA",
            p28 + b"CAAT", b"A" * 32, b"A" * 31, b"A" * 28, b"A" * 25, b"T" * 32, b"T" * 31 + b"N", b"T" * 29, b"N" + b"C" * 30,
            b"ACGTTGCAGGGGGGGGGGGGGGGGGGGGGGGGG"[:32], b"ACGTTGCATTTTTTTTTTTTTTTTTTTTTTTTT"[:32], b"ACGTTGCA" + b"C" * 17, b"GCGTTGCA" + b"C" * 17, b"GCGTTGCA" + b"C" * 18]
    pool += [bytes(rng.choice(list(b"ACGT"), size=int(l))) for l in rng.integers(25, 33, size=200)]
    ub = np.array([pool[i] for i in rng.integers(0, len(pool), size=n)], dtype="S40")
    lists = F.Lists(bt, ft, 1.0, 926)
    ora = O.run_bam2db(bt, ft, flags, xf, synth.as_cstr(cb), synth.as_cstr(gx), synth.as_cstr(ub), 1.0, 1.0, 926, b"x.bam", True)
    eng = F.Engine.from_lists(lists, umi_max_bases=32)
    try:
        c, g, u, m, e = F.pack_records(lists, flags, xf, synth.as_cstr(cb), synth.as_cstr(gx), synth.as_cstr(ub), long_umis=True)
        for a in range(0, n, 25_000):                            # several pushes: the sub-groups of one (cell, feature) meet in the sort
            eng.push(c[a:a + 25_000], g[a:a + 25_000], u[a:a + 25_000], m[a:a + 25_000], umi_ext=e[a:a + 25_000])
        res = eng.finish()
        assert_matches_oracle(res, ora)
        assert eng.format_umi_rows(eng.umi_rows()) == ora["umi"]
        # 33 bases (9 blob bytes) are beyond what a batch carries: refused loudly
        eng.reset()
        ub2 = ub.copy(); ub2[0] = b"A" * 33
        flags2 = flags.copy(); flags2[0] |= 1 | 2 | 4 | 8; xf2 = xf.copy(); xf2[0] = 25
        cb2 = synth.as_cstr(cb).copy(); cb2[0] = bar[0]; gx2 = synth.as_cstr(gx).copy(); gx2[0] = genes[0]
        c, g, u, m, e = F.pack_records(lists, flags2, xf2, cb2, gx2, synth.as_cstr(ub2), long_umis=True)
        eng.push(c, g, u, m, umi_ext=e)
        with pytest.raises(F.FastfError) as ei:
            eng.finish()
        assert "umi_max_bases" in str(ei.value)
    finally:
        eng.close()


def test_wide_keys_are_refused_by_the_64_bit_device_calls_only():
    """a sharded engine may have keys wider than 64 bits (tests/test_gpu_dist.py::test_sharded_pass_with_keys_wider_than_64_bits,
    tests/test_gpu_multi.py::test_multi_device_wide_keys_match_oracle); the fastf_dev_* calls that move ONE word per key refuse it
    with a message, and the wide calls refuse an engine whose keys fit"""
    case = Case(n=10, n_bar=70_000, n_gene=70_000)
    lists = case.lists()
    eng = F.Engine.from_lists(lists, umi_max_bases=16, n_shards=2, shard_rank=0)
    try:
        assert eng.wide and eng.key_bits > 64
        with pytest.raises(F.FastfError) as ei:
            eng.dev_probe_pack(0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0)
        assert "64 bits" in str(ei.value)
    finally:
        eng.close()
    small = Case(n=10, n_bar=20, n_gene=10)
    eng = F.Engine.from_lists(small.lists(), umi_max_bases=12)
    try:
        assert not eng.wide
        with pytest.raises(F.FastfError):
            eng.dev_adopt_wide(0, 0, 0)
    finally:
        eng.close()


# ---- the resident single-GPU pass (what bench.py times): streaming K1b -> segmented keys -> sort through the region map ----
STREAM_CASES = {
    "tiny": dict(n=100, n_bar=20, n_gene=10),
    "below one unit": dict(n=255, n_bar=20, n_gene=10, rate_depth=0.5),
    "ragged tail": dict(n=4096 * 3 + 257, n_bar=300, n_gene=80, rate_cell=0.5, rate_depth=0.5, p_unlisted_cb=0.2, p_bad_xf=0.2, p_n_umi=0.02),
    "keep all": dict(n=300_000, n_bar=2000, n_gene=900, umi_pool=512),
    "c3 shape": dict(n=700_000, n_bar=40_000, n_gene=5000, rate_cell=0.5, rate_depth=0.5, umi_len=12, dup_factor=4.0,
                     cell_dist="lognormal", gene_dist="zipf", p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.15, p_n_umi=0.001),
    "nothing survives": dict(n=50_000, n_bar=100, n_gene=50, rate_depth=0.0),
    "no hits": dict(n=30_000, n_bar=100, n_gene=50, p_unlisted_cb=1.0),
    "sparse gene ids": dict(n=200_000, n_bar=500, n_gene=3000, gene_stride=8, rate_depth=0.7, umi_pool=128),
    # the other instantiations of the streaming kernel: 32-bit cell scratch (more than 65 535 cells: the barcode table is the L2 one),
    # gene images that leave room for one workgroup per CU only — the bitmap form and the dense table
    "u32 cell scratch": dict(n=400_000, n_bar=70_000, n_gene=400, umi_pool=1024, rate_depth=0.6, p_unlisted_cb=0.1, data_seed=21),
    "one workgroup per CU, bitmap image": dict(n=250_000, n_bar=1500, n_gene=30_000, gene_stride=8, umi_pool=2048, rate_depth=0.8, p_bad_xf=0.1),
    "one workgroup per CU, dense table": dict(n=250_000, n_bar=1500, n_gene=60_000, gene_dist="zipf", umi_pool=2048, rate_cell=0.5),
}


@pytest.mark.parametrize("name", list(STREAM_CASES))
def test_resident_pass_streaming_k1b_matches_oracle(name):
    """fastf_dev_probe_pack(FASTF_PROBE_SEGMENTED [| FASTF_PROBE_BLOCKED]) -> fastf_dev_sort(FASTF_SORT_SEGMENTED) ->
    fastf_dev_reduce through ShardedPass (G = 1), two steps on the same buffers, against the oracle: from the blocked
    record layout, from the SoA arrays, and with the tile form of K1b forced"""
    import torch
    from fastf_amd.dist import HipStages, ShardedPass
    case = Case(**STREAM_CASES[name])
    ora = case.oracle()
    lists = case.lists()
    cbk, gxk, umi, meta = case.packed(lists)
    dev = torch.device("cuda", 0)
    t = lambda x: hostmem.to_device(x, dev)
    d = [t(x) for x in (cbk, gxk, umi, meta)]
    draws = t(F.mt_draws(case.seed, lists.mt_skip, case.n))
    for form in ("blocked", "soa", "tile"):
        # blocked: the engine's staging layout (one run of gx | umi | meta | cell scratch per 256-record unit, K1a fills the
        # scratch slice); soa: the four arrays of the ABI; tile: the tile form of K1b forced
        if form == "tile":
            os.environ["FASTF_NO_STREAM_K1B"] = "1"
        if os.environ.get("FASTF_TEST_TRACE"):
            print("[form] %s" % form, file=sys.stderr, flush=True)
        try:
            eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, umi_max_bases=12)
            st = HipStages(eng, dev)
            sp = ShardedPass(st, case.n, dev)
            lds_genes = "LDS" in eng.table_modes.split("genes:")[1]
            blk = st.block(d[1], d[2], d[3], case.n) if form == "blocked" else None
            assert (blk is not None) == (form == "blocked" and lds_genes)
            if form == "blocked" and blk is None:
                eng.close()
                continue
            # (first step: the 32-bit draws, turned into decisions by the call; second: the decision bits made once)
            for dr in (draws, sp.prepare_draws(draws)):
                if blk is not None:
                    sp.run(d[0], blk, None, None, case.n, dr)
                else:
                    sp.run(d[0], d[1], d[2], d[3], case.n, dr)
            f, c, k = sp.local_coo()
            hits, sampled, valid, err = sp.global_counters()
            assert err == 0
            assert sp.st.segmented == (form != "tile" and lds_genes)
            assert (sampled, valid) == (ora["sampled"], ora["valid"])
            assert len(f) == ora["nnz"]
            np.testing.assert_array_equal(c, ora["cell"].astype(np.int64))
            np.testing.assert_array_equal(f, ora["feature"].astype(np.int64))
            np.testing.assert_array_equal(k, ora["count"].astype(np.int64))
            eng.close()
        finally:
            os.environ.pop("FASTF_NO_STREAM_K1B", None)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 256, 257, 511, 4095, 4096, 4097, 12 * 256 - 1, 12 * 256, 12 * 256 + 1, 24 * 256 + 3,
                               16 * 4096 - 255, 16 * 4096 + 1])
def test_streaming_k1b_record_counts_around_its_unit_tile_and_round_sizes(n):
    """the hand-pipelined loop reads a unit's records one unit ahead, unconditionally — a unit that does not exist reads the last one
    that does, a lane beyond the records its unit's last record: record counts around a unit (256), a tile (4096) and one round
    of a 12-wave workgroup, blocked and SoA, every record a hit (the decision words of the last unit end the stream)"""
    import torch
    from fastf_amd.dist import HipStages, ShardedPass
    case = Case(n=n, n_bar=64, n_gene=40, rate_depth=0.5, umi_pool=16, p_no_cb=0.0, p_unlisted_cb=0.0)
    ora = case.oracle()
    lists = case.lists()
    cbk, gxk, umi, meta = case.packed(lists)
    dev = torch.device("cuda", 0)
    t = lambda x: hostmem.to_device(x, dev)
    d = [t(x) for x in (cbk, gxk, umi, meta)]
    draws = t(F.mt_draws(case.seed, lists.mt_skip, ora["total"]))       # exactly as many draws as there can be hits
    for form in ("blocked", "soa"):
        eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, umi_max_bases=12)
        try:
            st = HipStages(eng, dev)
            sp = ShardedPass(st, case.n, dev)
            blk = st.block(d[1], d[2], d[3], case.n) if form == "blocked" else None
            assert (blk is not None) == (form == "blocked")
            dr = sp.prepare_draws(draws)
            for _ in range(2):
                if blk is not None:
                    sp.run(d[0], blk, None, None, case.n, dr)
                else:
                    sp.run(d[0], d[1], d[2], d[3], case.n, dr)
            f, c, k = sp.local_coo()
            hits, sampled, valid, err = sp.global_counters()
            assert err == 0 and sp.st.segmented
            assert (sampled, valid) == (ora["sampled"], ora["valid"])
            np.testing.assert_array_equal(c, ora["cell"].astype(np.int64))
            np.testing.assert_array_equal(f, ora["feature"].astype(np.int64))
            np.testing.assert_array_equal(k, ora["count"].astype(np.int64))
        finally:
            eng.close()


def test_finish_without_copy_returns_views_of_the_engines_rows():
    """Engine.finish(copy=False): what a C caller of fastf_engine_finish gets — arrays that live in the engine's row buffer"""
    case = Case(n=60_000, n_bar=300, n_gene=100, umi_pool=64, rate_depth=0.7)
    lists = case.lists()
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed)
    try:
        eng.push(*case.packed(lists))
        a = eng.finish(copy=False)
        b = eng.finish()                                   # idempotent, copied
        assert not a["count"].flags["OWNDATA"] and b["count"].flags["OWNDATA"]
        for k in ("feature", "cell", "count"):
            np.testing.assert_array_equal(a[k], b[k])
        assert_matches_oracle(b, case.oracle())
    finally:
        eng.close()


def test_rows_on_loan_in_the_callers_pinned_memory():
    """fastf_engine_lend_rows: finish() writes the rows into pinned memory of the caller's when the matrix fits (bam2db()
    lends a decoder slot), into its own buffer when it does not; a reset ends the loan."""
    case = Case(n=80_000, n_bar=400, n_gene=120, umi_pool=64, rate_depth=0.8, p_n_umi=0.01)
    lists = case.lists()
    packed = case.packed(lists)
    ora = case.oracle()
    pb = F.PinnedBatch(40_000)                             # 960 000 bytes of pinned memory: 80 000 rows
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed)
    try:
        lo, hi = pb._p, pb._p + pb.n * 24
        eng.push(*packed)
        eng.lend_rows(pb._p, pb.n * 24)
        a = eng.finish(copy=False)
        assert 0 < a["nnz"] <= pb.n * 2
        for k in ("feature", "cell", "count"):
            assert lo <= a[k].ctypes.data < hi, k          # the rows lie in the loan
        assert_matches_oracle(eng.finish(), ora)
        # a loan that is too small for the matrix is left alone
        eng.reset(); eng.reseed(case.seed, lists.mt_skip)
        eng.push(*packed)
        pb.cb_key[:] = 0x5A5A5A5A5A5A5A5A
        eng.lend_rows(pb._p, 12 * (a["nnz"] - 1))
        b = eng.finish(copy=False)
        assert not (lo <= b["count"].ctypes.data < hi)
        assert np.all(pb.cb_key == 0x5A5A5A5A5A5A5A5A)
        assert_matches_oracle(eng.finish(), ora)
        # the loan ended with the reset: the next pass uses the engine's buffer again
        eng.reset(); eng.reseed(case.seed, lists.mt_skip)
        eng.lend_rows(pb._p, pb.n * 24)
        eng.reset(); eng.reseed(case.seed, lists.mt_skip)
        eng.push(*packed)
        c = eng.finish(copy=False)
        assert not (lo <= c["count"].ctypes.data < hi)
        with pytest.raises(F.FastfError):
            eng.lend_rows(pb._p, pb.n * 24)                # the rows of this pass are out already
        assert_matches_oracle(eng.finish(), ora)
    finally:
        eng.close()
        pb.close()


@pytest.mark.parametrize("kw", [
    dict(n=300_000, n_bar=2000, n_gene=900, rate_cell=0.7, rate_depth=0.6, umi_pool=512, p_unlisted_cb=0.05, p_bad_xf=0.1, p_n_umi=0.01),
    dict(n=90_000, n_bar=30, n_gene=20, umi_pool=64),                       # deep groups: one sort pass is not all there is
])
def test_push_after_finish_continues_the_job(kw):
    """finish() in the middle of a job, then more pushes and another finish (no reset): the second result is the whole job's.
    In stream mode the sort passes of the first finish have rewritten the key store, so the keys so far become regions again
    (rebase_store: runs of 64 K keys, their first-digit histograms counted by a kernel of their own) before the new chunks'
    regions are appended behind them"""
    case = Case(**kw)
    ora = case.oracle()
    lists = case.lists()
    eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed)
    try:
        packed = case.packed(lists)
        a, b = case.n // 3, 2 * case.n // 3
        eng.push(*(x[:a] for x in packed))
        first = eng.finish()
        assert first["total"] == a
        eng.push(*(x[a:b] for x in packed))
        eng.finish(); eng.umi_rows()                                        # (a full sort in between as well)
        eng.push(*(x[b:] for x in packed))
        res = eng.finish()
        assert_matches_oracle(res, ora, eng, case, lists, eng.umi_rows())
    finally:
        eng.close()
