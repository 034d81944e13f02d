"""Pins of the CPU oracle: reference known answers, the reference's own
libc-only sources (oracle/_ref), numpy's independent MT19937, and SQLite itself."""
import ctypes as C
import json
import os
import sqlite3

import numpy as np
import pytest

from oracle import oracle as O
from helpers import Case

FX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_8c.json")))


def test_mt_known_answers():
    assert O.mt_stream(926, 8).tolist() == FX["mt19937_seed_926_first8"]


@pytest.mark.parametrize("seed", [0, 1, 926, 5489, 0xFFFFFFFF])
def test_mt_matches_numpy_randomstate(seed):
    want = np.random.RandomState(seed)._bit_generator.random_raw(3000).astype(np.uint32)
    np.testing.assert_array_equal(O.mt_stream(seed, 3000), want)


def test_sample_int_known_answer():
    k = FX["sample_int"]
    assert O.sample_cells(k["n_total"], k["n_sample"], k["seed"]).tolist() == k["sorted"]


def test_depth_thresholds_known_answers():
    # T(rate) = first draw that the reference drops
    L = O.lib()
    for rate, T in FX["thresholds"].items():
        r = float(rate)
        assert L.oracle_keep_draw(T - 1, C.c_float(r)) == 1
        assert L.oracle_keep_draw(min(T, 0xFFFFFFFF), C.c_float(r)) == 0


def test_edge_case_fixture():
    fx = FX["edge_case"]
    recs = fx["records"]
    n = len(recs)
    r = O.run_bam2db(fx["barcodes"].encode(), fx["features"].encode(), np.full(n, 15, np.uint8), np.full(n, 25, np.int32),
                     np.array([x[0].encode() for x in recs], dtype="S8"), np.array([x[1].encode() for x in recs], dtype="S8"),
                     np.array([x[2].encode() for x in recs], dtype="S16"), 1.0, 1.0, 926, b"x.bam", True)
    assert [r["total"], r["sampled"], r["valid"]] == fx["counters"]
    lines = r["matrix"].decode().splitlines()
    assert lines[0] == "%%MatrixMarket matrix coordinate integer general"
    assert lines[1] == "%metadata_json: "
    assert lines[12] == fx["dims"]
    assert lines[13:] == fx["matrix_rows"]
    assert r["umi"].decode().splitlines() == fx["umi_rows"]


# ---------------- the reference's own sources (compiled in place by `make -C oracle ref`) ----------------
ref = O.ref_lib()
needs_ref = pytest.mark.skipif(ref is None, reason="oracle/_ref not built (reference sources absent)")


@needs_ref
@pytest.mark.parametrize("seed", [926, 1, 20240501])
def test_mt_matches_reference_sources(seed):
    ref.init_genrand(seed)
    want = [ref.genrand_int32() & 0xFFFFFFFF for _ in range(2000)]
    assert O.mt_stream(seed, 2000).tolist() == want
    ref.init_genrand(seed)
    L = O.lib()
    L.oracle_init_genrand(seed)
    for _ in range(100):
        assert L.oracle_genrand_real1() == ref.genrand_real1()


@needs_ref
@pytest.mark.parametrize("n,k,seed", [(6, 3, 926), (1000, 500, 926), (1000, 1000, 7), (1000, 0, 7), (17, 16, 3), (50000, 12345, 11)])
def test_sample_int_matches_reference_sources(n, k, seed):
    seq = ref.GetSeqInt(0, n - 1, 1)
    out = ref.SampleInt(seq, n, k, 0, seed)
    want = sorted(out[i] for i in range(k))
    assert O.sample_cells(n, k, seed).tolist() == want
    # the generator state after sampling is part of the contract (draws follow in the record loop)
    nxt_ref = ref.genrand_int32() & 0xFFFFFFFF
    O.sample_cells(n, k, seed)
    assert O.lib().oracle_genrand_int32() == nxt_ref


@needs_ref
def test_hashtable_first_wins_matches_reference_sources():
    HF = C.CFUNCTYPE(C.c_uint64, C.c_char_p, C.c_size_t)
    hf = HF(lambda s, n: O.lib().oracle_djb2(s, n))
    ht = ref.hash_table_create(1 << 20, C.cast(hf, C.c_void_p), None)
    vals = (C.c_size_t * 4)(1, 2, 3, 4)
    keys = [b"AAAA-1", b"CCCC-1", b"AAAA-1", b"GGGG-1"]
    ins = [ref.hash_table_insert(ht, k, C.addressof(vals) + 8 * i) for i, k in enumerate(keys)]
    assert ins == [True, True, False, True]            # duplicate refused, first value kept
    p = ref.hash_table_lookup(ht, b"AAAA-1")
    assert C.cast(p, C.POINTER(C.c_size_t))[0] == 1
    assert ref.hash_table_lookup(ht, b"TTTT-1") is None
    assert ref.hash_table_lookup(ht, b"AAAA") is None   # exact match, not prefix


# ---------------- SQLite: the reference's aggregate statement on the oracle's own rows ----------------
AGG = ("SELECT feature_index, cell_index, COUNT(DISTINCT encoded_umi) AS expression_level "
       " FROM umi GROUP BY cell_index, feature_index;")
NUMI = ("SELECT feature_index, cell_index, encoded_umi, COUNT(*) AS n_copy "
        " FROM umi GROUP BY cell_index, feature_index, encoded_umi;")


@pytest.mark.parametrize("kw", [
    dict(n=20000, n_bar=50, n_gene=30, umi_pool=32, p_n_umi=0.05),
    dict(n=30000, n_bar=200, n_gene=100, umi_len=12, dup_factor=3.0, p_n_umi=0.01, rate_depth=0.5, rate_cell=0.5),
])
def test_aggregate_matches_sqlite(kw):
    case = Case(**kw)
    r = O.run_bam2db(case.bt, case.ft, case.flags, case.xf, case.cb, case.gx, case.ub, case.rate_cell, case.rate_depth,
                     case.seed, case.label, True, want_rows=True)
    db = sqlite3.connect(":memory:")
    db.execute("CREATE TABLE umi (cell_index INTEGER, feature_index INTEGER, encoded_umi TEXT);")
    rows = []
    for c, f, ln, blob in zip(r["row_cell"], r["row_feature"], r["row_blob_len"], r["row_blob"]):
        rows.append((int(c), int(f), None if ln < 0 else bytes(blob[:ln])))
    db.executemany("INSERT INTO umi VALUES (?1, ?2, ?3);", rows)
    got = db.execute(AGG).fetchall()
    assert [tuple(x) for x in got] == list(zip(r["feature"].tolist(), r["cell"].tolist(), r["count"].tolist()))
    got_u = db.execute(NUMI).fetchall()
    assert len(got_u) == r["n_umi_rows"]
    assert [(a, b, d) for a, b, _, d in got_u] == list(zip(r["umi_feature"].tolist(), r["umi_cell"].tolist(), r["umi_ncopy"].tolist()))
    for (_, _, blob, _), txt in zip(got_u, r["umi_text"]):
        assert (blob is None) == (txt == b"NULL")


def test_mixed_length_blobs_collapse_like_sqlite():
    """10/11/12-bp UMIs that agree after zero padding are one blob; 13 bp is another"""
    for a, b, same in [(b"ACGTACGTAC", b"ACGTACGTACA", True), (b"ACGTACGTAC", b"ACGTACGTACAA", True),
                       (b"ACGTACGTAC", b"ACGTACGTACAAA", False), (b"ACGTACGTAC", b"ACGTACGTACC", False)]:
        assert (O.encode_dna(a) == O.encode_dna(b)) == same
    assert O.encode_dna(b"ACGTN") is None
    assert O.encode_dna(b"") == b""


def test_config0_digest_fixture():
    """BASELINE configs[0] through the oracle: counters, row count and output digests are stable (tests/golden)"""
    import hashlib
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config0_digest.json")))
    for name, want in fx.items():
        o = Case(**want["kw"]).oracle()
        assert [o["total"], o["sampled"], o["valid"]] == want["counters"] and o["nnz"] == want["nnz"]
        assert hashlib.md5(o["matrix"]).hexdigest() == want["matrix_md5"]
        assert hashlib.md5(o["barcodes"]).hexdigest() == want["barcodes_md5"]
        assert hashlib.md5(o["features"]).hexdigest() == want["features_md5"]
        assert hashlib.md5(o["umi"]).hexdigest() == want["umi_md5"]
