"""crb / extract on the GPU: the tag-histogram seam against numpy, the commands against the oracle
(oracle/fastf_oracle_tags.c, pinned against the reference's own tree code), the CLI, and the read_bam drop-in."""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np
import pytest

import fastf_amd as F
from fastf_amd import _lib
from fastf_amd.tags import KeyDict, TagHist, crb_text, extract_text
from oracle import oracle as O
from tag_helpers import TagCase

pytestmark = pytest.mark.gpu


def _np_hist(keys):
    keys = np.asarray(keys, dtype=np.uint64)
    idx = np.nonzero(keys)[0]
    u, first, cnt = np.unique(keys[idx], return_index=True, return_counts=True)
    return u, cnt.astype(np.uint64), idx[first].astype(np.uint64)


@pytest.mark.parametrize("n,pool,p_absent,seed", [(1, 1, 0.0, 0), (5000, 1, 0.3, 1), (200_000, 37, 0.1, 2),
                                                  (300_000, 150_000, 0.0, 3), (1_000_000, 5000, 0.5, 4), (10_000, 100, 1.0, 5)])
def test_single_tag_histogram_matches_numpy(n, pool, p_absent, seed):
    rng = np.random.default_rng(seed)
    vals = rng.integers(1, 1 << 62, pool, dtype=np.uint64) | (np.uint64(1) << np.uint64(62))
    keys = vals[rng.integers(0, pool, n)]
    keys[rng.random(n) < p_absent] = 0
    h = TagHist()
    try:
        cut = n // 3
        h.push(keys[:cut]); h.push(keys[cut:])
        r = h.finish()
    finally:
        h.close()
    u, cnt, first = _np_hist(keys)
    assert (r["n_records"], r["n_valid"]) == (n, int((keys != 0).sum()))
    np.testing.assert_array_equal(r["key1"], u)
    np.testing.assert_array_equal(r["count1"], cnt)
    np.testing.assert_array_equal(r["first1"], first)


@pytest.mark.parametrize("n,p1,p2,seed", [(3000, 5, 9, 1), (400_000, 900, 5000, 2), (250_000, 40_000, 3, 3)])
def test_pair_histogram_matches_numpy(n, p1, p2, seed):
    rng = np.random.default_rng(seed)
    v1 = rng.integers(1, 1 << 63, p1, dtype=np.uint64)
    v2 = rng.integers(1, 1 << 63, p2, dtype=np.uint64)
    k1, k2 = v1[rng.integers(0, p1, n)], v2[rng.integers(0, p2, n)]
    k1[rng.random(n) < 0.07] = 0
    k2[rng.random(n) < 0.05] = 0
    h = TagHist()
    try:
        h.push(k1[:n // 2], k2[:n // 2]); h.push(k1[n // 2:], k2[n // 2:])
        r = h.finish()
    finally:
        h.close()
    valid = (k1 != 0) & (k2 != 0)
    idx = np.nonzero(valid)[0]
    pairs = np.stack([k1[idx], k2[idx]], axis=1)
    u, first, cnt = np.unique(pairs, axis=0, return_index=True, return_counts=True)
    assert r["n_valid"] == int(valid.sum())
    np.testing.assert_array_equal(r["key1"][r["pair_k1"]], u[:, 0])
    np.testing.assert_array_equal(r["pair_key2"], u[:, 1])
    np.testing.assert_array_equal(r["pair_count"], cnt.astype(np.uint64))
    np.testing.assert_array_equal(r["pair_first"], idx[first].astype(np.uint64))
    # level 1 follows from the pairs
    u1, f1, c1 = np.unique(k1[idx], return_index=True, return_counts=True)
    live = r["count1"] > 0
    np.testing.assert_array_equal(r["key1"][live], u1)
    np.testing.assert_array_equal(r["count1"][live], c1.astype(np.uint64))
    np.testing.assert_array_equal(r["first1"][live], idx[f1].astype(np.uint64))


CASES = {
    "small": dict(n=3000, n_cb=30, seed=1),
    "medium": dict(n=120_000, n_cb=2000, seed=2, p_err=0.1, p_n=0.03),
    "sorted_cb": dict(n=20_000, n_cb=300, seed=3, sorted_cb=True),          # degenerate (list-shaped) CB tree
    "one_cb": dict(n=5000, n_cb=1, seed=4, p_no_cb=0.0),
}


@pytest.fixture(scope="module")
def bams(tmp_path_factory):
    d = tmp_path_factory.mktemp("tagbams")
    out = {}
    for name, kw in CASES.items():
        c = TagCase(**kw)
        p = str(d / (name + ".bam"))
        c.write(p)
        out[name] = (c, p)
    return out


@pytest.mark.parametrize("name", list(CASES))
def test_crb_text_equals_oracle(bams, name):
    c, path = bams[name]
    want, n_read, undef = O.run_crb(c.has_cb, np.ones(c.n), c.S(c.cb), c.S(c.cr))
    got, n = crb_text(path)
    assert undef == 0 and n == n_read == c.n
    assert got == want


@pytest.mark.parametrize("name", ["small", "medium"])
@pytest.mark.parametrize("tag", [b"CB", b"CR", b"UB", b"GX", b"GN"])
def test_extract_string_tag_equals_oracle(bams, name, tag):
    c, path = bams[name]
    vals = {b"CB": c.cb, b"CR": c.cr, b"UB": c.ub, b"GX": c.gx, b"GN": c.gn}[tag]
    present = {b"CB": c.has_cb, b"GX": c.has_gx, b"GN": c.has_gx}.get(tag, np.ones(c.n))
    want, total, valid = O.run_extract(present, vals=c.S(vals))
    got, n, nv = extract_text(path, tag, 0)
    assert got == want
    assert (2 * n, nv) == (total, valid)


@pytest.mark.parametrize("tag", [b"xf", b"NH"])
def test_extract_integer_tag_equals_oracle(bams, tag):
    c, path = bams["medium"]
    iv = c.xf if tag == b"xf" else c.nh
    want, total, valid = O.run_extract(np.ones(c.n), ivals=iv)
    got, n, nv = extract_text(path, tag, 1)
    assert got == want and (2 * n, nv) == (total, valid)


def test_extract_absent_tag_gives_empty_summary(bams):
    c, path = bams["small"]
    got, n, nv = extract_text(path, b"ZZ", 0)
    assert got == b"" and (n, nv) == (c.n, 0)


def test_cli_crb_and_extract(bams, tmp_path):
    c, path = bams["medium"]
    out = tmp_path / "crb.tsv.gz"
    r = subprocess.run([_lib.cli_path(), "crb", "-b", path, "--out=%s" % out], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    want, _, _ = O.run_crb(c.has_cb, np.ones(c.n), c.S(c.cb), c.S(c.cr))
    assert gzip.decompress(out.read_bytes()) == want
    assert r.stdout.splitlines() == ["Processed all %d reads" % c.n, "Writing to file...", "Done."]      # extract.c:125, main.c:272,284
    r = subprocess.run([_lib.cli_path(), "extract", "--bam", path, "-tUB", "-T", "0"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    want, total, valid = O.run_extract(np.ones(c.n), vals=c.S(c.ub))
    assert (tmp_path / "tag_summary.csv").read_bytes() == want
    assert r.stdout.splitlines() == ["Processed all %d reads" % total, "Valid reads: %d" % valid]          # extract.c:206-207
    r = subprocess.run([_lib.cli_path(), "extract", "-b", "/nonexistent.bam", "-t", "CB"], capture_output=True, text=True)
    assert r.returncode == 1 and "does not exist" in r.stderr                                              # main.c:386
    r = subprocess.run([_lib.cli_path(), "extract", "-b", path], capture_output=True, text=True)
    assert r.returncode == 1 and "--tag is required" in r.stderr                                           # main.c:392


def test_read_bam_dropin_tree_prints_like_the_reference(bams, tmp_path):
    """read_bam() returns the reference's CB_node tree; printed by the exported print_CB_node through zlib's gzFile"""
    c, path = bams["small"]
    L = F.lib()
    z = C.CDLL("libz.so.1")
    z.gzopen.restype = C.c_void_p
    z.gzopen.argtypes = [C.c_char_p, C.c_char_p]
    z.gzclose.argtypes = [C.c_void_p]
    root = L.read_bam(path.encode())
    out = str(tmp_path / "tree.gz").encode()
    fp = z.gzopen(out, b"w")
    L.print_CB_node(root, fp)
    z.gzclose(fp)
    L.free_CB_node(root)
    want, _, _ = O.run_crb(c.has_cb, np.ones(c.n), c.S(c.cb), c.S(c.cr))
    assert gzip.decompress(open(out, "rb").read()) == want


def test_large_bam(tmp_path):
    """a BAM of 800 k records: crb + extract of two tags vs the oracle"""
    c = TagCase(n=800_000, n_cb=8000, seed=11, umi_pool=200_000, n_gene=5000)
    path = str(tmp_path / "big.bam")
    c.write(path)
    want, _, _ = O.run_crb(c.has_cb, np.ones(c.n), c.S(c.cb), c.S(c.cr))
    got, n = crb_text(path)
    assert n == c.n and got == want
    for tag, vals in ((b"CR", c.cr), (b"UB", c.ub)):
        want, total, valid = O.run_extract(np.ones(c.n), vals=c.S(vals))
        got, n, nv = extract_text(path, tag, 0)
        assert got == want and nv == valid


def test_records_the_reference_cannot_handle_are_skipped_not_fatal(tmp_path):
    """CB without CR is a NULL dereference in the reference (extract.c:102-103); an integer-typed tag read as a string
    is one too (extract.c:189); -T values other than 0/1 fall through its switch.  Defined behaviour here: skip + warn."""
    from fastf_amd import synth
    n = 5000
    rng = np.random.default_rng(8)
    cbs = [b"ACGTACGTACGTAC%s-1" % (b"AC", b"GT", b"TT")[i % 3] for i in range(n)]
    crs = [b"ACGTACGTACGTAC%s" % (b"AC", b"GT", b"NN")[int(x)] for x in rng.integers(0, 3, n)]
    has_cr = rng.random(n) > 0.2
    z = np.zeros(n, dtype=np.uint8); e = np.array([b""] * n, dtype="S1")

    def aux(i):
        a = synth.aux_Z(b"CB", cbs[i]) + synth.aux_int(b"xf", 25, b"C")
        if has_cr[i]:
            a += synth.aux_Z(b"CR", crs[i])
        return a
    path = str(tmp_path / "u.bam")
    synth.write_bam(path, z, z.astype(np.int32), e, e, e, extra_aux=aux)
    want, n_read, undef = O.run_crb(np.ones(n), has_cr, TagCase.S(cbs), TagCase.S(crs))
    got, nr = crb_text(path)
    assert nr == n_read == n and undef == int((~has_cr).sum())
    assert got == want
    # string mode on the integer tag xf: every record is "valid" (the tag exists) but nothing can be inserted
    got, nr, nv = extract_text(path, b"xf", 0)
    assert got == b"" and (nr, nv) == (n, n)
    # -T 7: the reference's switch has no such case → counts only
    got, nr, nv = extract_text(path, b"CB", 7)
    assert got == b"" and (nr, nv) == (n, n)
