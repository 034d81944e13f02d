"""Pins of the crb / extract oracle (oracle/fastf_oracle_tags.c): the reference's own filter.c tree code
(oracle/_ref/libfastf_ref_tree.so: insert_tree, print_tree, print_tree_same_row) run on the same strings, plus
hand-checked known answers and a golden fixture generated from that reference code (tests/golden/tag_tree.json,
made by tests/golden/make_tag_tree.py in the build container)."""
import ctypes as C
import gzip
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(__file__)


def _S(strings, width=None):
    w = (width or max([len(s) for s in strings] + [1])) + 1
    return np.array(strings, dtype="S%d" % w)


def _ref_print_tree(R, strings):
    """the reference's insert_tree + print_tree (filter.c:105-148) through a libc FILE*"""
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    import tempfile
    root = None
    for s in strings:
        root = R.insert_tree(root, s)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.csv").encode()
        fp = libc.fopen(path, b"w")
        R.print_tree(root, fp)
        libc.fclose(fp)
        txt = open(path, "rb").read()
    R.free_tree_node(root)
    return txt


def _ref_same_row(R, strings):
    """insert_tree + print_tree_same_row (filter.c:160-169) through zlib's gzFile"""
    z = C.CDLL("libz.so.1")
    z.gzopen.restype = C.c_void_p
    z.gzopen.argtypes = [C.c_char_p, C.c_char_p]
    z.gzclose.argtypes = [C.c_void_p]
    import tempfile
    root = None
    for s in strings:
        root = R.insert_tree(root, s)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.gz").encode()
        fp = z.gzopen(path, b"w")
        R.print_tree_same_row(root, fp)
        z.gzclose(fp)
        txt = gzip.decompress(open(path, "rb").read())
    R.free_tree_node(root)
    return txt


def _random_tags(rng, n, pool, alphabet=b"ACGTN", length=12):
    vals = [bytes(rng.choice(list(alphabet), size=int(rng.integers(1, length + 1))).astype(np.uint8)) for _ in range(pool)]
    return [vals[i] for i in rng.integers(0, pool, n)]


def test_extract_known_answer():
    tags = [b"G", b"C", b"T", b"A", b"C", b"G", b"G", b"D"]
    txt, total, valid = O.run_extract(np.ones(8), vals=_S(tags))
    # insertion-order BST: G root; C left; T right; A left of C; D right of C.  Pre-order: G C A D T
    assert txt == b"G,3\nC,2\nA,1\nD,1\nT,1\n"
    assert (total, valid) == (16, 8)                      # total_count is incremented twice per record (extract.c:163,165)


def test_extract_integer_type_orders_by_decimal_string():
    iv = np.array([25, 17, 100, 25, -3, 9], dtype=np.int64)
    txt, total, valid = O.run_extract(np.array([1, 1, 1, 1, 1, 0]), ivals=iv)
    # strcmp order of "%d" strings: "-3" < "100" < "17" < "25"; root 25, then 17 (left), 100 (left of 17), -3 (left of 100)
    assert txt == b"25,2\n17,1\n100,1\n-3,1\n"
    assert (total, valid) == (12, 5)


def test_crb_known_answer():
    cb = [b"CC-1", b"AA-1", b"CC-1", b"TT-1", b"AA-1", b"CC-1", b""]
    cr = [b"CC", b"AA", b"CA", b"TT", b"AA", b"CC", b"GG"]
    has_cb = [1, 1, 1, 1, 1, 1, 0]
    txt, n, undef = O.run_crb(has_cb, np.ones(7), _S(cb), _S(cr))
    assert txt == b"CC-1;CC,2;CA,1;\nAA-1;AA,2;\nTT-1;TT,1;\n"
    assert (n, undef) == (7, 0)


@pytest.mark.parametrize("seed,n,pool", [(1, 2000, 50), (2, 5000, 2000), (3, 300, 1), (4, 1, 1)])
def test_tag_tree_matches_reference_filter_c(seed, n, pool):
    R = O.ref_tree_lib()
    if R is None:
        pytest.skip("oracle/_ref/libfastf_ref_tree.so not built (no /root/reference here)")
    rng = np.random.default_rng(seed)
    tags = _random_tags(rng, n, pool)
    txt, _, _ = O.run_extract(np.ones(n), vals=_S(tags))
    assert txt == _ref_print_tree(R, tags)


def test_crb_rows_match_reference_same_row_printer():
    R = O.ref_tree_lib()
    if R is None:
        pytest.skip("oracle/_ref/libfastf_ref_tree.so not built (no /root/reference here)")
    rng = np.random.default_rng(7)
    n = 4000
    cbs = _random_tags(rng, n, 40, alphabet=b"ACGT", length=8)
    crs = _random_tags(rng, n, 300, alphabet=b"ACGTN", length=8)
    txt, _, _ = O.run_crb(np.ones(n), np.ones(n), _S(cbs), _S(crs))
    rows = dict(line.split(b";", 1) for line in txt.split(b"\n") if line)
    # per CB, the CR part of the row is the reference's print_tree_same_row over that CB's CRs in record order
    for cb in set(cbs):
        mine = [crs[i] for i in range(n) if cbs[i] == cb]
        assert rows[cb] == _ref_same_row(R, mine)
    # and the CB order is the pre-order of the insertion-order BST: the reference's print_tree over the CBs
    order = [line.split(b";", 1)[0] for line in txt.split(b"\n") if line]
    want = [line.split(b",")[0] for line in _ref_print_tree(R, cbs).split(b"\n") if line]
    assert order == want


def test_golden_fixture_from_reference_tree_code():
    fx = json.load(open(os.path.join(HERE, "golden", "tag_tree.json")))
    for case in fx["extract"]:
        tags = [t.encode() for t in case["tags"]]
        txt, _, _ = O.run_extract(np.ones(len(tags)), vals=_S(tags))
        assert txt.decode() == case["print_tree"]
    for case in fx["same_row"]:
        tags = [t.encode() for t in case["tags"]]
        txt, _, _ = O.run_crb(np.ones(len(tags)), np.ones(len(tags)), _S([b"X"] * len(tags)), _S(tags))
        assert txt.decode() == "X;" + case["print_tree_same_row"] + "\n"
