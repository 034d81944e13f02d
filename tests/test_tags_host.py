"""Host side of crb / extract (CPU): exact key packing with registration + decoding, the BAM tag reader, and the
tree-order logic (insertion-order BST = Cartesian tree over first occurrences) against the oracle."""
import ctypes as C
import threading

import numpy as np
import pytest

import fastf_amd as F
from fastf_amd.tags import KeyDict
from oracle import oracle as O
from tag_helpers import TagCase, random_dna


STRINGS = [b"ACGTACGTACGTACGT-1", b"ACGTACGTACGTACGT", b"ACGNACGTACGTACGT", b"NNNN-12", b"N", b"ENSG00000123456", b"ENSG00000123457",
           b"MALAT1", b"", b"AC012345.1", b"G1;G2", b"A" * 24, b"A" * 25, b"ACGTN" * 3 + b"N", b"ACGTN" * 3 + b"NN", b"x007", b"x7",
           b"-3", b"25", b"0", b"00", b"ACGT-0", b"ACGT-00", b"ACGT-255", b"ACGT-254", b"acgt", b"ACGT-1-1", b"T" * 16 + b"-1"]


def test_keys_are_exact_and_decodable():
    d = KeyDict()
    keys = [d.intern(s) for s in STRINGS]
    assert len(set(keys)) == len(STRINGS) and 0 not in keys            # distinct strings, distinct non-zero keys
    for s, k in zip(STRINGS, keys):
        assert d.decode(k) == s
        assert d.intern(s) == k                                          # stable
    # the same string packs to the same key through the read-only route of the bam2db path
    L = F.lib()
    for s, k in zip(STRINGS, keys):
        assert int(L.fastf_keydict_pack(d.h, s, len(s))) == k
    d.close()


def test_intern_is_thread_safe():
    d = KeyDict()
    rng = np.random.default_rng(5)
    pool = [b"GENE-%d.%d" % (i % 13, i) for i in range(3000)] + random_dna(rng, 2000, 12, b"ACGTN") + [b"ID%07d" % i for i in range(2000)]
    got = [None] * 8

    def work(t):
        order = np.random.default_rng(t).permutation(len(pool))
        got[t] = {pool[i]: d.intern(pool[i]) for i in order}
    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    for t in range(1, 8):
        assert got[t] == got[0]
    assert len(set(got[0].values())) == len(set(pool))
    for s, k in got[0].items():
        assert d.decode(k) == s
    d.close()


def _read_tags(path, tag1, tag2, type_, cap=1 << 16, threads=4):
    L = F.lib()
    d = KeyDict()
    b = L.fastf_bam_open(path.encode(), threads)
    assert b
    k1s, k2s, undef = [], [], C.c_uint64(0)
    k1 = np.zeros(cap, dtype=np.uint64); k2 = np.zeros(cap, dtype=np.uint64)
    while True:
        n = L.fastf_bam_read_tags(b, d.h, tag1, tag2, type_, k1.ctypes.data, k2.ctypes.data if tag2 else None, cap, C.byref(undef))
        assert n >= 0
        if n == 0:
            break
        k1s.append(k1[:n].copy()); k2s.append(k2[:n].copy())
    L.fastf_bam_close(b)
    return d, np.concatenate(k1s), np.concatenate(k2s), int(undef.value)


def test_bam_tag_reader(tmp_path):
    c = TagCase(n=30_000, n_cb=100, seed=3)
    path = str(tmp_path / "t.bam")
    c.write(path)
    d, k1, k2, undef = _read_tags(path, b"CB", b"CR", 0, cap=7000)
    assert undef == 0 and len(k1) == c.n
    assert ((k1 != 0) == c.has_cb).all() and (k2 != 0).all()
    for i in list(range(0, c.n, 997)):
        if c.has_cb[i]:
            assert d.decode(k1[i]) == c.cb[i]
        assert d.decode(k2[i]) == c.cr[i]
    # integer mode: key = 2^32 | low 32 bits of the value (bam_aux2i → "%d", extract.c:192)
    d2, k, _, _ = _read_tags(path, b"NH", None, 1)
    np.testing.assert_array_equal(k, (np.uint64(1) << np.uint64(32)) | (c.nh.astype(np.int32).view(np.uint32).astype(np.uint64)))
    # string mode on an integer tag: bam_aux2Z gives NULL → undefined in the reference, absent here
    d3, k, _, undef = _read_tags(path, b"xf", None, 0)
    assert (k == 0).all() and undef == c.n
    with pytest.raises(AssertionError):
        _read_tags(path, b"TOOLONG", None, 0)


@pytest.mark.parametrize("seed,m,shuffle", [(1, 1, False), (2, 2, False), (3, 50, False), (4, 3000, False), (5, 3000, True), (6, 70000, True),
                                            (7, 70000, False)])
def test_tree_order_equals_insertion_order_bst(seed, m, shuffle):
    """the pre-order ranks computed without building the tree (two monotonic-stack sweeps over the strcmp order) against the
    oracle's insert_tree / print_tree; entries handed over in strcmp order (the sort is skipped) and shuffled (it is not);
    70 000 entries take the parallel order check"""
    rng = np.random.default_rng(seed)
    strs = sorted(set(random_dna(rng, m, 9, b"ACGTN-1")))
    if shuffle:
        strs = [strs[i] for i in rng.permutation(len(strs))]
    m = len(strs)
    first = rng.permutation(10 * m)[:m].astype(np.uint64)                # distinct first-occurrence positions
    L = F.lib()
    L.fastf_tag_tree_preorder.argtypes = [C.POINTER(C.c_char_p), C.c_void_p, C.c_uint32, C.c_void_p]
    L.fastf_tag_tree_preorder.restype = None
    arr = (C.c_char_p * m)(*strs)
    order = np.zeros(m, dtype=np.uint32)
    L.fastf_tag_tree_preorder(arr, first.ctypes.data, m, order.ctypes.data)
    # the oracle inserts the strings in order of first occurrence and prints the tree in pre-order
    seq = [strs[i] for i in np.argsort(first)]
    txt, _, _ = O.run_extract(np.ones(m), vals=TagCase.S(seq))
    want = [line.split(b",")[0] for line in txt.split(b"\n") if line]
    assert [strs[i] for i in order] == want
