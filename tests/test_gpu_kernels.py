"""Device-level entry points (fastf_dev_*) on HBM-resident buffers, checked against numpy
restatements of sort / group-by (bit-exact; integer work)."""
import numpy as np

from fastf_amd import hostmem
from oracle import oracle as O
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import fastf_amd as F
    assert torch.cuda.is_available()
    cells = np.arange(1, 1001, dtype=np.uint64) | (np.uint64(1) << np.uint64(62))
    feats = np.arange(1, 501, dtype=np.uint64) | (np.uint64(2) << np.uint64(62))
    eng = F.Engine(cells, feats, umi_max_bases=12)
    yield torch, F, eng
    eng.close()


def _t(torch, a):
    return hostmem.to_device(a, "cuda")


@pytest.mark.parametrize("n,bits", [(1, 8), (63, 17), (8192, 24), (8193, 33), (100_003, 47), (1_000_000, 56), (300_000, 64)])
def test_dev_sort_matches_numpy(env, n, bits):
    torch, F, eng = env
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    if bits < 64:
        keys &= np.uint64((1 << bits) - 1)
    d_keys = _t(torch, keys)
    d_tmp = torch.empty_like(d_keys)
    d_n = torch.tensor([n], dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    in_tmp = eng.dev_sort(d_keys.data_ptr(), d_tmp.data_ptr(), d_n.data_ptr(), n, key_bits=bits, stream=s)
    torch.cuda.synchronize()
    got = hostmem.to_host(d_tmp if in_tmp else d_keys).view(np.uint64)
    np.testing.assert_array_equal(got, np.sort(keys))
    assert eng.dev_error_bits() == 0


def test_dev_sort_respects_device_side_n(env):
    torch, F, eng = env
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 1 << 40, size=50_000, dtype=np.uint64)
    d_keys = _t(torch, keys)
    d_tmp = torch.zeros_like(d_keys)
    d_n = torch.tensor([12_345], dtype=torch.int64, device="cuda")
    in_tmp = eng.dev_sort(d_keys.data_ptr(), d_tmp.data_ptr(), d_n.data_ptr(), 50_000, key_bits=40,
                          stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = hostmem.to_host(d_tmp if in_tmp else d_keys).view(np.uint64)[:12_345]
    np.testing.assert_array_equal(got, np.sort(keys[:12_345]))


@pytest.mark.parametrize("n,n_groups,seed", [(1, 1, 0), (5000, 40, 1), (200_000, 150_000, 2), (200_000, 7, 3), (65_536, 65_536, 4)])
def test_dev_reduce_matches_numpy(env, n, n_groups, seed):
    torch, F, eng = env
    rng = np.random.default_rng(seed)
    # key layout of this engine: [cell 10][feature 9][nonnull 1][umi 24][len 2]
    fs, cs = 27, 36
    cell = rng.integers(1, 1001, size=n_groups, dtype=np.uint64)
    feat = rng.integers(1, 501, size=n_groups, dtype=np.uint64)
    g = rng.integers(0, n_groups, size=n)
    nonnull = (rng.random(n) > 0.1).astype(np.uint64)
    umi = rng.integers(0, 64, size=n, dtype=np.uint64) * nonnull
    ln = np.uint64(3) * nonnull
    keys = (cell[g] << np.uint64(cs)) | (feat[g] << np.uint64(fs)) | (nonnull << np.uint64(26)) | (umi << np.uint64(2)) | ln
    keys = np.sort(keys)
    d_keys = _t(torch, keys)
    d_n = torch.tensor([n], dtype=torch.int64, device="cuda")
    d_f = torch.empty(n, dtype=torch.int32, device="cuda")
    d_c = torch.empty(n, dtype=torch.int32, device="cuda")
    d_k = torch.empty(n, dtype=torch.int32, device="cuda")
    d_nnz = torch.zeros(1, dtype=torch.int64, device="cuda")
    eng.dev_reduce(d_keys.data_ptr(), d_n.data_ptr(), n, d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), d_nnz.data_ptr(),
                   stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    nnz = int(d_nnz.item())
    grp = keys >> np.uint64(fs)
    ug, start = np.unique(grp, return_index=True)
    assert nnz == len(ug)
    uk = np.unique(keys[(keys >> np.uint64(26)) & np.uint64(1) == 1])
    want = np.zeros(len(ug), dtype=np.int64)
    np.add.at(want, np.searchsorted(ug, uk >> np.uint64(fs)), 1)
    np.testing.assert_array_equal(hostmem.to_host(d_c)[:nnz].astype(np.int64), (ug >> np.uint64(cs - fs)).astype(np.int64))
    np.testing.assert_array_equal(hostmem.to_host(d_f)[:nnz].astype(np.int64), (ug & np.uint64((1 << 9) - 1)).astype(np.int64))
    np.testing.assert_array_equal(hostmem.to_host(d_k)[:nnz].astype(np.int64), want)
    assert eng.dev_error_bits() == 0


@pytest.mark.parametrize("mode", ["hash"])
@pytest.mark.parametrize("n,n_groups,low_values,seed", [(200_000, 50_000, 1 << 16, 1), (120_000, 3, 1 << 16, 2),
                                                        (60_000, 1, 5000, 3), (300_000, 200_000, 4, 4), (150_000, 2, 1 << 20, 5)])
def test_group_only_sort_plus_dedup_in_reduce(mode, n, n_groups, low_values, seed, monkeypatch):
    """FASTF_SORT_SKIP_LOW: only (cell, feature) is sorted; equal keys are neighbours of their group but unordered.  The
    reduce kernel must still count distinct non-NULL keys per (cell, feature) exactly — including groups far longer than
    a window (counted by giant_groups_kernel up to 65 536 keys) and heavy duplication.  What it cannot hold raises
    ERR_RUN_TOO_LONG: sort fully, reduce again."""
    import torch
    import fastf_amd as F
    cells = np.arange(1, 1001, dtype=np.uint64) | (np.uint64(1) << np.uint64(62))
    feats = np.arange(1, 501, dtype=np.uint64) | (np.uint64(2) << np.uint64(62))
    eng = F.Engine(cells, feats, umi_max_bases=12)
    try:
        _group_only_case(torch, eng, mode, n, n_groups, low_values, seed)
    finally:
        eng.close()


def _group_only_case(torch, eng, mode, n, n_groups, low_values, seed):
    skip = eng.skip_bits
    assert 0 < skip < 27 and eng.sort_passes(True) == (46 - skip + 7) // 8      # 46-bit keys: the digit grid ends at the top bit
    assert eng.sort_passes(True) == 3
    rng = np.random.default_rng(seed)
    fs, cs = 27, 36
    cell = rng.integers(1, 1001, size=n_groups, dtype=np.uint64)
    feat = rng.integers(1, 501, size=n_groups, dtype=np.uint64)
    g = rng.integers(0, n_groups, size=n)
    nonnull = (rng.random(n) > 0.05).astype(np.uint64)
    hi = rng.integers(0, 4, size=n, dtype=np.uint64) << np.uint64(22)          # few distinct sorted UMI bits → long runs
    lo = rng.integers(0, low_values, size=n, dtype=np.uint64) << np.uint64(2)  # part of it lies below bit 16
    umi = (hi | lo) & np.uint64((1 << 26) - 1)
    keys = (cell[g] << np.uint64(cs)) | (feat[g] << np.uint64(fs)) | ((nonnull << np.uint64(26)) | (umi * nonnull) | (np.uint64(3) * nonnull))
    d_keys = _t(torch, keys)
    d_tmp = torch.empty_like(d_keys)
    d_n = torch.tensor([n], dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    in_tmp = eng.dev_sort(d_keys.data_ptr(), d_tmp.data_ptr(), d_n.data_ptr(), n, stream=s, skip_low=True)
    src = d_tmp if in_tmp else d_keys
    got_keys = hostmem.to_host(src).view(np.uint64)
    assert (np.diff((got_keys >> np.uint64(skip)).astype(np.int64)) >= 0).all()           # sorted on the high bits
    np.testing.assert_array_equal(np.sort(got_keys), np.sort(keys))                       # a permutation
    d_f = torch.empty(n, dtype=torch.int32, device="cuda"); d_c = torch.empty_like(d_f); d_k = torch.empty_like(d_f)
    d_nnz = torch.zeros(1, dtype=torch.int64, device="cuda")
    eng.dev_reduce(src.data_ptr(), d_n.data_ptr(), n, d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), d_nnz.data_ptr(),
                   stream=s, skip_low=True)
    torch.cuda.synchronize()
    expect = n // n_groups > 65_536                           # a group beyond what the partition kernel takes
    flagged = bool(eng.dev_error_bits() & 16)
    assert flagged == expect
    if flagged:                                               # the documented contract: sort fully, reduce again
        eng.dev_clear_error_bits(16, s)
        other = d_keys if in_tmp else d_tmp
        in_other = eng.dev_sort(src.data_ptr(), other.data_ptr(), d_n.data_ptr(), n, stream=s)
        src = other if in_other else src
        eng.dev_reduce(src.data_ptr(), d_n.data_ptr(), n, d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), d_nnz.data_ptr(), stream=s)
        torch.cuda.synchronize()
        assert eng.dev_error_bits() == 0
    nnz = int(d_nnz.item())
    ug = np.unique(keys >> np.uint64(fs))
    assert nnz == len(ug)
    uk = np.unique(keys[(keys >> np.uint64(26)) & np.uint64(1) == 1])
    want = np.zeros(len(ug), dtype=np.int64)
    np.add.at(want, np.searchsorted(ug, uk >> np.uint64(fs)), 1)
    np.testing.assert_array_equal(hostmem.to_host(d_k)[:nnz].astype(np.int64), want)
    np.testing.assert_array_equal(hostmem.to_host(d_c)[:nnz].astype(np.int64), (ug >> np.uint64(cs - fs)).astype(np.int64))


def _want_rows(keys, fs=27, cs=36):
    ug = np.unique(keys >> np.uint64(fs))
    uk = np.unique(keys[(keys >> np.uint64(26)) & np.uint64(1) == 1])
    want = np.zeros(len(ug), dtype=np.int64)
    np.add.at(want, np.searchsorted(ug, uk >> np.uint64(fs)), 1)
    return (ug & np.uint64((1 << 9) - 1)).astype(np.int64), (ug >> np.uint64(cs - fs)).astype(np.int64), want


def _group_keys(rng, n, sizes_of_groups, null_frac=0.1, umi_values=1 << 24):
    """keys of the fixture engine's layout [cell 10][feature 9][nonnull 1][umi 24][len 2] with the given group sizes"""
    fs, cs = 27, 36
    ids = rng.choice(1000 * 500, size=len(sizes_of_groups), replace=False)
    cell = (ids // 500 + 1).astype(np.uint64); feat = (ids % 500 + 1).astype(np.uint64)
    g = np.repeat(np.arange(len(sizes_of_groups)), sizes_of_groups)[:n]
    m = len(g)
    nonnull = (rng.random(m) > null_frac).astype(np.uint64)
    umi = rng.integers(0, umi_values, size=m, dtype=np.uint64) * nonnull
    return (cell[g] << np.uint64(cs)) | (feat[g] << np.uint64(fs)) | (nonnull << np.uint64(26)) | (umi << np.uint64(2)) | (np.uint64(3) * nonnull)


@pytest.mark.parametrize("sizes,seed", [([1] * 70_000, 1), ([5000, 1, 1, 7000, 2048, 2047, 2049, 3, 4096, 1], 2),
                                        ([300_000], 3), ([2048] * 40, 4), ([1, 2047] * 60, 5)])
def test_reduce_windows_groups_longer_than_a_window_and_regions(env, sizes, seed):
    """K3 reads the keys once: windows cut at group heads, a group longer than a window carried as an open row, the rows
    left in per-workgroup regions and concatenated by the gather — into device arrays and straight into pinned host
    memory; the device-side key count is smaller than the caller's bound"""
    torch, F, eng = env
    rng = np.random.default_rng(seed)
    keys = np.sort(_group_keys(rng, 10**9, sizes, umi_values=64))
    n = len(keys)
    max_n = n + 12_345                                    # launch sized for an upper bound, n read on the device
    d_keys = _t(torch, np.concatenate([keys, np.full(max_n - n, np.uint64((1 << 46) - 1), np.uint64)]))
    d_n = torch.tensor([n], dtype=torch.int64, device="cuda")
    d_f = torch.empty(max_n, dtype=torch.int32, device="cuda"); d_c = torch.empty_like(d_f); d_k = torch.empty_like(d_f)
    d_nnz = torch.zeros(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    f, c, k = _want_rows(keys)
    # (1) segmented: only the count comes back, then the gather into device arrays
    eng.dev_reduce(d_keys.data_ptr(), d_n.data_ptr(), max_n, 0, 0, 0, d_nnz.data_ptr(), stream=s, segmented=True)
    torch.cuda.synchronize()
    assert int(d_nnz.item()) == len(f)
    eng.dev_rows_gather(d_n.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), stream=s)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(hostmem.to_host(d_f)[:len(f)].astype(np.int64), f)
    np.testing.assert_array_equal(hostmem.to_host(d_c)[:len(f)].astype(np.int64), c)
    np.testing.assert_array_equal(hostmem.to_host(d_k)[:len(f)].astype(np.int64), k)
    # (2) the same regions straight into pinned host memory
    h = [torch.empty(len(f) + 1, dtype=torch.int32).pin_memory() for _ in range(3)]
    for t in h:
        t.fill_(-7)
    eng.dev_rows_gather(d_n.data_ptr(), h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), stream=s)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(h[0].numpy()[:len(f)].astype(np.int64), f)
    np.testing.assert_array_equal(h[1].numpy()[:len(f)].astype(np.int64), c)
    np.testing.assert_array_equal(h[2].numpy()[:len(f)].astype(np.int64), k)
    assert h[0][len(f)] == -7 and h[2][len(f)] == -7       # nothing written past the last row
    assert eng.dev_error_bits() == 0


@pytest.mark.parametrize("sizes,too_long,seed", [([3] * 50_000, False, 1), ([2047, 1, 2047, 5, 900] * 30, False, 2),
                                                 ([100, 2048, 100], False, 3), ([40_000], False, 4), ([65_536, 9, 65_537, 1], True, 6),
                                                 ([5000, 1, 1, 7000, 2048, 2047, 2049, 3, 4096, 1, 30_000, 2], False, 7),
                                                 ([2100] * 2100, True, 8),      # more work items than giant_groups_kernel's list holds
                                                 ([3100] * 1500, True, 9),      # ... with groups of three items straddling the end of the list
                                                 (list(range(1, 600)), False, 5)])
def test_hash_dedup_reduce_on_group_sorted_keys(sizes, too_long, seed, monkeypatch):
    """The sort orders (cell, feature) only; K3 (reduce_hashed_kernel) finds the distinct UMIs of a group through its
    window-local exact set; a group that does not fit a window is counted by giant_groups_kernel, hash partition by hash
    partition; beyond 65 536 keys, or with more work items than the list holds (whose unwritten tail is never walked), it
    raises ERR_RUN_TOO_LONG (sort fully, reduce again)."""
    import torch
    import fastf_amd as F
    cells = np.arange(1, 1001, dtype=np.uint64) | (np.uint64(1) << np.uint64(62))
    feats = np.arange(1, 501, dtype=np.uint64) | (np.uint64(2) << np.uint64(62))
    eng = F.Engine(cells, feats, umi_max_bases=12)
    try:
        assert eng.skip_bits >= 22 and eng.sort_passes(True) == 3       # 46-bit keys: 19 group bits + the NULL flag in 3 passes
        rng = np.random.default_rng(seed)
        keys = _group_keys(rng, 10**9, sizes, umi_values=200 if seed != 7 else 1 << 24)   # heavy duplication / nearly all distinct
        keys = keys[rng.permutation(len(keys))]
        n = len(keys)
        # the launches are sized for the caller's BOUND on the key count (here 1x and 7x the count on the device): with a
        # loose bound the first chunks of K3 are empty and several workgroups start at key 0
        max_n = n * (7 if seed in (2, 5) else 1)
        d_keys = _t(torch, np.concatenate([keys, np.zeros(max_n - n, np.uint64)])); d_tmp = torch.empty_like(d_keys)
        d_n = torch.tensor([n], dtype=torch.int64, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        in_tmp = eng.dev_sort(d_keys.data_ptr(), d_tmp.data_ptr(), d_n.data_ptr(), max_n, stream=s, skip_low=True)
        src = d_tmp if in_tmp else d_keys
        d_f = torch.empty(max_n, dtype=torch.int32, device="cuda"); d_c = torch.empty_like(d_f); d_k = torch.empty_like(d_f)
        d_nnz = torch.zeros(1, dtype=torch.int64, device="cuda")
        eng.dev_reduce(src.data_ptr(), d_n.data_ptr(), max_n, d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), d_nnz.data_ptr(), stream=s, skip_low=True)
        torch.cuda.synchronize()
        flagged = bool(eng.dev_error_bits() & 16)
        assert flagged == too_long
        if flagged:
            eng.dev_clear_error_bits(16, s)
            other = d_keys if in_tmp else d_tmp
            in_other = eng.dev_sort(src.data_ptr(), other.data_ptr(), d_n.data_ptr(), max_n, stream=s)
            src = other if in_other else src
            eng.dev_reduce(src.data_ptr(), d_n.data_ptr(), max_n, d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), d_nnz.data_ptr(), stream=s)
            torch.cuda.synchronize()
            assert eng.dev_error_bits() == 0
        f, c, k = _want_rows(keys)
        assert int(d_nnz.item()) == len(f)
        np.testing.assert_array_equal(hostmem.to_host(d_f)[:len(f)].astype(np.int64), f)
        np.testing.assert_array_equal(hostmem.to_host(d_c)[:len(f)].astype(np.int64), c)
        np.testing.assert_array_equal(hostmem.to_host(d_k)[:len(f)].astype(np.int64), k)
    finally:
        eng.close()


@pytest.mark.parametrize("seed,skip,counts", [
    (926, 0, [1, 622, 1, 1, 623, 624, 625, 1248, 7, 100_000]),          # every position of a block boundary
    (926, 25_000, [3, 50_000, 0, 1_000_003]),                           # behind the draws of a cell sub-sampling (SampleInt)
    (0x39e, 624 * 1000 - 1, [1, 1, 624 * 3, 5]),
    (1, 623, [2_000_000]),
])
def test_device_mt19937_is_the_reference_stream(env, seed, skip, counts):
    """mt_fill_kernel (three sweeps per 624-word block) against the ORACLE's generator (oracle/fastf_oracle.c: mt19937ar.c:105-140
    restated, pinned to the reference's own file by tests/test_oracle_pins.py): launches of every size across block boundaries,
    continued from a host-side skip"""
    import ctypes as C
    torch, F, eng = env
    from fastf_amd import _lib
    L = _lib.lib()
    L.fastf_debug_mt_fill.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    n = int(sum(counts))
    out = np.zeros(max(n, 1), np.uint32)
    cs = np.asarray(counts, np.uint64)
    assert L.fastf_debug_mt_fill(0, seed, skip, cs.ctypes.data, len(counts), out.ctypes.data) == 0, L.fastf_last_error()
    np.testing.assert_array_equal(out[:n], O.mt_stream(seed, n, skip=skip))


@pytest.mark.parametrize("seed,skip,first,counts,rate,ring_bits", [
    (926, 0, 0, [1, 622, 1, 1, 623, 624, 625, 1248, 7, 100_000], 0.5, 1 << 17),     # every position of a block boundary
    (926, 25_000, 13, [3, 50_000, 0, 31, 1, 32, 33, 1_000_003], 0.5, 1 << 21),      # launches that start and end inside a word
    (0x39e, 624 * 1000 - 1, 4095, [1, 1, 624 * 3, 5, 59, 64], 0.123, 4096),         # the ring wraps inside the third launch
    (1, 623, 31, [2_000_000], 0.9, 1 << 21),
    (7, 5, 1000, [40, 700, 3], 1.0, 1024),
    (11, 0, 63, [1, 1, 62, 64, 65, 127, 1, 623, 625], 0.7, 1 << 13),                # words that take several launches to fill
    (12, 300, 5, [5, 2, 1, 0, 3], 0.3, 64),                                         # a ring of one word                                          # rate 1: the threshold is 2^32, every bit set
])
def test_device_decision_stream(env, seed, skip, first, counts, rate, ring_bits):
    """mt_fill_kernel<true> — what K1b reads: bit (r & 31) of ring word (r mod ring) >> 5 = draw (r - first) < threshold, for every
    absolute rank r the launches cover; a wave's ballot is a whole 64-bit word of the ring, the word a block or a launch ends in
    is carried.  Ranks never written keep the 0xFF the hook fills the ring with, except the zero tail of the last word."""
    import ctypes as C
    torch, F, eng = env
    from fastf_amd import _lib
    L = _lib.lib()
    thr = int(L.fastf_draw_threshold(rate))
    L.fastf_debug_mt_fill_bits.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
    n = int(sum(counts))
    out = np.zeros(ring_bits // 32, np.uint32)
    cs = np.asarray(counts, np.uint64)
    assert L.fastf_debug_mt_fill_bits(0, seed, skip, first, cs.ctypes.data, len(counts), thr, ring_bits, out.ctypes.data) == 0, L.fastf_last_error()
    got = np.unpackbits(out.view(np.uint8), bitorder="little")
    want = (O.mt_stream(seed, n, skip=skip).astype(np.uint64) < np.uint64(thr)).astype(np.uint8)
    keep = min(n, ring_bits - 64)                      # the newest ranks; older ones may have been overwritten by the wrap
    r = np.arange(first + n - keep, first + n, dtype=np.uint64)
    np.testing.assert_array_equal(got[(r % np.uint64(ring_bits)).astype(np.int64)], want[n - keep:])
    end = first + n
    if end % 64 and keep == n and n < ring_bits - 128:
        tail = np.arange(end, (end // 64 + 1) * 64, dtype=np.uint64) % np.uint64(ring_bits)
        assert not got[tail.astype(np.int64)].any()                                  # the rest of the last 64-bit word: zeros
        beyond = np.arange((end // 64 + 1) * 64, (end // 64 + 2) * 64, dtype=np.uint64) % np.uint64(ring_bits)
        assert got[beyond.astype(np.int64)].all()                                    # the word after it: untouched


J_SUB = 624 * 256                                      # draws per sub-stream of the parallel generator (mt_jump.h)


@pytest.mark.parametrize("seed,skip,n,rate", [
    (926, 0, 5 * J_SUB + 1234, 0.5),                   # a stream that starts at a block boundary (seeded: the first draw regenerates)
    (926, 100, 4 * J_SUB, 0.3),                        # ... in the middle of a block: the rest of it goes out first; the smallest parallel call
    (7, 7, 4 * J_SUB + 1, 0.7),
    (5489, 624 * 3 + 5, 20_000_000, 0.5),              # 63 sub-streams: one coarse jump (sub-stream 32), then the fine ones behind 0 and 32
    (1, 623, 9 * J_SUB - 623, 1.0),                    # ends exactly on a sub-stream boundary
    (926, 11, 33 * J_SUB + 5, 0.5),                    # sub-stream 32 is seated by the coarse launch and has one fine sub-stream behind it
    (3, 0, 1027 * J_SUB + 77, 0.25),                   # more than R^2 = 1024 sub-streams: a second round continues the first's last sub-stream
])
def test_parallel_mt_decisions_are_the_reference_stream(env, seed, skip, n, rate):
    """fastf_dev_mt_decisions: the MT19937 stream (mt19937ar.c:105-140) from many workgroups at once — sub-streams 624 x 256 draws
    apart, seated by jump-ahead in two launches (mt_jump_kernel: the state J draws on is g(F) applied to the state now, g = x^J mod
    the characteristic polynomial; sub-streams 32, 64, .. from the stream's state, then the 31 behind each) and generated side by
    side — must be the stream the ORACLE's one generator produces, decision for decision"""
    torch, F, eng0 = env
    from fastf_amd import _lib
    from helpers import Case
    L = _lib.lib()
    case = Case(n=10, n_bar=20, n_gene=10, rate_depth=rate)
    lists = case.lists()
    eng = F.Engine.from_lists(lists, rate_depth=rate, seed=926)
    try:
        thr = int(L.fastf_draw_threshold(rate))
        words = torch.full(((max(n, 4 * J_SUB + 99) + 63) // 64 * 2 + 2,), -1, dtype=torch.int32, device="cuda")
        eng.dev_mt_decisions(seed, skip, n, words.data_ptr())
        torch.cuda.synchronize()
        got = np.unpackbits(hostmem.to_host(words).view(np.uint8), bitorder="little")
        want = (O.mt_stream(seed, n, skip=skip).astype(np.uint64) < np.uint64(thr)).astype(np.uint8)
        np.testing.assert_array_equal(got[:n], want)
        assert not got[n:(n + 63) // 64 * 64].any()    # the tail of the last 64-bit word is zero
        # a second call on the same engine (the polynomials and the sub-stream states are in place) from another position
        eng.dev_mt_decisions(seed + 1, skip + 17, 4 * J_SUB + 99, words.data_ptr())
        torch.cuda.synchronize()
        got = np.unpackbits(hostmem.to_host(words).view(np.uint8), bitorder="little")
        want = (O.mt_stream(seed + 1, 4 * J_SUB + 99, skip=skip + 17).astype(np.uint64) < np.uint64(thr)).astype(np.uint8)
        np.testing.assert_array_equal(got[:4 * J_SUB + 99], want)
    finally:
        eng.close()
