"""GPU end-to-end: the fastF CLI / bam2db() on real BAM files vs the oracle's bytes, and the
sharded device path (n_shards > 1) simulated on one GPU."""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np
import pytest

import fastf_amd as F
from fastf_amd import hostmem
from fastf_amd import synth, _lib
from fastf_amd import workload as workload_mod
from fastf_amd.dist import owner_of_cell
from helpers import Case

pytestmark = pytest.mark.gpu


def _write_inputs(tmp_path, case, gz_lists=False):
    bam = tmp_path / "in.bam"
    synth.write_bam(str(bam), case.flags, case.xf, case.cb, case.gx, case.ub)
    if gz_lists:
        b, f = tmp_path / "barcodes.in.tsv.gz", tmp_path / "features.in.tsv.gz"
        b.write_bytes(gzip.compress(case.bt)); f.write_bytes(gzip.compress(case.ft))
    else:
        b, f = tmp_path / "barcodes.in.tsv", tmp_path / "features.in.tsv"
        b.write_bytes(case.bt); f.write_bytes(case.ft)
    return bam, b, f


def _read_gz(p):
    return gzip.decompress(open(p, "rb").read())


@pytest.mark.parametrize("name,kw,args,gz", [
    ("keepall", dict(n=60_000, n_bar=500, n_gene=200, umi_pool=64, p_n_umi=0.01), ["-c", "1", "-r", "1"], False),
    ("half", dict(n=120_000, n_bar=1000, n_gene=500, rate_cell=0.5, rate_depth=0.5, umi_len=12, dup_factor=3.0,
                  p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.15, p_n_umi=0.005, p_multi_gene=0.02),
     ["--cell=0.5", "--depth", "0.5", "-s926"], True),
    ("seed", dict(n=20_000, n_bar=50, n_gene=40, rate_cell=0.3, rate_depth=0.25, seed=0x39e, umi_pool=16),
     ["-c", "0.3", "-r0.25", "--seed=0x39e"], False),
])
def test_cli_outputs_equal_oracle_bytes(tmp_path, name, kw, args, gz):
    case = Case(**kw)
    bam, b, f = _write_inputs(tmp_path, case, gz)
    out = tmp_path / "out"; out.mkdir()
    case.label = str(bam).encode()
    ora = case.oracle()
    env = dict(os.environ, FASTF_BATCH_RECORDS="25000")          # several pushes
    r = subprocess.run([_lib.cli_path(), "bam2db", "-b", str(bam), "-a", str(b), "-f", str(f), "-o", str(out),
                        "-d", str(tmp_path / "unused.db"), "-u"] + args, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert _read_gz(out / "matrix.mtx.gz") == ora["matrix"]
    assert _read_gz(out / "barcodes.tsv.gz") == ora["barcodes"]
    assert _read_gz(out / "features.tsv.gz") == ora["features"]
    assert _read_gz(out / "umi.tsv.gz") == ora["umi"]
    assert "total fastQ reads: %d" % ora["total"] in r.stdout
    assert not (tmp_path / "unused.db").exists()


def test_cli_with_a_multi_million_line_barcode_list(tmp_path):
    """A raw whitelist instead of the filtered barcodes: 2.2 M lines x 36 601 genes = 22 + 16 key bits, which leaves room for
    12-base UMIs (27 bits: 65 > 64) only with keys wider than 64 bits — hashtable.c:70-115 takes any number of barcodes, so
    does bam2db() here (the engine sorts the (cell, feature) word and carries the rest of the key beside it).  -u rows too."""
    case = Case(n=150_000, n_bar=2_200_000, n_gene=36_601, rate_cell=1.0, rate_depth=0.8, umi_len=12, dup_factor=3.0,
                gene_dist="zipf", p_unlisted_cb=0.05, p_bad_xf=0.1, p_n_umi=0.01, data_seed=31)
    bam, b, f = _write_inputs(tmp_path, case)
    out = tmp_path / "out"; out.mkdir()
    case.label = str(bam).encode()
    ora = case.oracle()
    r = subprocess.run([_lib.cli_path(), "bam2db", "-b", str(bam), "-a", str(b), "-f", str(f), "-o", str(out), "-c", "1", "-r", "0.8", "-u"],
                       capture_output=True, text=True, env=dict(os.environ, FASTF_BATCH_RECORDS="40000"))
    assert r.returncode == 0, r.stderr
    assert _read_gz(out / "matrix.mtx.gz") == ora["matrix"]
    assert _read_gz(out / "umi.tsv.gz") == ora["umi"]
    assert _read_gz(out / "barcodes.tsv.gz") == ora["barcodes"]


def test_cli_runs_again_when_the_umis_are_longer_than_the_64_bit_key_holds(tmp_path):
    """70 k x 70 k lists leave 27 key bits: bam2db() starts with 12-base UMIs (the 64-bit path) and, when the file holds
    16-base UMIs, runs once more with keys wider than 64 bits — the reference takes a UMI of any length (bam2db_ds.c:417-419)"""
    case = Case(n=80_000, n_bar=70_000, n_gene=70_000, rate_cell=0.9, rate_depth=0.9, umi_len=16, umi_pool=2048, p_n_umi=0.01, data_seed=33)
    bam, b, f = _write_inputs(tmp_path, case)
    out = tmp_path / "out"; out.mkdir()
    case.label = str(bam).encode()
    ora = case.oracle()
    r = subprocess.run([_lib.cli_path(), "bam2db", "-b", str(bam), "-a", str(b), "-f", str(f), "-o", str(out), "-c", "0.9", "-r", "0.9"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "running again" in r.stderr
    assert _read_gz(out / "matrix.mtx.gz") == ora["matrix"]


@pytest.mark.parametrize("umi_len", [20, 30])
def test_cli_with_20_and_30_base_umis(tmp_path, umi_len):
    """small lists (16-base UMIs fit the 64-bit key: the first run), 20- or 30-base UMIs in the file: bam2db() runs again with room
    for 32 bases; matrix and -u rows are the oracle's (decode_DNA(..., 10) prints the first ten bases, bam2db_ds.c:629)"""
    case = Case(n=60_000, n_bar=400, n_gene=150, rate_cell=0.8, rate_depth=0.7, umi_len=umi_len, umi_pool=512, p_n_umi=0.02, p_bad_xf=0.1, data_seed=35)
    bam, b, f = _write_inputs(tmp_path, case)
    out = tmp_path / "out"; out.mkdir()
    case.label = str(bam).encode()
    ora = case.oracle()
    r = subprocess.run([_lib.cli_path(), "bam2db", "-b", str(bam), "-a", str(b), "-f", str(f), "-o", str(out), "-c", "0.8", "-r", "0.7", "-u"],
                       capture_output=True, text=True, env=dict(os.environ, FASTF_BATCH_RECORDS="25000"))
    assert r.returncode == 0, r.stderr
    assert "room for 32" in r.stderr
    assert _read_gz(out / "matrix.mtx.gz") == ora["matrix"]
    assert _read_gz(out / "umi.tsv.gz") == ora["umi"]


@pytest.mark.parametrize("env_extra", [
    dict(FASTF_BATCH_RECORDS="70000", FASTF_BAM_WINDOW="262144"),                           # the slot ring wraps, many windows per slot
    dict(FASTF_BATCH_RECORDS="70000", FASTF_BAM_WINDOW="262144", FASTF_GPU_INFLATE="2"),    # every window on the device: device-packed batches between host ones
    dict(FASTF_BATCH_RECORDS="20000", FASTF_BAM_WINDOW="131072", FASTF_BAM_SCOUT="0", FASTF_ZERO_COPY="0"),   # no scout, staged pushes
    dict(FASTF_BATCH_RECORDS="300000", FASTF_BAM_WINDOW="1048576", FASTF_LEND_ROWS="0", FASTF_BAM_MMAP="0"),  # read() windows, own row buffer
])
def test_cli_decoder_ring_windows_and_slots(tmp_path, env_extra):
    """bam2db()'s front end under small windows and small slots: the decoder fills a slot across several windows while the
    engine starts, walks through the ring of slots, hands device-packed batches over between host-packed ones, the scout
    walks ahead — the outputs stay the oracle's bytes and the counters the file's (record order = draw order)."""
    case = Case(n=260_000, n_bar=900, n_gene=300, rate_cell=0.6, rate_depth=0.5, umi_len=12, dup_factor=3.0,
                p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.15, p_n_umi=0.005, p_multi_gene=0.02)
    bam, b, f = _write_inputs(tmp_path, case)
    out = tmp_path / "out"; out.mkdir()
    case.label = str(bam).encode()
    ora = case.oracle()
    r = subprocess.run([_lib.cli_path(), "bam2db", "-b", str(bam), "-a", str(b), "-f", str(f), "-o", str(out), "-u",
                        "-c", "0.6", "-r", "0.5"], capture_output=True, text=True, env=dict(os.environ, **env_extra))
    assert r.returncode == 0, r.stderr
    assert _read_gz(out / "matrix.mtx.gz") == ora["matrix"]
    assert _read_gz(out / "umi.tsv.gz") == ora["umi"]
    assert "total fastQ reads: %d" % ora["total"] in r.stdout


@pytest.mark.parametrize("kw", [
    dict(n=0, n_bar=10, n_gene=5),                                   # header-only BAM
    dict(n=1, n_bar=10, n_gene=5),                                   # one record
    dict(n=3000, n_bar=40, n_gene=20, p_no_cb=1.0),                  # no record carries a CB: an empty matrix with its header
    dict(n=3000, n_bar=40, n_gene=20, p_bad_xf=1.0),                 # every record fails the xf test
    dict(n=3000, n_bar=40, n_gene=20, p_n_umi=1.0),                  # only NULL blobs: rows with count 0
])
def test_cli_edge_inputs(tmp_path, kw):
    """the CLI on degenerate inputs: the reference's files all the same (header, dims line with nnz 0, count-0 rows)"""
    case = Case(**kw)
    bam, b, f = _write_inputs(tmp_path, case)
    out = tmp_path / "out"; out.mkdir()
    case.label = str(bam).encode()
    ora = case.oracle()
    r = subprocess.run([_lib.cli_path(), "bam2db", "-b", str(bam), "-a", str(b), "-f", str(f), "-o", str(out), "-u"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert _read_gz(out / "matrix.mtx.gz") == ora["matrix"]
    assert _read_gz(out / "barcodes.tsv.gz") == ora["barcodes"]
    assert _read_gz(out / "features.tsv.gz") == ora["features"]
    assert _read_gz(out / "umi.tsv.gz") == ora["umi"]
    assert "total fastQ reads: %d" % ora["total"] in r.stdout


def test_cli_truncated_bam_fails_cleanly(tmp_path):
    """a BAM cut in the middle of a BGZF block: `Error: ...` on stderr, exit status 1 (bam2db_ds.h:60), no crash, no hang"""
    case = Case(n=40_000, n_bar=100, n_gene=50)
    bam, b, f = _write_inputs(tmp_path, case)
    data = open(bam, "rb").read()
    cut = tmp_path / "cut.bam"
    cut.write_bytes(data[:len(data) * 2 // 3])
    out = tmp_path / "out"; out.mkdir()
    r = subprocess.run([_lib.cli_path(), "bam2db", "-b", str(cut), "-a", str(b), "-f", str(f), "-o", str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert "Error" in r.stderr and "truncated" in r.stderr.lower()


REFMAIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "fastF_refmain")


@pytest.mark.skipif(not os.path.exists(REFMAIN), reason="oracle/_ref/fastF_refmain not built (needs /root/reference at build time)")
def test_reference_main_drives_the_engine(tmp_path):
    """the reference's OWN main.c + argparse.c (compiled in place, `make -C oracle refcli`) with libfastf_amd.so linked in
    place of bam2db_ds.c: `fastF bam2db ...` through the reference's dispatch table and option parser lands in the
    MI355X engine and writes the oracle's bytes"""
    case = Case(n=80_000, n_bar=400, n_gene=150, rate_cell=0.5, rate_depth=0.5, umi_len=12, dup_factor=3.0,
                p_no_cb=0.05, p_unlisted_cb=0.05, p_bad_xf=0.15, p_n_umi=0.005)
    bam, b, f = _write_inputs(tmp_path, case)
    out = tmp_path / "out"; out.mkdir()
    case.label = str(bam).encode()
    ora = case.oracle()
    r = subprocess.run([REFMAIN, "bam2db", "-b", str(bam), "-a", str(b), "-f", str(f), "-o", str(out),
                        "-d", str(tmp_path / "unused.db"), "--cell=0.5", "-r", "0.5", "-s", "926", "-u"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert _read_gz(out / "matrix.mtx.gz") == ora["matrix"]
    assert _read_gz(out / "barcodes.tsv.gz") == ora["barcodes"]
    assert _read_gz(out / "features.tsv.gz") == ora["features"]
    assert _read_gz(out / "umi.tsv.gz") == ora["umi"]


@pytest.mark.own_process
def test_bam2db_symbol_in_process(tmp_path):
    """the drop-in C symbol itself (what the reference's main.c would call)"""
    import torch  # noqa: F401
    case = Case(n=15_000, n_bar=100, n_gene=60, rate_depth=0.8, umi_pool=32)
    bam, b, f = _write_inputs(tmp_path, case)
    case.label = str(bam).encode()
    case.umi_copies = False
    ora = case.oracle()
    L = _lib.lib()
    rc = L.bam2db(str(bam).encode(), None, str(tmp_path).encode(), str(b).encode(), str(f).encode(),
                  C.c_float(1.0), C.c_float(0.8), 926)
    assert rc == 0
    assert _read_gz(tmp_path / "matrix.mtx.gz") == ora["matrix"]
    assert not (tmp_path / "umi.tsv.gz").exists()
    L.fastf_debug_live_registrations.restype = int
    assert L.fastf_debug_live_registrations() == 0          # window buffers, decoder slab, lent rows: all unregistered before release
    # error convention: 1 + message, no crash
    assert L.bam2db(b"/nonexistent.bam", None, str(tmp_path).encode(), str(b).encode(), str(f).encode(),
                    C.c_float(1.0), C.c_float(1.0), 926) == 1


@pytest.mark.own_process
@pytest.mark.parametrize("G", [2, 4, 8])
def test_sharded_device_path_on_one_gpu(G):
    """n_shards = G engines on one device: slices → K1a/K1b with a draw-rank base → per-shard buffers →
    (in-process exchange) → sort + reduce per shard → merged rows == oracle"""
    import torch
    case = Case(n=200_000, n_bar=700, n_gene=300, rate_cell=0.8, rate_depth=0.6, umi_pool=512,
                cell_dist="lognormal", p_unlisted_cb=0.1, p_bad_xf=0.1, p_n_umi=0.01)
    ora = case.oracle()
    lists = case.lists()
    cbk, gxk, umi, meta = case.packed(lists)
    n = case.n
    dev = torch.device("cuda")
    t = lambda a: hostmem.to_device(a, dev)
    draws = t(F.mt_draws(case.seed, lists.mt_skip, n))
    engs = [F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, n_shards=G, shard_rank=r) for r in range(G)]
    s = torch.cuda.current_stream().cuda_stream
    cuts = [n * r // G for r in range(G + 1)]
    stride = max(cuts[r + 1] - cuts[r] for r in range(G))
    try:
        hits = []
        sl = []
        for r in range(G):
            a, b = cuts[r], cuts[r + 1]
            sl.append([t(x[a:b].copy()) for x in (cbk, gxk, umi, meta)])
            h = torch.zeros(1, dtype=torch.int64, device=dev)
            engs[r].dev_count_hits(sl[r][0].data_ptr(), b - a, h.data_ptr(), s)
            hits.append(h)
        torch.cuda.synchronize()
        base, acc = [], 0
        for r in range(G):
            base.append(acc); acc += int(hits[r].item())
        keys_out = torch.zeros((G, G, stride), dtype=torch.int64, device=dev)
        kc = torch.zeros((G, G), dtype=torch.int64, device=dev)
        counters = torch.zeros((G, 4), dtype=torch.int64, device=dev)
        for r in range(G):
            a, b = cuts[r], cuts[r + 1]
            db = torch.tensor([base[r]], dtype=torch.int64, device=dev)
            engs[r].dev_probe_pack(sl[r][0].data_ptr(), sl[r][1].data_ptr(), sl[r][2].data_ptr(), sl[r][3].data_ptr(), b - a,
                                   draws.data_ptr(), n, keys_out[r].data_ptr(), stride, kc[r].data_ptr(), counters[r].data_ptr(), s,
                                   d_draw_base=db.data_ptr())
        torch.cuda.synchronize()
        kc_h = hostmem.to_host(kc)
        tot = hostmem.to_host(counters.sum(0))
        assert (int(tot[1]), int(tot[2]), int(tot[3])) == (ora["sampled"], ora["valid"], 0)
        F_, C_, K_ = [], [], []
        for sh in range(G):
            recv = torch.cat([keys_out[r, sh, :kc_h[r, sh]] for r in range(G)])
            m = recv.numel()
            tmp = torch.empty_like(recv)
            d_n = torch.tensor([m], dtype=torch.int64, device=dev)
            in_tmp = engs[sh].dev_sort(recv.data_ptr(), tmp.data_ptr(), d_n.data_ptr(), m, stream=s)
            src = tmp if in_tmp else recv
            f = torch.empty(max(m, 1), dtype=torch.int32, device=dev); c = torch.empty_like(f); k = torch.empty_like(f)
            nnz = torch.zeros(1, dtype=torch.int64, device=dev)
            engs[sh].dev_reduce(src.data_ptr(), d_n.data_ptr(), m, f.data_ptr(), c.data_ptr(), k.data_ptr(), nnz.data_ptr(), s)
            torch.cuda.synchronize()
            z = int(nnz.item())
            cc = hostmem.to_host(c[:z]).astype(np.int64)
            assert (owner_of_cell(cc, G) == sh).all()           # host mirror of the device ownership hash
            F_.append(hostmem.to_host(f[:z]).astype(np.int64)); C_.append(cc); K_.append(hostmem.to_host(k[:z]).astype(np.int64))
        Fa, Ca, Ka = np.concatenate(F_), np.concatenate(C_), np.concatenate(K_)
        order = np.lexsort((Fa, Ca))
        np.testing.assert_array_equal(Fa[order], ora["feature"].astype(np.int64))
        np.testing.assert_array_equal(Ca[order], ora["cell"].astype(np.int64))
        np.testing.assert_array_equal(Ka[order], ora["count"].astype(np.int64))
    finally:
        for e in engs:
            e.close()


@pytest.mark.own_process
def test_full_size_properties_config2():
    """BASELINE configs[1] at full size (10 M records): size-independent properties instead of the oracle:
    sortedness of rows, row-count/sum invariants, idempotence, and equality with a 2-batch push."""
    import torch  # noqa: F401
    bt, ft, bar, genes = synth.make_lists(10_000, 30_000, seed=4242)
    lists = F.Lists(bt, ft, 1.0, 926)
    rng = np.random.default_rng(3)
    n = 10_000_000
    cbk = lists.cell_keys[rng.integers(0, lists.n_cells, n)]
    gxk = lists.feature_keys[rng.integers(0, lists.n_features, n)]
    pool = rng.integers(0, 1 << 20, size=1 << 16, dtype=np.uint32)
    umi = (pool[rng.integers(0, len(pool), n)] << np.uint32(12)).astype(np.uint32)
    meta = np.full(n, 1 | 2 | 4 | (3 << 4), dtype=np.uint32)
    eng = F.Engine.from_lists(lists, rate_depth=0.5, seed=926, batch_records=6_000_000)
    try:
        eng.push(cbk, gxk, umi, meta)
        r1 = eng.finish()
        assert r1["total"] == n
        T = F.draw_threshold(0.5)
        assert r1["sampled"] == int((F.mt_draws(926, 0, n) < T).sum())      # every record hits: draws 0..n-1
        assert r1["valid"] == r1["sampled"]
        key = r1["cell"].astype(np.int64) * (1 << 20) + r1["feature"].astype(np.int64)
        assert (np.diff(key) > 0).all()                                      # strictly ascending (cell, feature)
        assert r1["count"].min() >= 1 and int(r1["count"].sum()) <= r1["valid"]
        # distinct (cell, feature, umi) triples of the kept records, computed independently on the host
        keep = F.mt_draws(926, 0, n) < T
        trip = np.unique(np.stack([cbk[keep], gxk[keep], umi[keep].astype(np.uint64)]), axis=1).shape[1]
        assert int(r1["count"].sum()) == trip
        r1b = eng.finish()                                                   # idempotent
        assert r1b["nnz"] == r1["nnz"] and np.array_equal(r1b["count"], r1["count"])
        eng.reset(); eng.reseed(926, 0)
        h = n // 3
        eng.push(cbk[:h], gxk[:h], umi[:h], meta[:h]); eng.push(cbk[h:], gxk[h:], umi[h:], meta[h:])
        r2 = eng.finish()
        for k in ("cell", "feature", "count"):
            assert np.array_equal(r1[k], r2[k])
    finally:
        eng.close()


def _expected_matrix_torch(job, dev, cb, gx, umi, meta, draws_h):
    """the matrix of a packed record stream, computed with torch ops only (sorted search, cumsum, unique) — an independent
    route to COUNT(DISTINCT umi) GROUP BY cell, feature with the reference's draw-per-hit rule (bam2db_ds.c:374-435)"""
    import torch
    lists = job.lists
    skeys = hostmem.to_device(lists.cell_keys, dev)            # cell_index i+1 <-> skeys[i]
    sk, order = torch.sort(skeys)
    pos = torch.searchsorted(sk, cb).clamp_(max=len(sk) - 1)
    hit = (cb != 0) & (sk[pos] == cb)
    cell_idx = (order[pos] + 1)                                                 # valid where hit
    rank = torch.cumsum(hit.to(torch.int64), 0) - 1
    n_hits = int(hit.sum().item())
    T = F.draw_threshold(workload_mod.RATE_DEPTH)
    keep_by_rank = hostmem.to_device(draws_h[:n_hits].astype(np.int64) < T, dev)
    kept = hit.clone()
    kept[hit] = keep_by_rank[rank[hit]]
    fkeys = hostmem.to_device(lists.feature_keys, dev)
    fs, forder = torch.sort(fkeys)
    fpos = torch.searchsorted(fs, gx).clamp_(max=len(fs) - 1)
    feat_ok = (gx != 0) & (fs[fpos] == gx)
    valid = kept & ((meta & 1) != 0) & feat_ok & ((meta & 2) != 0)
    c = cell_idx[valid]; f = forder[fpos[valid]] + 1
    nonnull = ((meta[valid] & 4) != 0).to(torch.int64)
    u = ((umi[valid].to(torch.int64) & 0xFFFFFFFF) >> 8) * nonnull              # 12 bases = the top 24 bits
    code = (c << 42) | (f << 26) | (nonnull << 25) | u
    ucode = torch.unique(code)
    grp, inv = torch.unique_consecutive(ucode >> 26, return_inverse=True)
    cnt = torch.zeros(len(grp), dtype=torch.int64, device=dev).index_add_(0, inv, (ucode >> 25) & 1)
    return dict(hits=n_hits, sampled=int(kept.sum().item()), valid=int(valid.sum().item()),
                cell=hostmem.to_host(grp >> 16), feature=hostmem.to_host(grp & 0xFFFF), count=hostmem.to_host(cnt))


@pytest.mark.own_process
def test_full_size_config3_exact():
    """BASELINE configs[2] at full size on one GPU (fastf_amd/workload.py: 200 M records, 50 k barcodes x 36 601 genes,
    --cell 0.5 --depth 0.5, log-normal cells, Zipf genes, 12-bp UMIs, 5 % without CB, 5 % unlisted CB, 15 % bad xf,
    0.1 % UMIs with N).  The three counters and EVERY row of the matrix — cell, feature and count — are compared with an
    independent computation (torch sorted search / cumsum / unique), for the resident device pass bench.py times and for
    the pinned streaming push path; the -u rows serve as a checksum of the matrix."""
    import torch
    from fastf_amd.dist import HipStages, ShardedPass
    dev = torch.device("cuda", 0)
    N = 200_000_000
    job = workload_mod.C3(N)
    lists = job.lists
    parts = [job.segment_packed(s_, dev) for s_ in range(workload_mod.SEGMENTS)]
    cb, gx, umi, meta = (torch.cat([p[i] for p in parts]) for i in range(4))
    del parts
    job._pool = None
    draws_h = F.mt_draws(workload_mod.SEED, lists.mt_skip, N)
    want = _expected_matrix_torch(job, dev, cb, gx, umi, meta, draws_h)
    torch.cuda.empty_cache()
    assert len(want["cell"]) > 5_000_000 and want["count"].min() == 0           # count-0 rows (N-only groups) are in

    def check(res_cell, res_feature, res_count, counters):
        assert counters == (want["sampled"], want["valid"])
        assert len(res_cell) == len(want["cell"])
        np.testing.assert_array_equal(np.asarray(res_cell, dtype=np.int64), want["cell"])
        np.testing.assert_array_equal(np.asarray(res_feature, dtype=np.int64), want["feature"])
        np.testing.assert_array_equal(np.asarray(res_count, dtype=np.int64), want["count"])

    # 1. the resident pass (streaming K1b -> segmented sort -> reduce), twice on the same buffers
    eng = F.Engine.from_lists(lists, rate_depth=workload_mod.RATE_DEPTH, seed=workload_mod.SEED, umi_max_bases=12)
    try:
        d_draws = hostmem.to_device(draws_h, dev)
        sp = ShardedPass(HipStages(eng, dev), N, dev)
        for _ in range(2):
            sp.run(cb, gx, umi, meta, N, d_draws)
        f_, c_, k_ = sp.local_coo()
        hits, sampled, valid, err = sp.global_counters()
        assert err == 0 and hits == want["hits"]
        check(c_, f_, k_, (sampled, valid))
        del sp, d_draws
    finally:
        eng.close()
    torch.cuda.empty_cache()
    # 2. the streaming push path from pinned host memory (what bam2db() and bench.py's device_path leg use)
    pb = F.PinnedBatch(N)
    pb.fill(0, cb, gx, umi, meta)
    del cb, gx, umi, meta
    torch.cuda.empty_cache()
    eng = F.Engine.from_lists(lists, rate_depth=workload_mod.RATE_DEPTH, seed=workload_mod.SEED, umi_max_bases=12, batch_records=8 << 20)
    try:
        eng.push_pinned(pb)
        res = eng.finish()
        assert res["total"] == N
        check(res["cell"], res["feature"], res["count"], (res["sampled"], res["valid"]))
        r2 = eng.finish()
        assert r2["nnz"] == res["nnz"] and np.array_equal(r2["count"], res["count"])          # idempotent
        rows = eng.umi_rows()
        assert int(rows["n_copy"].sum()) == res["valid"]                      # every valid read is in exactly one -u row
        assert int(rows["nonnull"].sum()) == int(res["count"].sum())          # distinct non-NULL UMIs == sum of the matrix
    finally:
        eng.close()
        pb.close()


@pytest.mark.own_process
def test_adversarial_share_of_config5():
    """One GPU's share of BASELINE configs[4] (1 B records / 8 GPUs = 125 M, 12 500 of the 100 k barcodes), keep-all,
    UMIs Zipf(1.5)-drawn from a 4 096-value pool per gene: most keys are duplicates and a few radix digits dominate.
    Exact check of the matrix against numpy's unique over packed (cell, gene, umi) codes."""
    import torch  # noqa: F401
    N = 125_000_000
    n_cells, n_genes = 12_500, 36_601
    bt, ft, bar, genes = synth.make_lists(n_cells, n_genes, seed=99)
    lists = F.Lists(bt, ft, 1.0, 926)
    rng = np.random.default_rng(9)
    gw = 1.0 / np.arange(1, n_genes + 1) ** 1.1; gw /= gw.sum()
    uw = 1.0 / np.arange(1, 4097) ** 1.5; uw /= uw.sum()
    eng = F.Engine.from_lists(lists, rate_depth=1.0, seed=926, umi_max_bases=12, batch_records=8 << 20)
    codes = []
    try:
        B = 25_000_000
        for off in range(0, N, B):
            n = min(B, N - off)
            c = rng.integers(0, n_cells, size=n).astype(np.int64)
            g = rng.choice(n_genes, size=n, p=gw).astype(np.int64)
            u = rng.choice(4096, size=n, p=uw).astype(np.int64)
            umi = (((u * 2654435761) ^ (g * 40503)) & 0xFFFFFF).astype(np.uint32)   # the gene's own pool of 4 096 UMIs
            codes.append((c << 40) | (g << 24) | umi.astype(np.int64))
            meta = np.full(n, 1 | 2 | 4 | (3 << 4), dtype=np.uint32)
            eng.push(lists.cell_keys[c], lists.feature_keys[g], (umi << np.uint32(8)).astype(np.uint32), meta)
        res = eng.finish()
        assert (res["total"], res["sampled"], res["valid"]) == (N, N, N)
        u = np.unique(np.concatenate(codes))
        del codes
        grp, cnt = np.unique(u >> 24, return_counts=True)
        assert res["nnz"] == len(grp)
        np.testing.assert_array_equal(res["cell"].astype(np.int64), (grp >> 16) + 1)
        np.testing.assert_array_equal(res["feature"].astype(np.int64), (grp & 0xFFFF) + 1)
        np.testing.assert_array_equal(res["count"].astype(np.int64), cnt)
    finally:
        eng.close()


@pytest.mark.own_process
def test_adversarial_config5_whole_on_one_gpu():
    """BASELINE configs[4] WHOLE on one GPU (tools/config5_whole.py): 1 B records, all 100 k barcodes (the L2 cell table),
    keep-all, Zipf-skewed UMI reuse; records generated on the device and pushed as device-resident batches, 1 B keys through
    K2/K3, every row of the 270 M-row matrix compared with torch.unique over the packed codes.  The maximum-size case:
    24 GB of records, 8 GB of keys, indices beyond 2^29 everywhere."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "config5_whole", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "config5_whole.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.run(1_000_000_000, 25_000_000, log=lambda s: print(s))
    assert out["rows"] > 200_000_000
