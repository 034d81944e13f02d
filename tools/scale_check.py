#!/usr/bin/env python3
"""BASELINE configs[2]-scale run on one GPU: 200 M synthetic packed records, 50 k barcodes x 36 601 genes,
--cell 0.5 --depth 0.5, log-normal cells, Zipf genes, 12-bp UMIs with duplication.  Checks size-independent
properties and prints stage timings.  usage: tools/scale_check.py [n_records]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import fastf_amd as F
from fastf_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
t0 = time.time()
bt, ft, bar, genes = synth.make_lists(50_000, 36_601, seed=77)
lists = F.Lists(bt, ft, 0.5, 926)
print("lists: %d of %d barcodes sampled, %d features, mt_skip %d" % (lists.n_cells, lists.n_lines_barcodes, lists.n_features, lists.mt_skip))
# keys of ALL barcodes (sampled or not): records hit the sampled half only
alld = F.Lists(bt, ft, 1.0, 926)
rng = np.random.default_rng(5)
w = rng.lognormal(0, 1, 50_000); w /= w.sum()
gw = 1.0 / np.arange(1, 36_602) ** 1.1; gw /= gw.sum()
n_mol = N // 4
mol_cell = rng.choice(50_000, size=n_mol, p=w).astype(np.int32)
mol_gene = rng.choice(36_601, size=n_mol, p=gw).astype(np.int32)
mol_umi = rng.integers(0, 1 << 24, size=n_mol, dtype=np.uint32)
print("molecules generated %.1fs" % (time.time() - t0))
eng = F.Engine.from_lists(lists, rate_depth=0.5, seed=926, umi_max_bases=12, batch_records=8 << 20)
B = 16_000_000
tot_push = 0.0
kept_expected = 0
for off in range(0, N, B):
    n = min(B, N - off)
    src = rng.integers(0, n_mol, size=n)
    cbk = alld.cell_keys[mol_cell[src]]
    gxk = alld.feature_keys[mol_gene[src]]
    umi = (mol_umi[src] << np.uint32(8)).astype(np.uint32)
    meta = np.full(n, 1 | 2 | 4 | (3 << 4), dtype=np.uint32)
    r = rng.random(n)
    cbk[r < 0.05] = 0                                   # 5 % without CB
    meta[(r > 0.05) & (r < 0.20)] &= ~np.uint32(1)      # 15 % xf not in {25,17}
    meta[r > 0.999] &= ~np.uint32(4)                    # 0.1 % UMIs with N
    t1 = time.time(); eng.push(cbk, gxk, umi, meta); tot_push += time.time() - t1
t1 = time.time(); res = eng.finish(); t_fin = time.time() - t1
print("push total %.2fs (host staging + H2D + K1), finish %.3fs" % (tot_push, t_fin))
print("total %d hits? sampled %d valid %d rows %d" % (res["total"], res["sampled"], res["valid"], res["nnz"]))
assert res["total"] == N
key = res["cell"].astype(np.int64) * (1 << 20) + res["feature"].astype(np.int64)
assert (np.diff(key) > 0).all(), "rows not strictly ascending"
assert res["cell"].min() >= 1 and res["cell"].max() <= lists.n_cells and res["feature"].max() <= lists.n_features
assert int(res["count"].sum()) <= res["valid"]
frac = res["sampled"] / (0.95 * N * 0.5)
print("sampled / (hits x 0.5) = %.4f (cells sampled ~half: expect ~0.5 by read mass)" % frac)
r2 = eng.finish(); assert r2["nnz"] == res["nnz"]
rows = eng.umi_rows(); assert int(rows["n_copy"].sum()) == res["valid"]
assert int(rows["nonnull"].sum()) == int(res["count"].sum())
print("umi rows %d, copies sum == valid, nonnull rows == sum of counts: OK" % rows["n"])
print("done in %.1fs" % (time.time() - t0))
