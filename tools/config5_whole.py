"""BASELINE configs[4] WHOLE on one MI355X: 1 B records, 100 k barcodes x 36 601 genes, keep-all, UMIs Zipf(1.5)-drawn from
a 4 096-value pool per gene (most keys are duplicates, a few radix digits dominate).  The records are generated on the
device in chunks and pushed as device-resident batches (fastf_engine_push_pinned); the matrix is compared row by row with
torch.unique over the packed (cell, gene, umi) codes.  Prints the push and finish times.

    python tools/config5_whole.py [records] [chunk]
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from fastf_amd import hostmem
import torch

import fastf_amd as F
from fastf_amd import synth
from fastf_amd._lib import Batch, check


def run(N=1_000_000_000, B=25_000_000, n_cells=100_000, n_genes=36_601, log=print):
    dev = torch.device("cuda", 0)
    bt, ft, _, _ = synth.make_lists(n_cells, n_genes, seed=99)
    lists = F.Lists(bt, ft, 1.0, 926)
    cell_keys = hostmem.to_device(lists.cell_keys, dev)
    feat_keys = hostmem.to_device(lists.feature_keys, dev)
    gcdf = torch.cumsum(1.0 / torch.arange(1, n_genes + 1, dtype=torch.float64, device=dev) ** 1.1, 0)
    ucdf = torch.cumsum(1.0 / torch.arange(1, 4097, dtype=torch.float64, device=dev) ** 1.5, 0)
    gcdf /= gcdf[-1].clone(); ucdf /= ucdf[-1].clone()
    gen = torch.Generator(device=dev); gen.manual_seed(9)
    codes = torch.empty(N, dtype=torch.int64, device=dev)
    eng = F.Engine.from_lists(lists, rate_depth=1.0, seed=926, umi_max_bases=12, batch_records=8 << 20, key_capacity=N // 2)
    L = eng._L
    try:
        t_push = 0.0
        meta = torch.full((B,), 1 | 2 | 4 | (3 << 4), dtype=torch.int32, device=dev)
        for off in range(0, N, B):
            n = min(B, N - off)
            c = torch.randint(0, n_cells, (n,), device=dev, generator=gen)
            g = torch.searchsorted(gcdf, torch.rand(n, dtype=torch.float64, device=dev, generator=gen)).clamp_(max=n_genes - 1)
            u = torch.searchsorted(ucdf, torch.rand(n, dtype=torch.float64, device=dev, generator=gen)).clamp_(max=4095)
            umi = ((u * 2654435761) ^ (g * 40503)) & 0xFFFFFF                       # the gene's own pool of 4 096 UMIs
            codes[off:off + n] = (c << 40) | (g << 24) | umi
            cbk, gxk = cell_keys[c], feat_keys[g]
            umi32 = (umi << 8).to(torch.int32)                                      # bit pattern of the u32 field (12 bases on top)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b = Batch(cbk.data_ptr(), gxk.data_ptr(), umi32.data_ptr(), meta.data_ptr(), n)
            check(L.fastf_engine_push_pinned(eng._h, C.byref(b)))
            eng.wait_input()
            t_push += time.perf_counter() - t0
        del c, g, u, umi, cbk, gxk, umi32
        t0 = time.perf_counter()
        res = eng.finish(copy=False)
        t_finish = time.perf_counter() - t0
        log("config5 whole: %d records, push %.3f s, finish %.3f s (%.1f M records/s device path), %d rows"
            % (N, t_push, t_finish, N / (t_push + t_finish) / 1e6, res["nnz"]))
        assert (res["total"], res["sampled"], res["valid"]) == (N, N, N)
        uq = torch.unique(codes)
        del codes
        grp, cnt = torch.unique_consecutive(uq >> 24, return_counts=True)
        del uq
        assert res["nnz"] == len(grp)
        for name, want in (("cell", (grp >> 16) + 1), ("feature", (grp & 0xFFFF) + 1), ("count", cnt)):
            got = hostmem.to_device(res[name], dev).to(torch.int64)
            assert torch.equal(got, want), name
            del got
        return dict(records=N, push_s=t_push, finish_s=t_finish, rows=int(res["nnz"]))
    finally:
        eng.close()


if __name__ == "__main__":
    a = [int(float(x)) for x in sys.argv[1:]]
    print(run(*a))
