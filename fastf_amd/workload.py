"""BASELINE configs[2] ("C3") as packed SoA, generated segment by segment.

    200 M records, 50 k barcodes x 36 601 genes, --cell 0.5 --depth 0.5 --seed 926;
    cell popularity log-normal (sigma 1), gene popularity Zipf(1.1), 12-bp UMIs, reads drawn from
    N/4 molecules (duplication factor 4); 5 % of the records carry no CB, 5 % a CB that is not in
    the barcode list (same length and "-1" suffix as the listed ones: the hard kind of miss),
    15 % an xf outside {25, 17}, 0.1 % a UMI with an N.

The job is a fixed sequence of SEGMENTS record segments, each generated from its own seed, so that
every rank count G that divides SEGMENTS sees the same job (rank r owns segments [r S/G, (r+1) S/G)),
and the first records of segment 0 can be rebuilt as strings for the CPU oracle.
Generators are numpy/torch PCG/Philox streams — never the MT19937 stream under test.
"""
import numpy as np

from . import synth

SEGMENTS = 8
N_BARCODES, N_GENES, UMI_LEN = 50_000, 36_601, 12
RATE_CELL, RATE_DEPTH, SEED = 0.5, 0.5, 926
P_NO_CB, P_UNLISTED, P_BAD_XF, P_N_UMI = 0.05, 0.05, 0.15, 0.001
META_FULL = 1 | 2 | 4 | (3 << 4)          # xf ok, UB present, every base in ACGT, blob length 3 bytes
# what the 200 M-record job comes to (single-GPU run, checked row by row against an independent torch computation in
# tests/test_gpu_e2e.py::test_full_size_config3_exact): every rank count must reproduce these totals
EXPECTED_200M = {"total": 200_000_000, "hits": 90_369_735, "sampled": 45_180_593, "valid": 38_401_542,
                 "keys": 38_401_542, "rows": 8_507_596}


def describe(n_total):
    return ("BASELINE configs[2]: %d synthetic records, %d barcodes x %d genes, --cell %.1f --depth %.1f --seed %d, "
            "log-normal cells, Zipf(1.1) genes, %d-bp UMIs from N/4 molecules, %.0f %% no CB, %.0f %% unlisted CB, "
            "%.0f %% bad xf, %.1f %% UMIs with N" % (n_total, N_BARCODES, N_GENES, RATE_CELL, RATE_DEPTH, SEED, UMI_LEN,
                                                  100 * P_NO_CB, 100 * P_UNLISTED, 100 * P_BAD_XF, 100 * P_N_UMI))


class C3:
    """lists + molecule pool of the job (identical on every rank)"""

    def __init__(self, n_total, list_seed=77, pool_seed=5):
        import fastf_amd as F
        self.n_total = int(n_total)
        self.seg_len = self.n_total // SEGMENTS
        assert self.seg_len * SEGMENTS == self.n_total, "records must divide into %d segments" % SEGMENTS
        self.bt, self.ft, self.bar, self.genes = synth.make_lists(N_BARCODES, N_GENES, seed=list_seed)
        self.lists = F.Lists(self.bt, self.ft, RATE_CELL, SEED)          # the sampled half: cell_index 1..25 000
        self.all_lists = F.Lists(self.bt, self.ft, 1.0, SEED)            # keys of every barcode line
        rng = np.random.default_rng(pool_seed)
        w = rng.lognormal(0.0, 1.0, N_BARCODES)
        self.cell_cdf = np.cumsum(w / w.sum())
        gw = 1.0 / np.arange(1, N_GENES + 1) ** 1.1
        self.gene_cdf = np.cumsum(gw / gw.sum())
        self.n_mol = max(1, self.n_total // 4)
        self.pool_seed = pool_seed
        self._pool = None
        fam = int(self.all_lists.cell_keys[0]) & ~((1 << 48) - 1)         # form, length and "-1" suffix of the listed barcodes
        self.cb_family = fam

    # ---- device generation (torch): used by bench.py and the full-size GPU tests ----
    def pool(self, dev):
        import torch
        if self._pool is None:
            g = torch.Generator(device=dev); g.manual_seed(1000 + self.pool_seed)
            from .hostmem import to_device
            ccdf = to_device(self.cell_cdf, dev); gcdf = to_device(self.gene_cdf, dev)
            u = torch.rand(self.n_mol, device=dev, dtype=torch.float64, generator=g)
            cell = torch.searchsorted(ccdf, u).clamp_(max=N_BARCODES - 1).to(torch.int32)
            u = torch.rand(self.n_mol, device=dev, dtype=torch.float64, generator=g)
            gene = torch.searchsorted(gcdf, u).clamp_(max=N_GENES - 1).to(torch.int32)
            umi = torch.randint(0, 1 << (2 * UMI_LEN), (self.n_mol,), device=dev, dtype=torch.int32, generator=g)
            self._pool = (cell, gene, umi)
        return self._pool

    def segment_indices(self, seg, dev, n=None):
        """(cell line 0.., gene 0.., umi code, kind, unlisted code) of the first n records of a segment.
        kind bits: 1 no CB, 2 unlisted CB, 4 bad xf, 8 UMI with N"""
        import torch
        n = self.seg_len if n is None else int(n)
        cell, gene, umi = self.pool(dev)
        g = torch.Generator(device=dev); g.manual_seed(7_000 + seg)
        # the draws below are made for the whole segment so that a prefix of a segment is a prefix of its records
        src = torch.randint(0, self.n_mol, (self.seg_len,), device=dev, dtype=torch.int64, generator=g)[:n]
        r = torch.rand(self.seg_len, device=dev, dtype=torch.float32, generator=g)[:n]
        r2 = torch.rand(self.seg_len, device=dev, dtype=torch.float32, generator=g)[:n]
        alt = torch.randint(0, 1 << 32, (self.seg_len,), device=dev, dtype=torch.int64, generator=g)[:n]
        kind = torch.zeros(n, dtype=torch.int32, device=dev)
        kind |= (r < P_NO_CB).to(torch.int32)
        kind |= ((r >= P_NO_CB) & (r < P_NO_CB + P_UNLISTED)).to(torch.int32) << 1
        kind |= (r2 < P_BAD_XF).to(torch.int32) << 2
        kind |= (r2 > 1.0 - P_N_UMI).to(torch.int32) << 3
        return cell[src], gene[src], umi[src], kind, alt

    def segment_packed(self, seg, dev, n=None):
        """packed SoA of a segment on the device: cb_key i64, gx_key i64, umi i32, meta i32 (bit patterns of u64/u32)"""
        import torch
        c, g, u, kind, alt = self.segment_indices(seg, dev, n)
        from .hostmem import to_device
        ck = to_device(self.all_lists.cell_keys, dev)
        fk = to_device(self.all_lists.feature_keys, dev)
        cb = ck[c.long()]
        cb = torch.where((kind & 2) != 0, (alt << 16) | self.cb_family, cb)
        cb = torch.where((kind & 1) != 0, torch.zeros_like(cb), cb)
        gx = fk[g.long()]
        umi = (u << (32 - 2 * UMI_LEN)).to(torch.int32)
        meta = torch.full_like(umi, META_FULL)
        meta = torch.where((kind & 4) != 0, meta & ~1, meta)
        meta = torch.where((kind & 8) != 0, meta & ~4, meta)
        return cb, gx, umi, meta

    # ---- the same records as strings (CPU oracle sample) ----
    def segment_strings(self, seg, dev, n):
        """(flags u8, xf i32, cb S, gx S, ub S) of the first n records of a segment, as oracle/fastf_oracle.c takes them"""
        from .hostmem import to_host
        c, g, u, kind, alt = (to_host(t) for t in self.segment_indices(seg, dev, n))
        cb = self.bar[c].copy()
        un = (kind & 2) != 0
        if un.any():
            m = synth._kmers(alt[un].astype(np.uint64), 16)
            suf = np.tile(np.frombuffer(b"-1", dtype=np.uint8), (int(un.sum()), 1))
            cb[un] = synth._as_S(np.concatenate([m, suf], axis=1), self.bar.dtype.itemsize)
        gx = self.genes[g].copy()
        ubm = synth._kmers(u.astype(np.uint64), UMI_LEN)
        nn = np.nonzero(kind & 8)[0]
        ubm[nn, nn % UMI_LEN] = ord("N")
        ub = synth._as_S(ubm, UMI_LEN + 1)
        flags = np.full(n, synth.HAS_CB | synth.HAS_XF | synth.HAS_GX | synth.HAS_UB, dtype=np.uint8)
        flags[(kind & 1) != 0] &= ~np.uint8(synth.HAS_CB)
        xf = np.where((np.arange(n) & 3) == 0, 17, 25).astype(np.int32)
        xf[(kind & 4) != 0] = 0
        return flags, xf, synth.as_cstr(cb), synth.as_cstr(gx), synth.as_cstr(ub)


class C2:
    """BASELINE configs[1]: 10 M synthetic records, 10 k barcodes x 30 k genes, keep-all (--cell 1 --depth 1), uniform
    cells and genes, 10-bp UMIs — round 1's bench workload, kept for comparison (bench.py --workload c2).
    Same interface as C3; generated on the host through the string-level generator and the product's own packer."""

    def __init__(self, n_total=10_000_000):
        import fastf_amd as F
        self.n_total = int(n_total)
        self.seg_len = self.n_total // SEGMENTS
        assert self.seg_len * SEGMENTS == self.n_total
        self.bt, self.ft, self.bar, self.genes = synth.make_lists(10_000, 30_000, seed=4242)
        self.lists = F.Lists(self.bt, self.ft, 1.0, SEED)
        self._cache = {}
        self._pool = None
        self._F = F

    def _seg(self, seg):
        if seg not in self._cache:
            fl, xf, cb, gx, ub = synth.make_records(self.seg_len, self.bar, self.genes, seed=100 + seg, umi_len=10)
            s = (fl, xf, synth.as_cstr(cb), synth.as_cstr(gx), synth.as_cstr(ub))
            self._cache[seg] = (s, self._F.pack_records(self.lists, *s))
        return self._cache[seg]

    def segment_packed(self, seg, dev, n=None):
        import torch
        n = self.seg_len if n is None else int(n)
        cbk, gxk, umi, meta = self._seg(seg)[1]
        from .hostmem import to_device
        return (to_device(cbk[:n], dev), to_device(gxk[:n], dev), to_device(umi[:n], dev), to_device(meta[:n], dev))

    def segment_strings(self, seg, dev, n):
        return tuple(a[:n] for a in self._seg(seg)[0])
