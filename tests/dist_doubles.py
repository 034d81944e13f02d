"""Test doubles for the device stages of fastf_amd.dist.ShardedPass: numpy restatements of
K1a/K1b/K2/K3 on packed records, so the sharded ORCHESTRATION (draw-rank base, ownership,
the exchange, the global merge) can be exercised over gloo without a GPU."""
import numpy as np
import torch

from fastf_amd.dist import owner_of_cell

META_XF_OK, META_HAS_UB, META_NONNULL = 1, 2, 4


class NumpyStages:
    def __init__(self, cell_keys, feature_keys, threshold, umi_max_bases=12):
        self.cells = {int(k): i + 1 for i, k in enumerate(cell_keys)}
        self.feats = {int(k): i + 1 for i, k in enumerate(feature_keys)}
        self.T = threshold
        self.umi_bits = 2 * umi_max_bases
        self.len_bits = max(1, int((umi_max_bases + 3) // 4).bit_length())
        self.feat_bits = max(1, len(feature_keys).bit_length())
        self.feat_shift = 1 + self.umi_bits + self.len_bits
        self.cell_shift = self.feat_shift + self.feat_bits

    def _cells_of(self, cb, n):
        k = cb[:n].numpy().view(np.uint64)
        return np.array([self.cells.get(int(x), 0) for x in k], dtype=np.int64)

    def count_hits(self, cb, n, out_hits):
        out_hits[0] = int((self._cells_of(cb, n) != 0).sum())

    def probe_pack(self, cb, gx, umi, meta, n, draws, draw_base, keys_out, stride, key_counts, counters, reuse_hits=False):
        cell = self._cells_of(cb, n)
        hit = cell != 0
        rank = np.cumsum(hit) - 1 + int(draw_base[0])
        d = draws.numpy().view(np.uint32)
        keep = hit.copy()
        keep[hit] = d[rank[hit]].astype(np.uint64) < np.uint64(self.T) if self.T < (1 << 32) else True
        m = meta[:n].numpy().view(np.uint32)
        g = gx[:n].numpy().view(np.uint64)
        feat = np.array([self.feats.get(int(x), 0) for x in g], dtype=np.int64)
        valid = keep & ((m & META_XF_OK) != 0) & (feat != 0) & ((m & META_HAS_UB) != 0)
        u = umi[:n].numpy().view(np.uint32).astype(np.uint64)
        nonnull = ((m & META_NONNULL) != 0)
        ln = ((m >> 4) & 7).astype(np.uint64)
        key = (cell.astype(np.uint64) << np.uint64(self.cell_shift)) | (feat.astype(np.uint64) << np.uint64(self.feat_shift))
        ufield = (np.uint64(1) << np.uint64(self.umi_bits + self.len_bits)) | ((u >> np.uint64(32 - self.umi_bits)) << np.uint64(self.len_bits)) | ln
        key = np.where(nonnull, key | ufield, key)
        G = keys_out.shape[0]
        own = owner_of_cell(cell, G) if G > 1 else np.zeros(n, dtype=np.int64)
        for s in range(G):
            ks = key[valid & (own == s)]
            c = int(key_counts[s])
            room = max(0, min(len(ks), stride - c))          # like the device: what does not fit the row is lost, the count goes on
            keys_out[s, c:c + room] = torch.from_numpy(ks[:room].view(np.int64))
            key_counts[s] = c + len(ks)
        counters[0] += int(hit.sum()); counters[1] += int(keep.sum()); counters[2] += int(valid.sum())

    def set_regions(self, counts, n_regions, stride, d_n):
        self.regions = (counts.clone(), n_regions, stride)
        d_n[0] = int(counts.sum())

    def sort_reduce(self, keys, tmp, d_n, max_n, feature, cell, count, nnz, fresh=True):
        n = int(d_n[0])
        if fresh and getattr(self, "regions", None) is not None:      # the keys lie in rows (fixed-capacity exchange)
            counts, R, stride = self.regions
            self.regions = None
            flat = keys.view(-1)
            rows = [flat[r * stride:r * stride + int(counts[r])] for r in range(R)]
            src = torch.cat(rows) if rows else flat[:0]
            assert len(src) == n
            keys = tmp
            keys[:n] = src
        k = np.sort(keys[:n].numpy().view(np.uint64))
        keys[:n] = torch.from_numpy(k.view(np.int64))
        grp = k >> np.uint64(self.feat_shift)
        ug, start = np.unique(grp, return_index=True)
        nn = (k >> np.uint64(self.umi_bits + self.len_bits)) & np.uint64(1)
        uk = np.unique(k[nn == 1])
        cnt = np.zeros(len(ug), dtype=np.int64)
        np.add.at(cnt, np.searchsorted(ug, uk >> np.uint64(self.feat_shift)), 1)
        z = len(ug)
        feature[:z] = torch.from_numpy((ug & np.uint64((1 << self.feat_bits) - 1)).astype(np.int32))
        cell[:z] = torch.from_numpy((ug >> np.uint64(self.feat_bits)).astype(np.int32))
        count[:z] = torch.from_numpy(cnt.astype(np.int32))
        nnz[0] = z
        return keys


class NumpyWideStages(NumpyStages):
    """the same chain with the key in two words (fastf_amd.dist: keys wider than 64 bits): group word = cell << feature_bits |
    feature, value = NULL flag | UMI of up to 24 bases | blob bytes — what fastf_dev_probe_pack_wide / _adopt_wide + finish do"""
    wide = True

    def __init__(self, cell_keys, feature_keys, threshold, umi_max_bases=24):
        assert 16 < umi_max_bases <= 24
        super().__init__(cell_keys, feature_keys, threshold, umi_max_bases)

    def probe_pack(self, *a, **k):
        raise AssertionError("wide stages take probe_pack_wide")

    def probe_pack_wide(self, cb, gx, umi, meta, ext, n, draws, draw_base, keys_out, vals_out, stride, key_counts, counters, reuse_hits=False):
        cell = self._cells_of(cb, n)
        hit = cell != 0
        rank = np.cumsum(hit) - 1 + int(draw_base[0])
        d = draws.numpy().view(np.uint32)
        keep = hit.copy()
        keep[hit] = d[rank[hit]].astype(np.uint64) < np.uint64(self.T) if self.T < (1 << 32) else True
        m = meta[:n].numpy().view(np.uint32)
        g = gx[:n].numpy().view(np.uint64)
        feat = np.array([self.feats.get(int(x), 0) for x in g], dtype=np.int64)
        valid = keep & ((m & META_XF_OK) != 0) & (feat != 0) & ((m & META_HAS_UB) != 0)
        u = umi[:n].numpy().view(np.uint32).astype(np.uint64) << np.uint64(32)
        if ext is not None:
            u |= ext[:n].numpy().view(np.uint32).astype(np.uint64)
        nonnull = (m & META_NONNULL) != 0
        ln = ((m >> 4) & 15).astype(np.uint64)
        key = (cell.astype(np.uint64) << np.uint64(self.feat_bits)) | feat.astype(np.uint64)
        val = (np.uint64(1) << np.uint64(self.umi_bits + self.len_bits)) | ((u >> np.uint64(64 - self.umi_bits)) << np.uint64(self.len_bits)) | ln
        val = np.where(nonnull, val, np.uint64(0))
        G = keys_out.shape[0]
        own = owner_of_cell(cell, G) if G > 1 else np.zeros(n, dtype=np.int64)
        for s in range(G):
            sel = valid & (own == s)
            ks, vs = key[sel], val[sel]
            c = int(key_counts[s])
            room = max(0, min(len(ks), stride - c))
            keys_out[s, c:c + room] = torch.from_numpy(ks[:room].view(np.int64))
            vals_out[s, c:c + room] = torch.from_numpy(vs[:room].view(np.int64))
            key_counts[s] = c + len(ks)
        counters[0] += int(hit.sum()); counters[1] += int(keep.sum()); counters[2] += int(valid.sum())

    def reduce_wide(self, keys, vals, n):
        k = keys[:n].numpy().view(np.uint64)
        v = vals[:n].numpy().view(np.uint64)
        order = np.lexsort((v, k))
        k, v = k[order], v[order]
        ug = np.unique(k)
        new = np.ones(n, dtype=bool)
        new[1:] = (k[1:] != k[:-1]) | (v[1:] != v[:-1])
        dist_nn = new & (v != 0)                          # a distinct non-NULL (group, value) pair
        cnt = np.zeros(len(ug), dtype=np.int64)
        np.add.at(cnt, np.searchsorted(ug, k[dist_nn]), 1)
        return ((ug & np.uint64((1 << self.feat_bits) - 1)).astype(np.int64), (ug >> np.uint64(self.feat_bits)).astype(np.int64), cnt)
