"""Shared helpers for the parity tests: one synthetic case → oracle result + packed inputs."""
import numpy as np

import fastf_amd as F
from fastf_amd import synth
from oracle import oracle as O


class Case:
    def __init__(self, n, n_bar, n_gene, rate_cell=1.0, rate_depth=1.0, seed=926, data_seed=1, umi_copies=True, gene_stride=1, gene_start=1, **kw):
        self.rate_cell, self.rate_depth, self.seed = rate_cell, rate_depth, seed
        self.bt, self.ft, self.bar, self.genes = synth.make_lists(n_bar, n_gene, seed=data_seed + 1000, gene_stride=gene_stride, gene_start=gene_start)
        fl, xf, cb, gx, ub = synth.make_records(n, self.bar, self.genes, seed=data_seed, **kw)
        self.flags, self.xf = fl, xf
        self.cb, self.gx, self.ub = synth.as_cstr(cb), synth.as_cstr(gx), synth.as_cstr(ub)
        self.n = n
        self.label = b"synthetic.bam"
        self.umi_copies = umi_copies

    def oracle(self):
        return O.run_bam2db(self.bt, self.ft, self.flags, self.xf, self.cb, self.gx, self.ub,
                            self.rate_cell, self.rate_depth, self.seed, self.label, self.umi_copies)

    def lists(self):
        return F.Lists(self.bt, self.ft, self.rate_cell, self.seed)

    def packed(self, lists):
        return F.pack_records(lists, self.flags, self.xf, self.cb, self.gx, self.ub)

    def packed_long(self, lists):
        """five arrays: the fifth holds bases 17.. of every UMI (engines with umi_max_bases > 16)"""
        return F.pack_records(lists, self.flags, self.xf, self.cb, self.gx, self.ub, long_umis=True)


def assert_matches_oracle(res, ora, eng=None, case=None, lists=None, rows=None):
    assert (res["total"], res["sampled"], res["valid"]) == (ora["total"], ora["sampled"], ora["valid"])
    assert res["nnz"] == ora["nnz"]
    np.testing.assert_array_equal(res["cell"].astype(np.int64), ora["cell"].astype(np.int64))
    np.testing.assert_array_equal(res["feature"].astype(np.int64), ora["feature"].astype(np.int64))
    np.testing.assert_array_equal(res["count"].astype(np.int64), ora["count"].astype(np.int64))
    if eng is not None and case is not None:
        txt = eng.format_matrix(res, case.label, case.rate_cell, case.rate_depth, lists.n_features, lists.n_cells)
        assert txt == ora["matrix"]
        assert lists.barcodes_text() == ora["barcodes"]
        assert lists.features_text() == ora["features"]
    if rows is not None:
        assert eng.format_umi_rows(rows) == ora["umi"]
